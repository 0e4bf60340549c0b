"""Training / evaluation driver with the reference's command line
(vqa_model/train.py:241-252, vqa_model/test.py:119-128):

    python -m vqa_amd.train --model VQAAttnModel --dataset SYNTH --epochs 2 --bs 256 --gpu 0
    python -m vqa_amd.train --model VQAAttnModel --dataset LR --data-path ../data/lrdataset --features-dir FEATS
    (--model VQAGAPModel and --model VQAModel, the reference's class names, take precomputed ResNet maps)

Same epoch structure as the reference (validation first, then training, train.py:48-228): per-epoch
validation loss / OA / AA, running training loss = sum(loss * B) / len(dataset), best-on-validation and
running checkpoints under the reference's file names.

``--dataset SYNTH`` uses device-resident synthetic VQA-v2-shaped data (precomputed region features).
``--dataset LR`` / ``HR`` read the reference's JSON files under ``--data-path`` with the reference's file names
(train.py:260-285), build the two VocabEncoders from the ``all_*`` files (train.py:286-290) and feed the model
through FeatureVQALoader + a DataLoader (train.py:30-31, 304-305); the image element of an item is the
precomputed feature file ``<features-dir>/<image id>.npy`` (``.npz`` of stage maps for VQAGAPModel), because the
frozen ResNet-152 is outside this package.

Data parallel: started by ``python -m torch.distributed.run --nproc-per-node N -m vqa_amd.train ...`` the driver is
one process per GPU (``LOCAL_RANK`` picks the device; RCCL through torch.distributed's "nccl" backend).  Every rank
holds the same weights, works on its own share of the data, and the gradient buckets are all-reduced behind
backward (GradReducer).  Validation loss and the accuracy counters are summed over ranks, so every rank takes the
same best-epoch decision; only rank 0 prints and writes checkpoints, with a barrier behind the files.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from .metrics import AccuracyMeter, HR_TYPES, LR_TYPES
from .model import VQAAttnModel, VQAGAPModel, VQAModel
from .parallel import GradReducer


class SynthVQADataset:
    """Synthetic stand-in for VQALoader: batches of (question int64 (B,L), answer int64 (B,),
    feats (B,R,Dv), type ids) already on the device (tuple order of VQALoader.py:176)."""

    def __init__(self, n_batches, B, R, Dv, L, Vq, C, device, seed, dtype=torch.bfloat16, stage_maps=None):
        """stage_maps: ((C_k, grid_k), ...) makes the image item the list of NCHW stage maps that
        VQAGAPModel takes instead of (B, R, Dv) region features."""
        g = torch.Generator(device=device).manual_seed(seed)
        self.batches = []
        for _ in range(n_batches):
            if stage_maps:
                feats = [torch.randn((B, c, s, s), device=device, generator=g).clamp_min_(0).to(dtype) for c, s in stage_maps]
            else:
                feats = torch.randn((B, R, Dv), device=device, generator=g).clamp_min_(0).to(dtype)
            lens = torch.randint(3, L + 1, (B,), device=device, generator=g)
            tok = torch.randint(1, Vq + 1, (B, L), device=device, generator=g)
            tok = torch.where(torch.arange(L, device=device)[None, :] < (lens[:, None] - 1), tok, torch.zeros_like(tok))
            # learnable synthetic rule so that the loss visibly falls: the answer is a function of two tokens
            ans = (tok[:, 0] * 7 + tok[:, 1] * 13) % C
            types = torch.randint(0, 4, (B,), device=device, generator=g, dtype=torch.int32)
            self.batches.append((tok, ans, feats, types))
        self.B = B

    def __len__(self):
        return len(self.batches) * self.B

    def __iter__(self):
        return iter(self.batches)


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized()


def _rank_world():
    return (dist.get_rank(), dist.get_world_size()) if _dist_on() else (0, 1)


def _sum_over_ranks(values, device):
    """[float, ...] summed over the ranks of the process group (fp64; the identity without one)."""
    if not _dist_on() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def evaluate(model, dataset, meter):
    """One no-grad pass: mean loss over the dataset, per-type accuracy, OA, AA (train.py:50-161).  Under a
    process group every rank passes over its own share; the loss sum, the sample count and the integer accuracy
    counters are summed over ranks, so the figures are those of one process on the whole set."""
    model.eval()
    meter.reset()
    dev = meter.counters.device
    running = torch.zeros(1, device=dev)
    with torch.no_grad():
        for question, answer, image, types in dataset:
            running += model.eval_step(image, question, answer) * question.shape[0]
            meter.update(model, answer, types)
    res = meter.result()
    loss_sum, n = _sum_over_ranks([float(running), len(dataset)], dev)
    res['loss'] = loss_sum / n
    res['samples'] = int(n)
    model.check_status()
    return res


def train(model, train_dataset, validate_dataset, num_epochs, learning_rate, args, Dataset='SYNTH', out_dir=None,
          resume=None, reducer=None, out=print):
    """The epoch loop of train.py:48-233.  resume: a TrainState_<model>.pth written by an earlier call (weights,
    Adam moments, epoch).  reducer: the GradReducer to use (default: one over the default process group, which
    is a no-op without one)."""
    rank, world = _rank_world()
    say = out if rank == 0 else (lambda *a, **k: None)
    types = LR_TYPES if Dataset == 'LR' else HR_TYPES
    dev = next(model.parameters()).device
    meter = AccuracyMeter(types, device=dev)
    if reducer is None:
        reducer = GradReducer()
    # data-parallel ranks share the weights' seed but must not share dropout masks (they hash the local row)
    model.dropout_rank = rank
    min_val, best_epoch, history, first_epoch = float('inf'), 0, [], 0
    if resume:
        st = torch.load(resume, map_location=dev, weights_only=True)
        model.load_state_dict(st['model'])
        model.load_optimizer_state_dict(st['optimizer'])
        first_epoch, min_val, best_epoch = int(st['epoch']) + 1, float(st['min_val']), int(st['best_epoch'])
    for epoch in range(first_epoch, num_epochs):
        say('Epoch Number:', epoch)
        val = evaluate(model, validate_dataset, meter)
        say('epoch #%d val loss: %.3f' % (epoch, val['loss']))
        say('epoch #%d OA: %.3f' % (epoch, val['OA']))
        say('epoch #%d AA: %.3f' % (epoch, val['AA']))
        model.train()
        running = torch.zeros(1, device=dev)
        for question, answer, image, _ in train_dataset:
            loss = model.train_step(image, question, answer, lr=learning_rate, reducer=reducer)
            running += loss * question.shape[0]
        model.sync(check=False)                 # a deferred update of the last step belongs to this epoch
        loss_sum, n = _sum_over_ranks([float(running), len(train_dataset)], dev)
        train_loss = loss_sum / n
        model.check_status()
        say('epoch #%d loss: %.3f' % (epoch, train_loss))
        history.append({'val': val, 'train_loss': train_loss})
        if val['loss'] < min_val:               # the reduced loss: the same decision on every rank
            min_val, best_epoch, is_best = val['loss'], epoch, True
        else:
            is_best = False
        if out_dir:
            if rank == 0:                       # the weights are identical on every rank: one writer
                os.makedirs(out_dir, exist_ok=True)
                sd = model.state_dict()
                if is_best:
                    torch.save(sd, f'{out_dir}/ModelBestRSVQA_{args.model}.pth')
                torch.save(sd, f'{out_dir}/ModelRunningRSVQA_{args.model}.pth')
                torch.save({'model': sd, 'optimizer': model.optimizer_state_dict(), 'epoch': epoch,
                            'min_val': min_val, 'best_epoch': best_epoch}, f'{out_dir}/TrainState_{args.model}.pth')
            if world > 1:
                dist.barrier()                  # nobody runs ahead of (or exits before) the files being complete
    if history:
        b = best_epoch - first_epoch
        if 0 <= b < len(history):               # train.py:230-233
            say('Best epoch #%d loss: %.3f' % (best_epoch, history[b]['train_loss']))
            say('Best epoch #%d val loss: %.3f' % (best_epoch, history[b]['val']['loss']))
            say('Best epoch #%d OA: %.3f' % (best_epoch, history[b]['val']['OA']))
            say('Best epoch #%d AA: %.3f' % (best_epoch, history[b]['val']['AA']))
    return history


MODEL_CLASSES = {'VQAAttnModel': VQAAttnModel, 'VQAGAPModel': VQAGAPModel, 'VQAModel': VQAModel}


def model_and_geometry(args, dev, **model_kw):
    """The model class chosen by ``args.model`` (the if / elif chain of train.py:311-338, test.py:176-203) on
    ``dev`` and the shapes of the synthetic data that feeds it."""
    if args.model not in MODEL_CLASSES:
        raise SystemExit('--model must be one of ' + ', '.join(MODEL_CLASSES))
    Vq, C, L, R, Dv = 10000, 3000, 14, 196, 2048
    input_size, stage_maps = 512, None
    if args.model == 'VQAGAPModel':       # ResNet-152 stage outputs of a 256 x 256 image (the LR tiles)
        input_size, stage_maps = 256, ((256, 64), (512, 32), (1024, 16), (2048, 8))
    elif args.model == 'VQAModel':        # last ResNet map of a 512 x 512 image: 16 x 16 x 2048
        R = (input_size // 32) ** 2
    model = MODEL_CLASSES[args.model](['<EOS>'] + [f'w{i}' for i in range(Vq - 1)], ['<EOS>'] + [f'a{i}' for i in range(C - 1)],
                                      args, input_size, **model_kw).to(dev)
    return model, {'Vq': Vq, 'C': C, 'L': L, 'R': R, 'Dv': Dv, 'stage_maps': stage_maps}


def dataset_files(Dataset: str, data_path: str) -> dict:
    """The reference's file table (train.py:260-285, test.py:136-151) under ``data_path``."""
    j = lambda name: os.path.join(data_path, name)
    if Dataset == 'LR':
        pre, allq, alla = 'LR_split_', 'all_questions.json', 'all_answers.json'
    else:
        pre, allq, alla = 'USGS_split_', 'USGSquestions.json', 'USGSanswers.json'
    files = {'all_questions': j(allq), 'all_answers': j(alla)}
    for split in ('train', 'val', 'test'):
        for kind in ('questions', 'answers', 'images'):
            files[f'{split}_{kind}'] = j(f'{pre}{split}_{kind}.json')
    return files


def real_data(args, dev, split_names, rank=0, world=1, shuffle_first=True, feature_dtype=torch.bfloat16, **model_kw):
    """Vocabulary encoders, the datasets of ``split_names`` (e.g. ('train', 'val')) and the model for
    ``--dataset LR | HR`` (train.py:286-343): VocabEncoder over the all-questions / all-answers files
    (``range_numbers`` for LR only, train.py:287-290), FeatureVQALoader per split, batches through a DataLoader."""
    from .loader import FeatureBatches, FeatureVQALoader
    from .vocab_encoder import VocabEncoder, MAX_ANSWERS, LEN_QUESTION
    if not args.features_dir:
        raise SystemExit('--dataset LR / HR needs --features-dir (precomputed <image id>.npy / .npz feature files)')
    files = dataset_files(args.dataset, args.data_path or ('../data/lrdataset/' if args.dataset == 'LR' else '../data/hrdataset/'))
    enc_q = VocabEncoder(files['all_questions'], questions=True,
                         len_question=getattr(args, 'len_question', None) or LEN_QUESTION)
    enc_a = VocabEncoder(files['all_answers'], questions=False, range_numbers=args.dataset == 'LR',
                         max_answers=getattr(args, 'max_answers', None) or MAX_ANSWERS)
    types = LR_TYPES if args.dataset == 'LR' else HR_TYPES
    patch_size = 256 if args.dataset == 'LR' else 512            # train.py:299-302
    sets = []
    for i, split in enumerate(split_names):
        is_train = split == 'train'
        ds = FeatureVQALoader(args.features_dir, files[f'{split}_images'], files[f'{split}_questions'],
                              files[f'{split}_answers'], enc_q, enc_a, train=is_train)
        sets.append(FeatureBatches(ds, args.bs, dev, types, shuffle=is_train and shuffle_first, feature_dtype=feature_dtype,
                                   num_workers=getattr(args, 'workers', 2), rank=rank, world=world, drop_uneven=is_train))
    if args.model not in MODEL_CLASSES:
        raise SystemExit('--model must be one of ' + ', '.join(MODEL_CLASSES))
    model = MODEL_CLASSES[args.model](enc_q.getVocab(), enc_a.getVocab(), args, patch_size, **model_kw).to(dev)
    return model, sets, enc_q, enc_a


def add_data_arguments(parser):
    parser.add_argument('--data-path', type=str, default=None,
                        help='LR / HR: folder with the JSON files (reference: ../data/lrdataset/, ../data/hrdataset/)')
    parser.add_argument('--features-dir', type=str, default=None,
                        help='LR / HR: folder with one precomputed feature file per image id (<id>.npy or <id>.npz)')
    parser.add_argument('--max-answers', type=int, default=None, help='answer vocabulary cap (reference: 100)')
    parser.add_argument('--len-question', type=int, default=None, help='question length in tokens (reference: 20)')
    parser.add_argument('--workers', type=int, default=2, help='DataLoader worker processes (reference: 2)')


def init_distributed(args):
    """(rank, world, device, started: this call created the process group).  Under torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the
    process joins the "nccl" (RCCL) group on the device of its local rank -- the environment is read before anything
    touches a GPU; otherwise the device is ``--gpu``, as upstream (train.py:312)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    gpu = int(os.environ.get('LOCAL_RANK', '0')) if launched else args.gpu
    dev = torch.device(f'cuda:{gpu}')
    torch.cuda.set_device(dev)
    started = launched and not _dist_on()
    if started:
        from .parallel import connect_collectives, reserve_compute_units_for_resident_kernels
        reserve_compute_units_for_resident_kernels()
        dist.init_process_group('nccl', device_id=dev)
        connect_collectives(dev)
    return rank, world, dev, started


def main(argv=None, **model_kw):
    parser = argparse.ArgumentParser(description='MI355X VQA training')
    parser.add_argument('--bs', type=int, default=40)
    parser.add_argument('--gpu', type=int, default=0)
    parser.add_argument('--epochs', type=int, default=150)
    parser.add_argument('--lr', type=float, default=0.00001)
    parser.add_argument('--model', type=str, required=True)
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--out', type=str, default=None, help='checkpoint directory (reference: ../data/dump-models/<dataset>)')
    parser.add_argument('--resume', type=str, default=None, help='TrainState_<model>.pth to continue from')
    parser.add_argument('--synth-batches', type=int, default=8, help='SYNTH: training batches per epoch and rank')
    parser.add_argument('--force-collectives', action='store_true',
                        help='issue the gradient collectives even in a group of one rank (exercises the RCCL path on one GPU)')
    add_data_arguments(parser)
    args = parser.parse_args(argv)
    if args.dataset not in ('SYNTH', 'LR', 'HR'):
        raise SystemExit('--dataset must be SYNTH, LR or HR')
    rank, world, dev, we_started_the_group = init_distributed(args)
    try:
        if args.dataset == 'SYNTH':
            model, geo = model_and_geometry(args, dev, **model_kw)
            # every rank draws its own share of the synthetic data
            train_ds = SynthVQADataset(args.synth_batches, args.bs, geo['R'], geo['Dv'], geo['L'], geo['Vq'], geo['C'], dev,
                                       seed=42 + 1000 * rank, stage_maps=geo['stage_maps'])
            val_ds = SynthVQADataset(2, args.bs, geo['R'], geo['Dv'], geo['L'], geo['Vq'], geo['C'], dev,
                                     seed=43 + 1000 * rank, stage_maps=geo['stage_maps'])
        else:
            dt = model_kw.get('dtype', torch.bfloat16)
            model, (train_ds, val_ds), _, _ = real_data(args, dev, ('train', 'val'), rank, world, feature_dtype=dt, **model_kw)
        reducer = GradReducer(force=args.force_collectives)
        return train(model, train_ds, val_ds, args.epochs, args.lr, args, args.dataset, args.out, args.resume,
                     reducer=reducer)
    finally:
        if we_started_the_group and _dist_on():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
