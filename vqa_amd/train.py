"""Training / evaluation driver with the reference's command line
(vqa_model/train.py:241-252, vqa_model/test.py:119-128):

    python -m vqa_amd.train --model VQAAttnModel --dataset SYNTH --epochs 2 --bs 256 --gpu 0
    (--model VQAGAPModel and --model VQAModel, the reference's class names, take precomputed ResNet maps)

Same epoch structure as the reference (validation first, then training, train.py:48-228): per-epoch
validation loss / OA / AA, running training loss = sum(loss * B) / len(dataset), best-on-validation and
running checkpoints under the reference's file names.  ``--dataset SYNTH`` uses device-resident synthetic
VQA-v2-shaped data (precomputed region features); the LR / HR datasets of the reference need image files
and a feature extractor that are outside this package.
"""
from __future__ import annotations

import argparse
import os

import torch

from .metrics import AccuracyMeter, HR_TYPES
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


def evaluate(model, dataset, meter):
    """One no-grad pass: mean loss over the dataset, per-type accuracy, OA, AA (train.py:50-161)."""
    model.eval()
    meter.reset()
    running = torch.zeros(1, device=meter.counters.device)
    with torch.no_grad():
        for question, answer, image, types in dataset:
            model._bptt = False
            model._forward_impl(*model._prepare_inputs(image, question, None))
            running += model.loss_and_grad_logits(answer) * question.shape[0]
            meter.update(model, answer, types)
    res = meter.result()
    res['loss'] = float(running) / len(dataset)
    model.check_status()
    return res


def train(model, train_dataset, validate_dataset, num_epochs, learning_rate, args, Dataset='SYNTH', out_dir=None,
          resume=None):
    """resume: a TrainState_<model>.pth written by an earlier call (weights, Adam moments, epoch)."""
    meter = AccuracyMeter(HR_TYPES, device=next(model.parameters()).device)
    reducer = GradReducer()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        # data-parallel ranks share the weights' seed but must not share dropout masks (they hash the local row)
        model.dropout_rank = torch.distributed.get_rank()
    min_val, best_epoch, history, first_epoch = float('inf'), 0, [], 0
    if resume:
        st = torch.load(resume, weights_only=True)
        model.load_state_dict(st['model'])
        model.load_optimizer_state_dict(st['optimizer'])
        first_epoch, min_val, best_epoch = int(st['epoch']) + 1, float(st['min_val']), int(st['best_epoch'])
    for epoch in range(first_epoch, num_epochs):
        print('Epoch Number:', epoch)
        val = evaluate(model, validate_dataset, meter)
        print('epoch #%d val loss: %.3f' % (epoch, val['loss']))
        print('epoch #%d OA: %.3f' % (epoch, val['OA']))
        print('epoch #%d AA: %.3f' % (epoch, val['AA']))
        model.train()
        running = torch.zeros(1, device=meter.counters.device)
        for question, answer, image, _ in train_dataset:
            loss = model.train_step(image, question, answer, lr=learning_rate, reducer=reducer)
            running += loss * question.shape[0]
        train_loss = float(running) / len(train_dataset)
        model.check_status()
        print('epoch #%d loss: %.3f' % (epoch, train_loss))
        history.append({'val': val, 'train_loss': train_loss})
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            if val['loss'] < min_val:
                min_val, best_epoch = val['loss'], epoch
                torch.save(model.state_dict(), f'{out_dir}/ModelBestRSVQA_{args.model}.pth')
            torch.save(model.state_dict(), f'{out_dir}/ModelRunningRSVQA_{args.model}.pth')
            torch.save({'model': model.state_dict(), 'optimizer': model.optimizer_state_dict(), 'epoch': epoch,
                        'min_val': min_val, 'best_epoch': best_epoch}, f'{out_dir}/TrainState_{args.model}.pth')
    return history


def model_and_geometry(args, dev, **model_kw):
    """The model class chosen by ``args.model`` (the if / elif chain of train.py:311-338, test.py:176-203) on
    ``dev`` and the shapes of the synthetic data that feeds it."""
    classes = {'VQAAttnModel': VQAAttnModel, 'VQAGAPModel': VQAGAPModel, 'VQAModel': VQAModel}
    if args.model not in classes:
        raise SystemExit('--model must be one of ' + ', '.join(classes))
    Vq, C, L, R, Dv = 10000, 3000, 14, 196, 2048
    input_size, stage_maps = 512, None
    if args.model == 'VQAGAPModel':       # ResNet-152 stage outputs of a 256 x 256 image (the LR tiles)
        input_size, stage_maps = 256, ((256, 64), (512, 32), (1024, 16), (2048, 8))
    elif args.model == 'VQAModel':        # last ResNet map of a 512 x 512 image: 16 x 16 x 2048
        R = (input_size // 32) ** 2
    model = classes[args.model](['<EOS>'] + [f'w{i}' for i in range(Vq - 1)], ['<EOS>'] + [f'a{i}' for i in range(C - 1)],
                                args, input_size, **model_kw).to(dev)
    return model, {'Vq': Vq, 'C': C, 'L': L, 'R': R, 'Dv': Dv, 'stage_maps': stage_maps}


def main(argv=None):
    parser = argparse.ArgumentParser(description='MI355X VQA training')
    parser.add_argument('--bs', type=int, default=40)
    parser.add_argument('--gpu', type=int, default=0)
    parser.add_argument('--epochs', type=int, default=150)
    parser.add_argument('--lr', type=float, default=0.00001)
    parser.add_argument('--model', type=str, required=True)
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--out', type=str, default=None, help='checkpoint directory (reference: ../data/dump-models/<dataset>)')
    parser.add_argument('--resume', type=str, default=None, help='TrainState_<model>.pth to continue from')
    args = parser.parse_args(argv)
    if args.dataset != 'SYNTH':
        raise SystemExit('only --dataset SYNTH is available (LR / HR need image files and a feature extractor)')
    dev = torch.device(f'cuda:{args.gpu}')
    torch.cuda.set_device(dev)
    model, geo = model_and_geometry(args, dev)
    train_ds = SynthVQADataset(8, args.bs, geo['R'], geo['Dv'], geo['L'], geo['Vq'], geo['C'], dev, seed=42,
                               stage_maps=geo['stage_maps'])
    val_ds = SynthVQADataset(2, args.bs, geo['R'], geo['Dv'], geo['L'], geo['Vq'], geo['C'], dev, seed=43,
                             stage_maps=geo['stage_maps'])
    return train(model, train_ds, val_ds, args.epochs, args.lr, args, args.dataset, args.out, args.resume)


if __name__ == '__main__':
    main()
