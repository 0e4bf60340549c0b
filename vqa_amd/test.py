"""Evaluation driver with the reference's command line (vqa_model/test.py:119-128):

    python -m vqa_amd.test --model VQAAttnModel --dataset SYNTH --bs 256 --gpu 0 [--models-dir DIR]
    python -m vqa_amd.test --model VQAAttnModel --dataset LR --data-path ../data/lrdataset --features-dir FEATS

``--dataset LR`` / ``HR`` evaluate the reference's test split (``*_split_test_*.json`` under ``--data-path``,
test.py:136-151) through VocabEncoder + FeatureVQALoader + a DataLoader, as vqa_amd.train does for the other splits.

As vqa_model/test.py:29-110, 172-205: build the model, load ``ModelBestRSVQA_<model>.pth`` from the model folder
(``../data/dump-models/<dataset>`` upstream; ``--models-dir`` here), run one no-grad pass in eval mode over the test
set and print the test loss, the accuracy of every non-empty question type, OA and AA (AA divides by the constant 4,
test.py:108).  The argmax and the per-type counters run on the device (vqa_argmax_accuracy); only the eight integers
come back to the host.
"""
from __future__ import annotations

import argparse
import os

import torch

from .metrics import AccuracyMeter, HR_TYPES, LR_TYPES
from .train import SynthVQADataset, add_data_arguments, evaluate, model_and_geometry, real_data


def test(model, test_dataset, args, Dataset='HR', out=print):
    """vqa_model/test.py:29-110.  Returns the result dictionary of AccuracyMeter.result() plus 'loss'."""
    types = LR_TYPES if Dataset == 'LR' else HR_TYPES
    meter = AccuracyMeter(types, device=next(model.parameters()).device)
    res = evaluate(model, test_dataset, meter)
    out('test loss: %.3f' % res['loss'])
    out('Per category accuracy')
    for type_str in types:                          # only the types that occurred, test.py:100-102
        if type_str in res['acc']:
            out(f"{type_str}: {res['acc'][type_str]}")
    out('OA: %.3f' % res['OA'])
    out('AA: %.3f' % res['AA'])
    return res


def main(argv=None, **model_kw):
    parser = argparse.ArgumentParser(description='MI355X VQA evaluation')
    parser.add_argument('--bs', type=int, default=70)
    parser.add_argument('--gpu', type=int, default=0)
    parser.add_argument('--lr', type=float, default=0.00001)          # accepted and unused, as upstream
    parser.add_argument('--model', type=str, required=True)
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--models-dir', type=str, default=None,
                        help='folder holding ModelBestRSVQA_<model>.pth (reference: ../data/dump-models/<dataset>)')
    parser.add_argument('--batches', type=int, default=4, help='SYNTH: number of test batches')
    add_data_arguments(parser)
    args = parser.parse_args(argv)
    if args.dataset not in ('SYNTH', 'LR', 'HR'):
        raise SystemExit('--dataset must be SYNTH, LR or HR')
    dev = torch.device(f'cuda:{args.gpu}')
    torch.cuda.set_device(dev)
    if args.dataset == 'SYNTH':
        model, geo = model_and_geometry(args, dev, **model_kw)
        test_ds = SynthVQADataset(args.batches, args.bs, geo['R'], geo['Dv'], geo['L'], geo['Vq'], geo['C'], dev, seed=44,
                                  stage_maps=geo['stage_maps'])
    else:
        model, (test_ds,), _, _ = real_data(args, dev, ('test',), feature_dtype=model_kw.get('dtype', torch.bfloat16),
                                            **model_kw)
    folder = args.models_dir or f'../data/dump-models/{args.dataset}'
    path = os.path.join(folder, f'ModelBestRSVQA_{args.model}.pth')
    model.load_state_dict(torch.load(path, map_location=dev, weights_only=True))
    return test(model, test_ds, args, args.dataset)


if __name__ == '__main__':
    main()
