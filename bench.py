"""Throughput of the VQA train step on N MI355X of one node.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + softmax-CE + backward + gradient all-reduce + Adam on one
synthetic VQA-v2-shaped batch per rank (256 samples, 14x14x2048 bf16 region
features, 14-token questions, 3000 answers; device-resident, several distinct
batches in rotation so that the 256 MiB Infinity Cache cannot hold the inputs).
Dropout is on (training mode).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'train samples/sec (whole node) at 1/2/4/8 MI355X; fused-attn HBM GB/s % peak'
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0    # MI355X dense bf16 MFMA peak


def synth_batches(n, B, R, Dv, L, Vq, C, device, seed, dtype):
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    for _ in range(n):
        feats = torch.randn((B, R, Dv), device=device, generator=g).clamp_min_(0).to(dtype)
        lens = torch.randint(3, L + 1, (B,), device=device, generator=g)
        tok = torch.randint(1, Vq + 1, (B, L), device=device, generator=g)
        pos = torch.arange(L, device=device)[None, :]
        tok = torch.where(pos < (lens[:, None] - 1), tok, torch.zeros_like(tok))
        ans = torch.randint(0, C, (B,), device=device, generator=g)
        out.append((feats, tok, ans))
    return out


def cpu_baseline(B, R, Dv, L, Vq, C, threads):
    """The CPU oracle (a port of the reference's arithmetic, fp32) timed on the host cores
    on a bounded sample: a few steps of fwd + loss + bwd + Adam at the per-GPU batch."""
    from oracle import vqa_ref
    torch.set_num_threads(threads)
    Bs = B
    p = vqa_ref.init_params(vq=Vq, C=C, E=620, Hq=2400, Dv=Dv, A=1200, F=1200, Hf=256, seed=0)
    names = vqa_ref.trainable_names(p)
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    b = vqa_ref.synthetic_batch(Bs, R, Dv, L, Vq, C, seed=42)
    seeds = [vqa_ref.site_seed(0, 1, s) for s in range(4)]

    def step(i):
        _, _, g = vqa_ref.loss_and_grads(p, b['feats'], b['tokens'], b['answers'], None, 1, drop_seeds=seeds)
        for k in names:
            vqa_ref.adam_step(p[k], g[k], m[k], v[k], i, 1e-5)

    for i in range(3):                        # warm-up (allocator, thread pool)
        step(1 + i)
    n, times = 10, []
    for i in range(n):
        t0 = time.perf_counter()
        step(4 + i)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[n // 2]
    return {'value': Bs / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': f'median of {n} timed steps (fwd+CE+bwd+Adam, fp32, dropout on) of the CPU oracle at the per-GPU '
                      f'batch {Bs}, same model widths and input shapes, after 3 warm-up steps'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch')
    ap.add_argument('--regions', type=int, default=196)
    ap.add_argument('--hops', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unfolded', action='store_true', help='skip the second timed loop (question encoder with nothing tabulated)')
    ap.add_argument('--unfolded-last', action='store_true', help='A/B: run the unfolded loop after the headline loop (cold clocks in the headline loop)')
    ap.add_argument('--no-fuse-norm', action='store_true', help='A/B: separate row-norm pass before the image projection')
    ap.add_argument('--no-fold-embedding', action='store_true',
                    help='A/B: gather embeddings and run the LSTM input projection every step instead of the per-token table')
    ap.add_argument('--no-fold-first-step', action='store_true', help='A/B: run the recurrent product of step 1 as a GEMM')
    ap.add_argument('--no-resident-whole', action='store_true',
                    help='A/B: sequence packing and the two tabulated steps as launches of their own before the resident launch')
    ap.add_argument('--no-resident-encoder', action='store_true',
                    help='A/B: one recurrent GEMM + one cell launch per step instead of the weight-stationary launch')
    ap.add_argument('--no-swap-wgrad', action='store_true', help='A/B: dW_va in the nn.Linear orientation')
    ap.add_argument('--no-merge-head', action='store_true',
                    help="A/B: the head's weight / bias gradients in a launch of their own instead of the attention backward's")
    ap.add_argument('--no-split-attn', action='store_true', help='A/B (batch < #CUs): one attention workgroup per sample')
    ap.add_argument('--no-pad-rows', action='store_true', help='A/B: projected features P / dP with unpadded 2400-byte rows')
    ap.add_argument('--rccl-single', action='store_true',
                    help='N=1 only: run the bucketed all-reduce path over a one-rank RCCL group (its cost without peers)')
    ap.add_argument('--event-stride', type=int, default=10,
                    help='HIP-event pairs go around every n-th launch of the roofline kernels (and every n-th step of the '
                         'gradient reducer) inside the timed region: an event record costs the stream about 10 us')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or args.rccl_single:
        from vqa_amd.parallel import reserve_compute_units_for_resident_kernels
        reserve_compute_units_for_resident_kernels()
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    elif args.rccl_single:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1, device_id=dev)
    if world > 1 or args.rccl_single:
        from vqa_amd.parallel import connect_collectives
        connect_collectives(dev)       # RCCL's peer connections are made here, not inside the first steps

    from vqa_amd.model import VQAAttnModel
    from vqa_amd.parallel import GradReducer

    B, R, Dv, L, Vq, C = args.batch, args.regions, 2048, 14, 10000, 3000
    model = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, hops=args.hops, seed=0).to(dev)
    model.train()
    model.dropout_rank = rank          # same weights on every rank, different dropout masks
    if args.no_fuse_norm:
        model.fuse_row_norm = False
    if args.no_fold_embedding:
        model.fold_embedding = False
    if args.no_resident_encoder:
        model.resident_encoder = False
    if args.no_resident_whole:
        model.resident_whole = False
    if args.no_fold_first_step:
        model.fold_first_step = False
    if args.no_swap_wgrad:
        model.swap_big_wgrad = False
    if args.no_pad_rows:
        model.pad_projection_rows = False
    if args.no_merge_head:
        model.merge_head_group = False
    if args.no_split_attn:
        model.split_small_batches = False
    reducer = GradReducer(force=args.rccl_single) if (world > 1 or args.rccl_single) else None
    nb = max(2, min(4, (320 << 20) // (B * R * Dv * 2) + 1))      # > 256 MiB of features in rotation
    batches = synth_batches(nb, B, R, Dv, L, Vq, C, dev, 42 + rank, torch.bfloat16)

    def step(i):
        f, t, a = batches[i % nb]
        return model.train_step(f, t, a, lr=1e-5, reducer=reducer)

    # the same step with nothing of the frozen question encoder tabulated (embedding gather + every LSTM product
    # computed each step): a second, shorter timed loop.  It runs FIRST, so the headline loop below starts on a chip
    # whose clocks have settled (after a cold start the step time keeps falling for about 20 steps) and its W warm-up
    # steps only have to build the tables and fill the caches.
    def timed_unfolded():
        folded = (model.fold_embedding, model.fold_first_step)
        model.fold_embedding = model.fold_first_step = False
        for i in range(5):
            step(i)
        model.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        n_unf = max(10, args.steps // 3)
        t0 = time.perf_counter()
        for i in range(n_unf):
            step(i)
        model.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e_unf = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e_unf], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e_unf = float(t)
        model.fold_embedding, model.fold_first_step = folded
        return {'value_unfolded': world * B * n_unf / e_unf, 'ms_per_step_unfolded': e_unf / n_unf * 1e3,
                'unfolded_steps': n_unf, 'unfolded_measured': 'before the headline loop'}

    unfolded = None
    if not args.no_unfolded and (model.fold_embedding or model.fold_first_step) and not args.unfolded_last:
        unfolded = timed_unfolded()

    for i in range(args.warmup):
        step(i)
    model.sync()                      # no deferred update of the warmup leaks into the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.kernel_events = {}
    # (short runs sample more densely: at least four timed launches of each kernel; a pair costs the step about 4 us)
    stride = max(1, min(args.event_stride, args.steps // 4))
    model.kernel_event_stride = stride
    if reducer is not None:
        reducer.timing = {}
        reducer.timing_stride = stride
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    model.sync(check=False)           # the K-th step's deferred last-bucket all-reduce + Adam is timed too
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.check_status()              # (the read-back of the kernels' status words is not part of a step)
    my_elapsed_local = elapsed
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    def avg_ms(name):
        ev = model.kernel_events.get(name, [])
        return (sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)) if ev else (0.0, 0)

    diag = None
    try:                                   # diagnostics must never cost the run its result line
        if world > 1 or args.rccl_single:
            # what the first multi-GPU run needs to explain itself: who took part, what was reduced, what the compute
            # stream waited for, how uneven the ranks were
            props = torch.cuda.get_device_properties(dev)
            # the three roofline kernels as THIS rank timed them: RCCL's kernels share the CUs with the products they run
            # beside, so scaling loss shows up here first
            me = {'rank': rank, 'local_rank': local, 'device': torch.cuda.current_device(), 'name': props.name,
                  'uuid': str(getattr(props, 'uuid', '')), 'ms_per_step': my_elapsed_local / args.steps * 1e3,
                  'kernel_ms': {'attn_fwd': avg_ms('attn_fwd')[0], 'proj': avg_ms('proj:0')[0],
                                'wgrad': avg_ms('wgrad:attention.0.linear_va.weight')[0]}}
            everyone = [None] * world
            dist.all_gather_object(everyone, me)
            diag = {'rccl_ranks': everyone,
                    'ms_per_step_min': min(e['ms_per_step'] for e in everyone),
                    'ms_per_step_max': max(e['ms_per_step'] for e in everyone),
                    'kernel_ms_max_over_ranks': {k: max(e['kernel_ms'][k] for e in everyone) for k in me['kernel_ms']},
                    # what is known about the collectives without extra output: the library version and any
                    # algorithm / protocol forced through the environment (RCCL picks them per message size otherwise)
                    'rccl': {'version': '.'.join(str(v) for v in torch.cuda.nccl.version()),
                             'env': {k: v for k, v in os.environ.items()
                                     if k.startswith(('NCCL_', 'RCCL_')) or k == 'HSA_ENABLE_IPC_MODE_LEGACY'}}}
        if reducer is not None and reducer.timing:
            launches = reducer.timing.get('launch', [])
            waits = reducer.timing.get('wait', [])
            timed_steps = max((args.steps + reducer.timing_stride - 1) // reducer.timing_stride, 1)
            per_step = max(len(launches) // timed_steps, 1)
            exposed = sum(a.elapsed_time(b) for a, b in waits) / timed_steps
            comm = {'timed_steps': timed_steps,
                    'collectives_per_step': per_step,
                    'allreduce_bytes_per_step': sum(nby for nby, _ in launches) // timed_steps,
                    'bytes_of_each_collective': [nby for nby, _ in launches[:per_step]],
                    'waits_per_step': len(waits) / timed_steps,
                    'exposed_wait_ms_per_step': exposed,
                    'note': 'exposed_wait = HIP-event time the compute stream spent in cross-stream waits for the '
                            'collectives (everything else of them ran under backward / the next forward)'}
            # span of a collective: from its launch to the end of the first wait recorded after it (an upper bound:
            # the wait sits one step later in the fully deferred schedule)
            spans = []
            wi = 0
            for _nby, ev in launches[:-per_step] if len(launches) > per_step else []:
                while wi < len(waits) and ev.elapsed_time(waits[wi][1]) <= 0:
                    wi += 1
                if wi < len(waits):
                    spans.append(ev.elapsed_time(waits[wi][1]))
            if spans:
                comm['launch_to_wait_end_ms_avg'] = sum(spans) / len(spans)
                comm['launch_to_wait_end_ms_max'] = max(spans)
            diag = dict(diag or {}, communication=comm)
    except Exception as exc:               # noqa: BLE001
        diag = dict(diag or {}, diagnostics_error=repr(exc))

    if rank == 0:
        k6_ms, k6_n = avg_ms('attn_fwd')
        # algorithmic bytes of the fused attention forward per sample and hop (bf16):
        # P (R*A) + V (R*Dv) + qa (A) + pooled (Dv) at 2 B, inv_norm (R) + p (R) at 4 B
        A = model.A
        k6_bytes = B * (2 * R * A + 2 * R * Dv + 2 * A + 4 * R + 4 * R + 2 * Dv)
        achieved = k6_bytes / (k6_ms * 1e-3) / 1e9 if k6_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'attn_fwd_traffic.json')
        if os.path.exists(tpath) and B == 256 and R == 196 and args.hops == 1:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        # the two MFMA-bound products: region projection (forward) and its weight gradient, 2 B R Dv A flop each
        gemm_flop = 2.0 * B * R * Dv * A
        gemms = {}
        for key, label in (('proj:0', 'image projection P = normalize(V) W_va^T + b (k_gemm_bf16_sp, row norms fused)'),
                           ('wgrad:attention.0.linear_va.weight', 'weight gradient dW_va = dP^T V (k_gemm_bf16_sp, split-K partial sums)')):
            ms, n = avg_ms(key)
            if n:
                tf = gemm_flop / (ms * 1e-3) / 1e12
                gemms[key.split(':')[0]] = {'kernel': label, 'flop_per_launch': gemm_flop, 'avg_launch_ms': ms, 'timed_launches': n,
                                            'achieved': tf, 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / MFMA_PEAK_TFLOPS}
        out = {
            'metric': METRIC, 'value': world * B * args.steps / elapsed, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'VQA-v2-shaped synthetic, {R} regions x {Dv} bf16 precomputed feats, '
                                   f'{L}-token questions, {C} answers, batch {B}/GPU, {args.hops} attention hop(s), '
                                   f'fwd+CE+bwd+all-reduce+Adam, dropout on',
                       'global_batch': world * B, 'parallelism': f'dp{world}',
                       'question_encoder': 'frozen, as the reference trains it; embedding x first LSTM input projection '
                                           + ('tabulated per token id when the weights are loaded' if model.fold_embedding
                                              else 'computed every step')
                                           + ('; step 1 recurrent product tabulated too' if model.fold_embedding
                                              and model.fold_first_step else '')
                                           + ('; steps 2.. in one launch with W_hh resident in registers'
                                              if getattr(model, '_xg_table_u', None) is not None else '')},
            'roofline': {'bound': 'hbm', 'kernel': 'k_attn_fwd_online (fused attention forward)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'avg_launch_ms': k6_ms, 'timed_launches': k6_n,
                         'algorithmic_bytes_per_launch': k6_bytes},
            'roofline_gemm': dict(gemms, bound='mfma', note='HIP events on the launch stream inside the timed region'),
            'loss': float(loss),
        }
        if diag:
            out['multi_gpu'] = diag
    if not args.no_unfolded and args.unfolded_last and (model.fold_embedding or model.fold_first_step):
        model.kernel_events = None
        unfolded = dict(timed_unfolded(), unfolded_measured='after the headline loop')
    if unfolded is not None and rank == 0:
        out.update(unfolded)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(B, R, Dv, L, Vq, C, threads=min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if world > 1 or args.rccl_single:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
