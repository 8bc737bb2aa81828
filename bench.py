"""Headline benchmark: train notes/sec of the base feature extractor (encoder + label attention + LDAM, fwd+bwd+Adam).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is what train_base.py:103-113 does per minibatch: zero_grad, model.forward (train mode: both dropouts on),
loss.mean().backward(), Adam(lr 1e-3, wd 1e-5).step().  Inputs are resident in HBM before the timed region.
Workload at N=1: BASELINE.json configs[1] (B=16, T=2500, E=H=100, L=9000, N=12000 nodes, V=50001, F=200, k=10, C=2).
At N>1 the label dimension is sharded L/N per rank and the global batch grows to 16*N (each rank encodes 16 notes and
attends 16*N notes x L/N labels: per-GPU work is fixed -> "weak" scaling).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel: label_attn_bwd, timed with
events on the launch stream inside the timed region; algorithmic FLOPs as DESIGN.md states them) and, at N=1,
`cpu_baseline` (the numpy oracle timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def workload(args, world):
    from tests.golden import synth
    B = args.batch_per_gpu * world
    case = synth.synth_base(seed=1234, B=B, T=args.T, E=args.E, H=args.E, L=args.L, N=args.N, Td=30, V=args.V, F=200,
                            labels_per_note=15, emb_scale=0.1, zipf=True)
    return case


def algorithmic_flops(d, B, L_local):
    """Forward FLOPs of the two attention contractions for B notes x L_local labels; backward is
    credited 2x forward (the S recompute is not credited)."""
    Tp = d['T'] - d['k'] + 1
    return 2.0 * B * L_local * Tp * (d['E'] + d['F'])


def step_flops(d, B):
    Tp = d['T'] - d['k'] + 1
    E, F, L, D, N, H = d['E'], d['F'], d['L'], d['D'], d['N'], d['H']
    fwd = (2.0 * B * Tp * F * E * d['k'] + 2.0 * B * Tp * F * E + 2.0 * B * L * Tp * (E + F) + 2.0 * B * L * F * D
           + 2.0 * B * L * D + 2.0 * 2 * N * 3 * H * (E + H))
    return 3.0 * fwd


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this script).  FETCH_SIZE is doubled as MI355X_MICROARCH.md
    prescribes for gfx950; returns None when no measurement is committed (bench.py cannot run the profiler itself)."""
    fn = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(fn) as f:
            d = json.load(f)[kernel]
        return (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


class KernelTimer:
    """Wraps HipOps.attn_fwd / attn_bwd with event pairs on the launch stream (torch's current stream)."""

    def __init__(self, ops):
        self.ops = ops
        self.events = {'attn_fwd': [], 'attn_bwd': []}
        self.enabled = False
        ops.attn_fwd = self._wrap('attn_fwd', ops.attn_fwd)
        ops.attn_bwd_main = self._wrap('attn_bwd', ops.attn_bwd_main)  # the label_attn_bwd launch alone

    def _wrap(self, name, fn):
        def inner(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.events[name].append((e0, e1))
            return r
        return inner

    def mean_ms(self, name):
        ev = self.events[name]
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None


def cpu_baseline(case, args):
    """Oracle (numpy restatement) timed on the host cores on a bounded sample of the same workload: the 16 notes of the
    N=1 batch, fwd+bwd over all labels incl. the label encoder, three passes (~10-15 s on the box's 16-core share)."""
    from oracle import base_model as om
    d = case['dims']
    nb, reps = min(16, d['B']), 3
    margin = om.class_margin_from_count(case['target_count'])
    x, y, mask = case['x'][:nb], case['y'][:nb], case['mask'][:nb]
    t0 = time.perf_counter()
    for _ in range(reps):
        om.base_forward_backward(case['params'], case['bufs'], x, y, mask, d['L'], margin, 2.0, drop_seed=1234, l_chunk=1500)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    return {'value': nb * reps / dt, 'unit': 'notes/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} passes over {nb} notes of the same workload (T={d["T"]}, L={d["L"]}, N={d["N"]}), fwd+bwd incl. '
                      f'label encoder (no optimizer step), numpy fp32 oracle, BLAS threads={cores}, {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=16)
    ap.add_argument('--T', type=int, default=2500)
    ap.add_argument('--E', type=int, default=100)
    ap.add_argument('--L', type=int, default=9000)
    ap.add_argument('--N', type=int, default=12000)
    ap.add_argument('--V', type=int, default=50001)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd+bwd only (diagnostic, not the metric)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # GZSL_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices);
    # the measured configuration is always one rank per GPU over RCCL ('nccl' IS RCCL on ROCm)
    backend = os.environ.get('GZSL_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == 'nccl' else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    # GZSL_BENCH_FORCE_DIST=1 (under torchrun --nproc-per-node 1) takes the distributed code path with ONE rank: a
    # rehearsal of the RCCL calls on a one-GPU box
    dist_on = world > 1 or os.environ.get('GZSL_BENCH_FORCE_DIST') == '1'
    if dist_on:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from gzsl_text_amd.modules.common_module import get_optimizer
    from tests import helpers

    case = workload(args, world)
    d = case['dims']
    model = helpers.build_model(case, device=dev)
    model.train()
    model.set_dropout_seed(1234)
    if dist_on:
        model.enable_label_sharding()
    x, y, mask = helpers.batch_tensors(case, dev)
    opt = get_optimizer(1e-3, model, weight_decay=1e-5)
    timer = KernelTimer(model._engine().ops)

    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model(x, y, mask)
        loss.mean().backward()
        if not args.no_optimizer:
            opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    trace = os.environ.get('GZSL_BENCH_TRACE') == '1'

    def note(msg):
        if trace:
            print(f'[bench rank {rank}] {msg}', file=sys.stderr, flush=True)

    if trace:
        import faulthandler
        faulthandler.dump_traceback_later(45, repeat=False, file=sys.stderr)
    note('model built')
    for i in range(args.warmup):
        loss = step()
        note(f'warmup step {i} issued')
    fence()
    note('warmup done')
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss)), 'loss is not finite'

    B = d['B']
    L_local = d['L'] // world + (1 if rank < d['L'] % world else 0)
    bwd_ms, fwd_ms = timer.mean_ms('attn_bwd'), timer.mean_ms('attn_fwd')
    fl_fwd = algorithmic_flops(d, B, L_local)
    ach = 2 * fl_fwd / (bwd_ms * 1e-3) / 1e12
    out = {
        'metric': 'train notes/sec (enc+label-attn+LDAM fwd/bwd)', 'value': B * args.steps / dt, 'unit': 'notes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'BASELINE.json configs[1]: base model fwd+bwd+Adam, B={B} (16/GPU), T={d["T"]}, '
                               f'E=H={d["E"]}, F=200, k=10, L={d["L"]} ({L_local}/GPU), N={d["N"]}, V={d["V"]}, '
                               f'train mode (emb dropout .5, attn dropout .2), LDAM C=2, graph_encoder=gate',
                   'global_batch': B, 'seq_len': d['T'], 'parallelism': f'label-shard x{world}' if world > 1 else 'single'},
        'roofline': {'bound': 'mfma', 'kernel': 'label_attn_bwd_kernel', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TFLOPS,
                     'traffic': measured_traffic('label_attn_bwd_kernel') if world == 1 else None,
                     'traffic_note': 'bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from profiles/r01_pmc_traffic.json '
                                     '(rocprofv3 PMC passes of this command at N=1; not re-measured in this run)',
                     'ms_per_launch': bwd_ms,
                     'algorithmic_flops_per_launch': 2 * fl_fwd,
                     'also': {'label_attn_fwd_kernel': {'ms_per_launch': fwd_ms,
                                                        'achieved': fl_fwd / (fwd_ms * 1e-3) / 1e12,
                                                        'frac': fl_fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                              'whole_step': {'algorithmic_flops': step_flops(d, B) / world,
                                             'achieved': step_flops(d, B) / world / (dt / args.steps) / 1e12,
                                             'frac': step_flops(d, B) / world / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS}}},
        'loss': float(loss.detach()),
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(case, args)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
