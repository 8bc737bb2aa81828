"""Headline benchmark: train notes/sec of the base feature extractor (encoder + label attention + LDAM, fwd+bwd+Adam).

    python bench.py --gpus 1 --steps 600 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is what train_base.py:103-113 does per minibatch: zero_grad, model.forward (train mode: both dropouts on),
loss.mean().backward(), Adam(lr 1e-3, wd 1e-5).step().  Inputs are resident in HBM before the timed region.

--config cfg2 (default): BASELINE.json configs[1] (B=16, T=2500, E=H=100, L=9000, N=12000 nodes, V=50001, F=200, k=10,
    C=2).  At N>1 the label dimension is sharded L/N per rank and the global batch grows to 16*N (each rank encodes 16
    notes and attends 16*N notes x L/N labels: per-GPU work is fixed -> "weak" scaling).
--config cfg3: BASELINE.json configs[2] (T=4000, global batch fixed at 16, L=9000 sharded L/N per rank -> "strong").
--config real: cfg2 with E=H=200 (the width the reference's drivers build from the shipped resources).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel (label_attn_bwd), timed with events on the launch stream INSIDE the timed region;
                .also: forward attention, whole step, at N=1 the same numbers at E=H=200 (`real_e200`, separate short
                pass), achieved GB/s of every HBM-bound kernel of the step (`hbm`, separate short pass; algorithmic
                bytes as SURVEY.md 8d / DESIGN.md 4 state them, peak 8 TB/s) and forward-only eval scoring of the
                same batch (`eval_forward`)
                and the WGAN-GP step of configs[3] (`gan_step`: Bg = 8 as a HIP graph, 4096, 16384; samples/s, launches,
                executed-GEMM-FLOP fraction of the fp32 matrix peak, numpy-oracle CPU leg at Bg = 4096)
  cpu_baseline  (N=1) the numpy oracle timed on the host cores on a bounded sample of the same workload
  comm          (N>1) per collective: ms per step the launching stream sat blocked on it (exposed communication) and
                bytes per call; plus ms/step of a short pass with the own-notes-first ordering switched the other way.
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
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a float4 copy reaches)

CONFIGS = {
    'cfg2': dict(T=2500, E=100, weak=True, name='BASELINE.json configs[1]'),
    'cfg3': dict(T=4000, E=100, weak=False, name='BASELINE.json configs[2]'),
    'real': dict(T=2500, E=200, weak=True, name='BASELINE.json configs[1] at the drivers\' real width E=H=200'),
}


def workload(args, world, E=None):
    from gzsl_text_amd.utils import synth
    B = args.batch_per_gpu * world if args.weak else args.batch_per_gpu
    E = args.E if E is None else E
    return synth.synth_base(seed=1234, B=B, T=args.T, E=E, H=E, L=args.L, N=args.N, Td=30, V=args.V, F=200,
                            labels_per_note=15, emb_scale=0.1, zipf=True)


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
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json,
    newest round first: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this script).  FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950; None when no measurement is committed (bench.py cannot run the profiler)."""
    for rnd in ('r03', 'r02', 'r01'):
        fn = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_traffic.json')
        try:
            with open(fn) as f:
                d = json.load(f)[kernel]
            return (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0, os.path.basename(fn)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


class KernelTimer:
    """Wraps HipOps methods with event pairs on the launch stream (torch's current stream at the call)."""

    def __init__(self, ops, names):
        self.ops = ops
        self.events = {n: [] for n in names}
        self.enabled = False
        for n in names:
            setattr(ops, n, self._wrap(n, getattr(ops, n)))

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

    def total_ms(self, name, steps):
        ev = self.events[name]
        return float(np.sum([a.elapsed_time(b) for a, b in ev])) / steps if ev else None


def hbm_bytes(d, B, nnz):
    """Algorithmic bytes PER STEP of the HBM-bound kernels (SURVEY.md 8d formulas; atomics counted as the bytes added),
    keyed by the HipOps method that launches them; `calls` = launches per step the bytes are summed over."""
    T, E, F, L, D, N, H, V, Td = d['T'], d['E'], d['F'], d['L'], d['D'], d['N'], d['H'], d['V'], d['Td']
    BT = B * T
    return {
        'emb_gather': BT * (8 + 4 + 2 * E * 4),
        'emb_scatter_add_': BT * (8 + 4) + 2 * BT * E * 4,
        'label_avg': N * Td * (8 + 4) + N * Td * E * 4 + N * E * 4,
        'label_avg_bwd_': N * Td * (8 + 4) + N * E * 4 + N * Td * E * 4,
        'spmm': 4 * (nnz * (8 + H * 4) + 2 * N * H * 4),                   # 2 forward + 2 adjoint launches
        'gru_gate_fwd': 2 * 8 * N * H * 4,                                 # reads gi, gh [N,3H], h; writes hn
        'gru_gate_bwd': 2 * 15 * N * H * 4,                                # + dhn in, dgi, dgh, dh out
        'logits_fwd': B * L * D * 4 + L * D * 4 + B * L * 4,
        'logits_bwd': 2 * B * L * D * 4 + 2 * L * D * 4 + B * L * 4,
        'logits_bwd_delta': 2 * B * L * D * 4 + 2 * L * D * 4 + 2 * B * L * 4 + D * 4,  # + delta out (replaces attn_delta)
        'attn_delta': 2 * B * L * F * 4 + B * L * 4,
        'ldam_bce': 3 * B * L * 4 + L * 4,
    }


def hbm_pass(model, step, timer_ops, d, B, steps=5):
    """Short pass after the timed region: event-time every HBM-bound kernel of the step and report achieved GB/s."""
    names = ['emb_gather', 'emb_scatter_add_', 'label_avg', 'label_avg_bwd_', 'spmm', 'gru_gate_fwd', 'gru_gate_bwd',
             'logits_fwd', 'logits_bwd', 'logits_bwd_delta', 'ldam_bce']
    ops = timer_ops
    t = KernelTimer(ops, names)
    # attn_delta is launched inside HipOps.attn_bwd: time it through the C-ABI call table
    from gzsl_text_amd import ops_hip
    orig_call = ops_hip.call
    delta_ev = []

    def call(name, *a):
        if name != 'gzsl_attn_delta' or not t.enabled:
            return orig_call(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        delta_ev.append((e0, e1))
    ops_hip.call = call
    side = os.environ.get('GZSL_NO_SIDE_STREAM')
    os.environ['GZSL_NO_SIDE_STREAM'] = '1'  # one stream: a kernel's events then bracket that kernel alone
    try:
        t.enabled = True
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        t.enabled = False
    finally:
        ops_hip.call = orig_call
        if side is None:
            del os.environ['GZSL_NO_SIDE_STREAM']
        else:
            os.environ['GZSL_NO_SIDE_STREAM'] = side
    nnz = int(model._graph_buffers().csr[1].numel())
    by = hbm_bytes(d, B, nnz)
    out = {}
    for n in names + ['attn_delta']:
        ms = (float(np.sum([a.elapsed_time(b) for a, b in delta_ev])) / steps) if n == 'attn_delta' else t.total_ms(n, steps)
        if not ms:
            continue
        gbs = by[n] / (ms * 1e-3) / 1e9
        out[n] = {'bytes_per_step': int(by[n]), 'ms_per_step': ms, 'GBps': gbs, 'frac_of_8TBps': gbs / PEAK_HBM_GBS}
    return out


def cpu_baseline(case, args):
    """Oracle (numpy restatement) timed on the host cores on a bounded sample of the same workload: the 16 notes of the
    N=1 batch, fwd+bwd over all labels incl. the label encoder, three passes (~10-15 s on the box's 16-core share).
    The optimizer step of the GPU leg is NOT part of this leg (elementwise over 5.3 M parameters: noise here)."""
    from oracle import base_model as om
    d = case['dims']
    nb, reps = min(16, d['B']), 3
    margin = om.class_margin_from_count(case['target_count'])
    x, y, mask = case['x'][:nb], case['y'][:nb], case['mask'][:nb]
    t0 = time.perf_counter()
    for _ in range(reps):
        om.base_forward_backward(case['params'], case['bufs'], x, y, mask, d['L'], margin, 2.0, drop_seed=1234, l_chunk=1500)
    dt = time.perf_counter() - t0
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    try:  # the thread count numpy's BLAS actually runs with
        from threadpoolctl import threadpool_info
        blas = [p['num_threads'] for p in threadpool_info() if p.get('user_api') == 'blas']
        if blas:
            threads = max(blas)
    except Exception:
        pass
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        physical = None
    return {'value': nb * reps / dt, 'unit': 'notes/s', 'cores': threads, 'kind': 'port',
            'physical_cores_on_host': physical, 'logical_cpus_on_host': os.cpu_count(),
            'sample': f'{reps} passes over {nb} notes of the same workload (T={d["T"]}, L={d["L"]}, N={d["N"]}), fwd+bwd incl. '
                      f'label encoder (no optimizer step), numpy fp32 oracle, BLAS threads={threads}, {dt:.1f} s'}


def attention_lines(timer, d, B, L_local):
    bwd_ms, fwd_ms = timer.mean_ms('attn_bwd_main'), timer.mean_ms('attn_fwd')
    fl = algorithmic_flops(d, B, L_local)
    return bwd_ms, fwd_ms, fl


def eval_forward_pass(model, x, y, mask, d, steps=10):
    """Forward-only scoring of the same batch (train_base.py:118-140's loop body; BASELINE configs[4] "full-label eval
    scoring" on one GPU): eval mode, no dropout, torch.no_grad()."""
    model.eval()
    try:
        with torch.no_grad():
            for _ in range(2):
                model(x, y, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model(x, y, mask)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
    finally:
        model.train()
    fwd = step_flops(d, d['B']) / 3.0
    return {'config': f'eval mode, no_grad, same batch; {steps} passes', 'ms_per_batch': dt * 1e3, 'notes_per_s': d['B'] / dt,
            'frac_of_fp32_matrix_peak': fwd / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}


PEAK_BF16_MFMA_TFLOPS = 2516.8  # MI355X_MICROARCH.md: dense bf16 matrix peak; six bf16 products per fp32 product


def bf16x6_step_pass(model, step, ops, timer, x, y, mask, d, B, steps=20):
    """OPT-IN switch, reported beside the headline and never part of it (`value` / `dtype: f32` stay on the exact-fp32
    MFMA kernels): both attention kernels with each fp32 product formed as six bf16 MFMAs (csrc/attn_bf16.hip,
    csrc/attn_bwd_bf16.hip; inputs, outputs and accumulation stay fp32; the kernels pass the fp32 kernels' parity cases at
    the same tolerances).  Same step, same batch, switch on.  `frac` = fp32-equivalent FLOP rate over 2516.8 / 6 TFLOP/s."""
    assert not ops.attn_bf16x6
    ops.attn_bf16x6 = True
    try:
        timer.events['attn_fwd'].clear()
        timer.events['attn_bwd_main'].clear()
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        timer.enabled = False
        fwd_ms, bwd_ms = timer.total_ms('attn_fwd', steps), timer.total_ms('attn_bwd_main', steps)
        ev = eval_forward_pass(model, x, y, mask, d)
    finally:
        timer.enabled = False
        ops.attn_bf16x6 = False
        timer.events['attn_fwd'].clear()
        timer.events['attn_bwd_main'].clear()
    fl = algorithmic_flops(d, B, d['L'])
    peak = PEAK_BF16_MFMA_TFLOPS / 6.0
    return {'config': f'GZSL_ATTN_BF16X6=1 (forward and backward attention as six bf16 MFMAs per fp32 product); {steps} steps',
            'ms_per_step': dt * 1e3, 'notes_per_s': B / dt, 'loss': float(loss.detach()),
            'peak_fp32_equivalent_tflops': peak,
            'label_attn_fwd': {'ms_per_call_incl_repack': fwd_ms, 'fp32_equivalent_tflops': fl / (fwd_ms * 1e-3) / 1e12,
                               'frac': fl / (fwd_ms * 1e-3) / 1e12 / peak},
            'label_attn_bwd': {'ms_per_call_incl_repack_and_dq_reduce': bwd_ms,
                               'fp32_equivalent_tflops': 2 * fl / (bwd_ms * 1e-3) / 1e12,
                               'frac': 2 * fl / (bwd_ms * 1e-3) / 1e12 / peak},
            'whole_step_frac_of_fp32_matrix_peak': step_flops(d, B) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'eval_forward': {'ms_per_batch': ev['ms_per_batch'], 'notes_per_s': ev['notes_per_s']}}


def real_e200_pass(args, dev, steps=5):
    """The same step at E=H=200 (what the reference's drivers build from the shipped resources), N=1, short pass."""
    from gzsl_text_amd.modules.common_module import get_optimizer
    from gzsl_text_amd.utils import synth
    case = workload(args, 1, E=200)
    d = case['dims']
    model = synth.build_model(case, device=dev)
    model.train()
    model.set_dropout_seed(1234)
    x, y, mask = synth.batch_tensors(case, dev)
    opt = get_optimizer(1e-3, model, weight_decay=1e-5)
    timer = KernelTimer(model._engine().ops, ['attn_fwd', 'attn_bwd_main'])

    def step():
        opt.zero_grad(set_to_none=True)
        _, loss = model(x, y, mask)
        loss.mean().backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    bwd_ms, fwd_ms, fl = attention_lines(timer, d, d['B'], d['L'])
    return {'config': f'E=H=200, D=400, otherwise the N=1 workload; {steps} steps', 'ms_per_step': dt * 1e3,
            'notes_per_s': d['B'] / dt,
            'label_attn_bwd_kernel': {'ms_per_launch': bwd_ms, 'achieved': 2 * fl / (bwd_ms * 1e-3) / 1e12,
                                      'frac': 2 * fl / (bwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
            'label_attn_fwd_kernel': {'ms_per_launch': fwd_ms, 'achieved': fl / (fwd_ms * 1e-3) / 1e12,
                                      'frac': fl / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
            'whole_step_frac': step_flops(d, d['B']) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}


GAN_REF_FLOPS_PER_SAMPLE = 7.8e7  # the REFERENCE's count (incl. the double backward of its gradient penalty)


def gan_cpu_baseline(case, n_iters, reg_ratio):
    """oracle/gan.py (numpy) on the host cores: ONE full WGAN-GP step (5 critic iterations + Adam, 1 generator iteration
    + Adam) on the same Bg = 4096 batch the GPU leg runs."""
    from oracle import gan as og
    d = {k: v.copy() for k, v in case['d'].items()}
    gen = {k: v.copy() for k, v in case['g'].items()}
    kwp = {k: v.copy() for k, v in case['kw'].items()}
    dstate = {k: {} for k in d}
    t0 = time.perf_counter()
    for it in range(n_iters):
        I = case['iters'][it]
        _, _, grads, _ = og.critic_iteration(d, gen, I['real'], I['lab'], I['noise'], I['alpha'], I['sim'])
        for k in d:
            d[k] = og.adam_step(d[k], grads[k], dstate[k], 1e-4)
    I = case['iters'][n_iters]
    _, _, gg, kg, _ = og.generator_iteration(d, gen, kwp, I['lab'], I['noise'], I['sim'], case['kw_idx'], case['kw_mask'],
                                            reg_ratio)
    for k in gen:
        gen[k] = og.adam_step(gen[k], gg[k], {}, 1e-4)
    for k in kg:
        kwp[k] = og.adam_step(kwp[k], kg[k], {}, 1e-4)
    dt = time.perf_counter() - t0
    Bg = case['dims']['Bg']
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    return {'value': Bg / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': f'1 full step (5 critic + 1 generator iteration, Adam) at Bg={Bg}, numpy fp32 oracle, {dt:.2f} s'}


def gan_step_pass(dev, cpu_leg=True, bgs=(8, 4096, 16384), steps=20):
    """The WGAN-GP step of BASELINE configs[3] (train_gan.py:223-308: 5 critic iterations + 1 generator iteration, Adam
    lr 1e-4 betas .5/.999, keyword head top_k=20 Vk=4096 reg_ratio 0.01, add-zero similarity weights; D=400, Ls=noise=400,
    ngh=ndh=800) on the whole-iteration engine (gan_engine.WganGpStep.full_step), inputs resident in HBM.  Bg=8 is the
    drivers' default batch (config.py:42), replayed as ONE HIP graph; 4096 / 16384 are "large batch".  `frac` divides the
    GEMM FLOPs the engine EXECUTES (2MNK summed over its gzsl_gemm launches of one step, counted at the C ABI) by the
    step time and the fp32 matrix peak; the reference-count figure (7.8e7 FLOP/sample, which includes a double backward
    the closed-form penalty never runs) is stated beside it, not instead."""
    from gzsl_text_amd import ops_hip
    from gzsl_text_amd.gan_engine import WganGpStep, graphed_step
    from gzsl_text_amd.ops_hip import HipOps
    from gzsl_text_amd.utils import synth
    n_iters, reg_ratio = 5, 0.01
    out = {}
    for Bg in bgs:
        case = synth.synth_gan(seed=7, Bg=Bg, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=4096, top_k=20, n_iters=n_iters)

        def dv(dd):
            return {k: torch.from_numpy(v).to(dev) for k, v in dd.items()}

        graph = Bg <= 32
        step = WganGpStep(HipOps(), dv(case['g']), dv(case['d']), dv(case['kw']), lr=1e-4, critic_iters=n_iters,
                          reg_ratio=reg_ratio, capturable=graph)
        its = [{k: torch.from_numpy(v).to(dev) for k, v in it.items()} for it in case['iters']]
        kw_idx, kw_mask = torch.from_numpy(case['kw_idx']).to(dev), torch.from_numpy(case['kw_mask']).to(dev)
        cb = [(it['real'], it['lab'], it['noise'], it['alpha'], it['sim']) for it in its[:n_iters]]
        gb = (its[n_iters]['lab'], its[n_iters]['noise'], its[n_iters]['sim'], kw_idx, kw_mask)

        def one():
            return step.full_step(cb, gb)[1][0]

        one()  # lazily created state
        # count one eager step at the C ABI: launches, and the FLOPs of its GEMMs
        orig_call, cnt = ops_hip.call, {'launches': 0, 'gemm_flops': 0.0, 'gemms': 0}

        def call(name, *a):
            cnt['launches'] += 1
            if name == 'gzsl_gemm':
                g = a[0]
                cnt['gemms'] += 1
                cnt['gemm_flops'] += 2.0 * g.M * g.N * g.K
            return orig_call(name, *a)
        ops_hip.call = call
        try:
            one()
        finally:
            ops_hip.call = orig_call
        fn = graphed_step(one) if graph else one
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        n = steps * (10 if graph else 1)
        t0 = time.perf_counter()
        for _ in range(n):
            g_cost = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert bool(torch.isfinite(g_cost))
        line = {'samples_per_s': Bg / dt, 'ms_per_step': dt * 1e3, 'hip_graph': graph, 'steps': n,
                'c_abi_launches_per_step': cnt['launches'], 'gemm_launches_per_step': cnt['gemms'],
                'executed_gemm_flops_per_step': cnt['gemm_flops'],
                'achieved': cnt['gemm_flops'] / dt / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': cnt['gemm_flops'] / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                'reference_count_flops_per_step': GAN_REF_FLOPS_PER_SAMPLE * Bg,
                'frac_on_reference_count': GAN_REF_FLOPS_PER_SAMPLE * Bg / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        if cpu_leg and Bg == 4096:
            line['cpu_baseline'] = gan_cpu_baseline(case, n_iters, reg_ratio)
        out[f'Bg{Bg}'] = line
        del step, its, cb, gb
        torch.cuda.empty_cache()
    out['config'] = ('BASELINE.json configs[3]: one WGAN-GP step = 5 critic + 1 generator iteration with Adam, keyword head '
                     '(top_k=20, Vk=4096, reg_ratio=0.01), add-zero sim weights, D=400, Ls=noise=400, ngh=ndh=800; f32')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600)   # ~5 s of GPU work at N=1: longer than the driver's smi sampling period
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default=os.environ.get('GZSL_BENCH_CONFIG', 'cfg2'))
    ap.add_argument('--batch-per-gpu', type=int, default=16, help='notes per GPU (weak configs) / global batch (cfg3)')
    ap.add_argument('--T', type=int, default=None)
    ap.add_argument('--E', type=int, default=None)
    ap.add_argument('--L', type=int, default=9000)
    ap.add_argument('--N', type=int, default=12000)
    ap.add_argument('--V', type=int, default=50001)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the short post-passes (E=200 line, HBM kernels, comm A/B)')
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd+bwd only (diagnostic, not the metric)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.T = cfg['T'] if args.T is None else args.T
    args.E = cfg['E'] if args.E is None else args.E
    args.weak = cfg['weak']

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
    from gzsl_text_amd.utils import synth

    case = workload(args, world)
    d = case['dims']
    model = synth.build_model(case, device=dev)
    model.train()
    model.set_dropout_seed(1234)
    if dist_on:
        model.enable_label_sharding()
    x, y, mask = synth.batch_tensors(case, dev)
    opt = get_optimizer(1e-3, model, weight_decay=1e-5)
    ops = model._engine().ops
    ops.attn_bf16x6 = False  # the metric is the exact-fp32 MFMA path whatever the environment says (see bf16x6_step_pass)
    timer = KernelTimer(ops, ['attn_fwd', 'attn_bwd_main'])  # attn_bwd_main = the label_attn_bwd launch alone

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
    if dist_on:
        model._shard.start_profile()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    # host time to ISSUE a step into empty queues (median of 5; with steps queued behind each other the HIP runtime
    # throttles the issuing thread to the GPU's pace, which says nothing about the boundary's own cost)
    issue = []
    for _ in range(0 if dist_on else 5):  # (N = 1 only: the collectives' exposed-time profile counts the timed steps)
        t1 = time.perf_counter()
        loss = step()
        issue.append(time.perf_counter() - t1)
        fence()
    dt_issue = sorted(issue)[2] if issue else None
    comm = None
    if dist_on:
        comm = model._shard.profile_summary(args.steps)
        model._shard.prof = None
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss)), 'loss is not finite'

    B = d['B']
    L_local = d['L'] // world + (1 if rank < d['L'] % world else 0)
    bwd_ms, fwd_ms, fl_fwd = attention_lines(timer, d, B, L_local)
    # (the own-notes-first ordering splits each attention launch into note blocks: sum the pieces of a step)
    n_bwd = len(timer.events['attn_bwd_main']) / max(args.steps, 1)
    fwd_ms *= len(timer.events['attn_fwd']) / max(args.steps, 1)
    ach = 2 * fl_fwd / (bwd_ms * n_bwd * 1e-3) / 1e12
    traffic, traffic_file = measured_traffic('label_attn_bwd_kernel') if (world == 1 and args.config == 'cfg2') else (None, None)
    per_gpu = f'{args.batch_per_gpu}/GPU' if args.weak else 'fixed'
    out = {
        'metric': 'train notes/sec (enc+label-attn+LDAM fwd/bwd)', 'value': B * args.steps / dt, 'unit': 'notes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'{cfg["name"]}: base model fwd+bwd+Adam, B={B} ({per_gpu}), T={d["T"]}, '
                               f'E=H={d["E"]}, F=200, k=10, L={d["L"]} ({L_local}/GPU), N={d["N"]}, V={d["V"]}, '
                               f'train mode (emb dropout .5, attn dropout .2), LDAM C=2, graph_encoder=gate',
                   'global_batch': B, 'seq_len': d['T'], 'parallelism': f'label-shard x{world}' if world > 1 else 'single'},
        'roofline': {'bound': 'mfma', 'kernel': 'label_attn_bwd_kernel', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
                     'traffic_note': f'bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from profiles/{traffic_file} '
                                     '(rocprofv3 PMC passes of this command at N=1; not re-measured in this run)',
                     'ms_per_launch': bwd_ms * n_bwd, 'launches_per_step': n_bwd,
                     'algorithmic_flops_per_launch': 2 * fl_fwd,
                     'also': {'label_attn_fwd_kernel': {'ms_per_launch': fwd_ms,
                                                        'achieved': fl_fwd / (fwd_ms * 1e-3) / 1e12,
                                                        'frac': fl_fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                              'whole_step': {'algorithmic_flops': step_flops(d, B) / world,
                                             'achieved': step_flops(d, B) / world / (dt / args.steps) / 1e12,
                                             'frac': step_flops(d, B) / world / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS}}},
        'loss': float(loss.detach()),
        # Boundary overhead: Python + ctypes + ATen time per step on the issuing thread (one step
        # into empty queues, median of 5, after the timed region).  Below ms_per_step the host runs ahead of the GPU and the
        # boundary costs nothing.
        'host_issue_ms_per_step': None if dt_issue is None else dt_issue * 1e3,
    }
    if dist_on:
        out['comm'] = {'world_size_seen_by_rccl': dist.get_world_size(), 'backend': dist.get_backend(),
                       'own_notes_first': bool(model._shard.own_first),
                       'collectives': comm,
                       # one line to read scaling off: what the launching stream waited on collectives, and what is left
                       'exposed_ms_per_step': sum(v['exposed_ms_per_step'] for v in comm.values()),
                       'ms_per_step_minus_exposed': dt / args.steps * 1e3 - sum(v['exposed_ms_per_step'] for v in comm.values()),
                       'rank0_attn_fwd_ms_per_step': fwd_ms, 'rank0_attn_bwd_ms_per_step': bwd_ms * n_bwd,
                       'note': 'exposed_ms_per_step = event-timed time the launching stream sat blocked on the collective '
                               '(rank 0); no scaling curve has been measured by the builder (one-GPU boxes only)'}
        if not args.no_extras and world > 1 and model._shard.batch_split(B) is not None:
            # A/B: the other ordering, a short pass (max over ranks).  Never allowed to take the headline line down.
            sh = model._shard
            default_order = sh.own_first
            try:
                sh.own_first = not default_order
                n_ab = max(3, min(20, args.steps))
                for _ in range(2):
                    step()
                fence()
                sh.start_profile()
                t1 = time.perf_counter()
                for _ in range(n_ab):
                    step()
                fence()
                dt_ab = time.perf_counter() - t1
                tt = torch.tensor([dt_ab], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                out['comm']['other_ordering'] = {'own_notes_first': bool(sh.own_first), 'steps': n_ab,
                                                 'ms_per_step': float(tt.item()) / n_ab * 1e3,
                                                 'collectives': sh.profile_summary(n_ab)}
            except Exception as exc:  # noqa: BLE001
                out['comm']['other_ordering'] = {'error': repr(exc)}
            finally:
                sh.prof = None
                sh.own_first = default_order
    if world == 1 and not args.no_extras:
        try:
            out['roofline']['also']['hbm'] = hbm_pass(model, step, ops, d, B)
            out['roofline']['also']['eval_forward'] = eval_forward_pass(model, x, y, mask, d)
            out['roofline']['also']['bf16x6_step'] = bf16x6_step_pass(model, step, ops, timer, x, y, mask, d, B)
            if args.config == 'cfg2' and args.E == 100:
                del model, opt
                torch.cuda.empty_cache()
                out['roofline']['also']['real_e200'] = real_e200_pass(args, dev)
                out['roofline']['also']['gan_step'] = gan_step_pass(dev, cpu_leg=not args.no_cpu_baseline)
        except Exception as exc:  # noqa: BLE001  (the post-passes never take the headline line down)
            out['roofline']['also']['extras_error'] = repr(exc)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(case, args)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
