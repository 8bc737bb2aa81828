#!/bin/bash
# Diagnostic loop for csrc/attn_bwd_bf16.hip on the GPU box: parity tests, launch times, per-section clocks.
timeout -k 10 300 python -u -m pytest tests/test_gpu_attn_bwd_bf16.py -x -q --timeout 120 --timeout-method thread > gpurun_out/bwd_t.log 2>&1 || { tail -30 gpurun_out/bwd_t.log; exit 1; }
tail -2 gpurun_out/bwd_t.log
GZSL_ATTN_BF16X6=1 timeout -k 10 120 python tools/time_attn.py 2>&1 | grep -v amdgpu.ids || exit 1
timeout -k 10 120 python tools/prof_bwd_phases.py 2>&1 | grep -v amdgpu.ids
