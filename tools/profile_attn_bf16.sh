#!/bin/bash
# Kernel trace + SQ counters of the bf16x6 attention kernels alone at the bench shape (tools/time_attn.py).
#   usage: bash tools/profile_attn_bf16.sh tag
set -u
tag=${1:-bf}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp GZSL_ATTN_BF16X6=1 N_IT=3
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.trace -- python3 tools/time_attn.py > $out/trace.log 2>&1 || exit 1
python3 tools/summarize_profile.py /tmp/$tag.trace $out/attn "rocprofv3 --kernel-trace --stats -- GZSL_ATTN_BF16X6=1 python3 tools/time_attn.py" 2 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/$tag.sq -- python3 tools/time_attn.py > $out/pmc_sq.log 2>&1 || exit 1
python3 tools/pmc_summary.py $(ls /tmp/$tag.sq/*/*.db | head -1) > $out/pmc_sq.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d /tmp/$tag.sq2 -- python3 tools/time_attn.py > $out/pmc_sq2.log 2>&1 || exit 1
python3 tools/pmc_summary.py $(ls /tmp/$tag.sq2/*/*.db | head -1) > $out/pmc_sq2.txt 2>&1 || exit 1
head -12 $out/attn_summary.md; grep -A1 "bf16" $out/pmc_sq.txt; grep -A1 "bf16" $out/pmc_sq2.txt
