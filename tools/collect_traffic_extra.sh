#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs) for the E=H=200 workload and for the bf16x6 attention kernels alone.
#   usage: bash tools/collect_traffic_extra.sh tag
set -u
tag=${1:-tr}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c -d /tmp/$tag.e200.$c -- python3 bench.py --config real --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/e200_$c.log 2>&1 || exit 1
  python3 tools/pmc_summary.py $(ls /tmp/$tag.e200.$c/*/*.db | head -1) > $out/e200_$c.txt 2>&1 || exit 1
done
export GZSL_ATTN_BF16X6=1 N_IT=3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c -d /tmp/$tag.bf.$c -- python3 tools/time_attn.py > $out/bf_$c.log 2>&1 || exit 1
  python3 tools/pmc_summary.py $(ls /tmp/$tag.bf.$c/*/*.db | head -1) > $out/bf_$c.txt 2>&1 || exit 1
done
grep -A1 "label_attn" $out/e200_*.txt $out/bf_*.txt | cut -c1-220
