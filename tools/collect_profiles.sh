#!/bin/bash
# Collects the round's profile evidence on the GPU box and leaves only SMALL files under gpurun_out/$1 (the raw rocprofv3
# databases are hundreds of MB; gpurun copies back at most 64 MiB):
#   kernel trace + stats of a 7-step bench run, FETCH_SIZE / WRITE_SIZE / SQ counter passes (separate runs, as the
#   microarchitecture guide prescribes), each summarised by tools/pmc_summary.py.
#   usage: bash tools/collect_profiles.sh r02_final
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.trace -- python3 $B > $out/trace.log 2>&1 || exit 1
python3 tools/summarize_profile.py /tmp/$tag.trace $out/bench_cfg2 "rocprofv3 --kernel-trace --stats --output-format csv -- python3 $B (7 steps traced)" || exit 1
B3="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass -d /tmp/$tag.$name -- python3 $B3 > $out/pmc_$name.log 2>&1 || exit 1
  db=$(ls /tmp/$tag.$name/*/*.db | head -1)
  python3 tools/pmc_summary.py $db > $out/pmc_$name.txt 2>&1 || exit 1
done
echo "profiles collected in $out"; ls -la $out
