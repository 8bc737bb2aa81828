#!/bin/bash
# Kernel-trace summary of the WGAN-GP step at two batch sizes; leaves only the small stats CSVs under gpurun_out/$1.
#   usage (on the GPU box): bash tools/profile_gan.sh ganprof
set -u
tag=${1:-ganprof}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.4096 -o gp -- python3 $R/tools/bench_gan.py --bg 4096 --steps 10 > $out/b4096.log 2>&1 || exit 1
find /tmp/$tag.4096 -name "*kernel_stats.csv" -exec cp {} $out/k4096.csv \;
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.8 -o gp -- python3 $R/tools/bench_gan.py --bg 8 --steps 20 > $out/b8.log 2>&1 || exit 1
find /tmp/$tag.8 -name "*kernel_stats.csv" -exec cp {} $out/k8.csv \;
echo done
