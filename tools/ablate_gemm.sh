#!/bin/bash
# Builds experiment variants of the GEMM kernel into tools/_abl/ (timing only for GZ_GEMM_ABL != 0: wrong results).
#   tools/ablate_gemm.sh NAME "-DGZ_GEMM_ABL=3" [NAME2 "-DGZ_GEMM_PIPE=1" ...]
set -e
cd "$(dirname "$0")/../gzsl_text_amd/csrc"
mkdir -p ../../tools/_abl
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $2 -c gemm.hip -o /tmp/gemm_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_$1.o attn.o elementwise.o gan.o -o ../../tools/_abl/libgemm_$1.so
  shift 2
done
