#!/bin/bash
# Builds the library and the BB_PROF diagnostic build of the bf16x6 attention backward (tools/prof_bwd_phases.py).
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -DBB_PROF $BB_DEFS gzsl_text_amd/csrc/attn_bwd_bf16.hip -o tools/_diag/libbwd_prof.so 2>&1 | grep -E " error" 
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E " error|built|Error" | head -3
