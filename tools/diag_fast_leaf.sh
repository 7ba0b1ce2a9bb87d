#!/bin/bash
# Timing-only diagnostic: rebuild libmadnlp_hip.so with a pivot kernel that does no arithmetic (-DMNK_DIAG_FAST_LEAF=1, results
# void), to be run on the GPU box with tools/dag_time.py; `tools/diag_fast_leaf.sh off` restores the product build.
# (Run in the build container: hipcc cross-compiles; the .so travels with the snapshot.)
cd "$(dirname "$0")/.."
L=madnlp.jl_amd/lib
D=""; [ "$1" != "off" ] && D="-DMNK_DIAG_FAST_LEAF=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c madnlp.jl_amd/csrc/factor.hip -o $L/obj/factor.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $L/obj/*.o -o $L/libmadnlp_hip.so && echo "built with '$D'"
