#!/bin/bash
# Alternate build of the library for A/B runs through MNK_LIBPATH: recompiles the named sources with extra flags and links
# them with the shipped objects.  usage: tools/build_alt.sh <tag> "<extra flags>" <source.hip> [...]
#   e.g. tools/build_alt.sh leaf0 "-DMNK_LEAF_V=0" factor.hip  ->  madnlp.jl_amd/lib/libmadnlp_hip_leaf0.so
set -e
cd "$(dirname "$0")/.."
tag=$1; flags=$2; shift 2
lib=madnlp.jl_amd/lib; obj=$lib/obj; alt=$lib/obj_$tag
mkdir -p $alt
python -c "import madnlp_jl_amd as mj; mj.build()" > /dev/null
objs=""
for o in $obj/*.o; do
  b=$(basename $o .o)
  use=$o
  for s in "$@"; do
    if [ "$s" = "$b.hip" ]; then
      extra=""
      case $b in gemm_f64|dag|bk) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c madnlp.jl_amd/csrc/$s -o $alt/$b.o
      use=$alt/$b.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -o $lib/libmadnlp_hip_$tag.so
echo built $lib/libmadnlp_hip_$tag.so
