# A/B of the shipped library against an alternate build (tools/build_alt.sh <tag> ...) on the task-DAG schedule, alternating runs on one box.
# usage: tools/leaf_ab.sh <tag> [N ...]   -> gpurun_out/leaf_ab_<tag>.txt
tag=${1:-leaf0}; shift
mkdir -p gpurun_out
out=gpurun_out/leaf_ab_$tag.txt
: > $out
for rep in 1 2; do
  timeout 300 python tools/leaf_ab.py ${@:-2048 6100 11192} 2>&1 | grep -v amdgpu.ids >> $out
  MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_$tag.so timeout 300 python tools/leaf_ab.py ${@:-2048 6100 11192} 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
