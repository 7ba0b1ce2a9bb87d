# staged hand-over of the pivot chain's diagonal blocks (leaf64.h, pp_strip) against the previous build: time and bits
mkdir -p gpurun_out
out=gpurun_out/r5_staged_handover_ab.txt
: > $out
for rep in 1 2; do
  timeout 300 python tools/leaf_ab.py 2048 6100 11192 2>&1 | grep -v amdgpu.ids >> $out
  MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_prev.so timeout 300 python tools/leaf_ab.py 2048 6100 11192 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
