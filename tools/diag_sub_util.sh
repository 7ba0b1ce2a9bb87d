mkdir -p gpurun_out
out=gpurun_out/r5_diag_sub_util.txt
: > $out
for alg in CHOLESKY LDL; do for lib in madnlp.jl_amd/lib/libmadnlp_hip.so madnlp.jl_amd/lib/libmadnlp_hip_nosub.so; do
  echo "=== $lib $alg" >> $out
  MNK_LIBPATH=$lib timeout 200 python tools/dag_util.py 11192 $alg 2>&1 | grep -v amdgpu.ids | grep -v "^  *[0-9]* k-steps: *[0-9] \|k-steps:  *[1-9][0-9] " >> $out
done; done
cat $out
