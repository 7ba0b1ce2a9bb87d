# subtract-order accumulation of the diagonal tiles (MNK_DAG_DIAG_SUB) against the sum-then-subtract order: AC-OPF trajectory and time
mkdir -p gpurun_out
out=gpurun_out/r5_diag_sub_ab.txt
: > $out
L=madnlp.jl_amd/lib
for lib in $L/libmadnlp_hip.so $L/libmadnlp_hip_sub2.so $L/libmadnlp_hip_nosub.so; do
  echo "=== $lib" >> $out
  MNK_LIBPATH=$lib timeout 300 python tools/acopf_trajectory.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep "^# case\|^# trials" >> $out
done
for rep in 1 2 3; do for lib in $L/libmadnlp_hip.so $L/libmadnlp_hip_sub2.so $L/libmadnlp_hip_nosub.so; do
  echo "=== $lib" >> $out
  for n in 11192 6100; do
  MNK_LIBPATH=$lib timeout 120 python tools/dag_time.py $n LDL 2>&1 | grep -v amdgpu.ids >> $out
  MNK_LIBPATH=$lib timeout 120 python tools/dag_time.py $n CHOLESKY 2>&1 | grep -v amdgpu.ids >> $out
  done
done; done
cat $out
