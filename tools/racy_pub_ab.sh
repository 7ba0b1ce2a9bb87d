# timing-only (results void): the chain's critical steps without the wait for their stores in front of the publication
mkdir -p gpurun_out
out=gpurun_out/r5_racy_pub_ab.txt
: > $out
L=madnlp.jl_amd/lib
for rep in 1 2; do for lib in $L/libmadnlp_hip.so $L/libmadnlp_hip_racy.so; do
  echo "=== $lib" >> $out
  for n in 2048 6100 11192; do MNK_LIBPATH=$lib timeout 120 python tools/dag_time.py $n LDL 2>&1 | grep -v amdgpu.ids >> $out; done
done; done
cat $out
