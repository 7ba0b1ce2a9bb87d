mkdir -p gpurun_out
out=gpurun_out/r5_chunk_accuracy.txt
: > $out
for opt in "dag_chunk=1,dag_taper0=1" "dag_chunk=2,dag_taper0=1" "dag_chunk=4,dag_taper0=1" "dag_chunk=16"; do
  echo "=== MNK_OPTIONS=$opt" >> $out
  MNK_OPTIONS=$opt timeout 300 python tools/acopf_trajectory.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep "^# case\|^# trials\|^# first" >> $out
  MNK_OPTIONS=$opt timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
