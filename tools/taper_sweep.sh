mkdir -p gpurun_out; out=gpurun_out/taper_sweep.txt; : > $out
for rep in 1 2; do
for t in 2 1 4 8; do
  echo "== dag_taper0=$t" >> $out; MNK_OPTIONS=dag_taper0=$t timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out
done
for c in 32 128; do
  echo "== dag_chunk=$c" >> $out; MNK_OPTIONS=dag_chunk=$c timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out
done
done
grep -E "^==|factorize" $out | paste - - | awk '{print $2, $8, $9}'
