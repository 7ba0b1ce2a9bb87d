mkdir -p gpurun_out; out=gpurun_out/deep_sweep_dma.txt; : > $out
for n in 4608 5120 5376 5632 5888 6144 6656 7168; do
  for rep in 1 2; do
  echo "== N=$n deep (every row in the band)" >> $out; MNK_OPTIONS=dag_deep_rows=100000 timeout 120 python tools/dag_time.py $n LDL 2>&1 | grep -v amdgpu.ids >> $out
  echo "== N=$n band + bulk" >> $out; MNK_OPTIONS=dag_deep_rows=0 timeout 120 python tools/dag_time.py $n LDL 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
