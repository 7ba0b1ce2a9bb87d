# >= 10 000 factorizations of the C3 system per bound of the schedule's device-side waits (polls of ~0.17 us:
# 294 000 = 0.05 s, 588 000 = 0.1 s, 1 760 000 = 0.3 s), 672-workgroup bulk grid; a bound is clean if nothing falls back
mkdir -p gpurun_out
out=gpurun_out/r5_spin_bound_sweep.txt
: > $out
for lim in 294000 588000 1760000; do
  echo "== dag_spin_limit=$lim" >> $out
  timeout 400 python tools/c5_loop.py 640 16 dag_spin_limit=$lim 2>&1 | grep -v amdgpu.ids | tail -12 >> $out
done
cat $out
