# A/B of the shipped library against an alternate build on the persistent solve: time per solve + hash of the solution's bits, alternating runs on one box.
# usage: tools/solve_ab.sh <tag> [N ...]   -> gpurun_out/solve_ab_<tag>.txt
tag=${1:-dah1}; shift
mkdir -p gpurun_out
out=gpurun_out/solve_ab_$tag.txt
: > $out
for rep in 1 2; do
  for n in ${@:-2048 11192 22384}; do
    echo "== libmadnlp_hip.so N=$n" >> $out
    SOLVE_HASH=1 timeout 300 python tools/solve_time.py $n 2>&1 | grep -v amdgpu.ids >> $out
    echo "== libmadnlp_hip_$tag.so N=$n" >> $out
    SOLVE_HASH=1 MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_$tag.so timeout 300 python tools/solve_time.py $n 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
