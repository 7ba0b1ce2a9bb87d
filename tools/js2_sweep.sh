# the second phase (deep band: every remaining row a strip of the chain, on a larger chain partition) for the chain-bound TAIL
# only: switch when 8 / 6 / 16 strip-columns remain (round 3 measured the switch at 96 strips: slower)
mkdir -p gpurun_out
out=gpurun_out/r5_js2_sweep.txt
: > $out
run() { echo "== MNK_DAG_CUS2=$1 dag_js2=$2" >> $out; MNK_DAG_CUS2=$1 MNK_OPTIONS="dag_js2=$2" timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out; }
echo "== default" >> $out; timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out
run 32 36
run 32 38
run 64 28
run 64 32
run 32 40
echo "== default" >> $out; timeout 120 python tools/dag_time.py 11192 LDL 2>&1 | grep -v amdgpu.ids >> $out
cat $out
