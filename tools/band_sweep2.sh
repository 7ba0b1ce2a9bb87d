# fewer chain CUs than 16: does the bulk kernel gain more from the CUs than the chain loses?  (round 5)
mkdir -p gpurun_out
out=gpurun_out/band_sweep2.txt
: > $out
run() { echo "== MNK_DAG_CUS=$1 DAG_BAND=$2 N=$3" >> $out; MNK_DAG_CUS=$1 DAG_BAND=$2 timeout 120 python tools/dag_time.py $3 LDL 2>&1 | grep -v amdgpu.ids >> $out; }
run 16 16 11192
run 8 8 11192
run 12 12 11192
run 16 12 11192
run 16 8 11192
run 8 8 11192
run 12 12 11192
run 16 16 11192
cat $out
