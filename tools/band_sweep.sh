mkdir -p gpurun_out
out=gpurun_out/r5_band_sweep.txt
: > $out
run() { echo "== MNK_DAG_CUS=$1 DAG_BAND=$2 N=$3" >> $out; MNK_DAG_CUS=$1 DAG_BAND=$2 timeout 120 python tools/dag_time.py $3 LDL 2>&1 | grep -v amdgpu.ids >> $out; }
run 16 16 11192
run 32 16 11192
run 32 24 11192
run 32 32 11192
run 16 16 11192
run 32 32 11192
run 32 24 11192
run 32 32 16384
run 16 16 16384
run 32 32 8000
run 16 16 8000
cat $out
