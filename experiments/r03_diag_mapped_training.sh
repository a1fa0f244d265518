cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for trial in 1 2 3 4; do
PORT=$((29731 + trial))
EXTRA="A=1"
[ $trial -ge 4 ] && EXTRA="WM_VMM_FREE_VA=1"
for r in 0 1; do env $EXTRA WM_TEST_ONLY=mapped_training WM_TEST_DIAG=1 WM_TEST_REPS=3 OMP_NUM_THREADS=1 WM_EXCHANGE_CHUNKS=1 timeout 300 python tests/_dist_worker.py $r 2 $PORT hip > gpurun_out/r03/diag${trial}_rank$r.txt 2>&1 & done
wait
echo "=== trial $trial $EXTRA"
for r in 0 1; do grep -E "DIAG|FAIL|row |RANK|Error" gpurun_out/r03/diag${trial}_rank$r.txt | head -12; grep -c "^ok" gpurun_out/r03/diag${trial}_rank$r.txt; done
done
