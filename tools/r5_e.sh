cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5e
rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_irse.py -m gpu -x -q -k "winograd" > $O/t_wino.log 2>&1; echo "wino rc=$?"; tail -5 $O/t_wino.log
for mw in 4 8; do
echo "== W2E_TUNE_MW=$mw batch 8" >> $O/mw_ab.txt
W2E_TUNE_MW=$mw timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 30 --only 12,14,16 2>&1 | grep -v amdgpu >> $O/mw_ab.txt
echo "== W2E_TUNE_MW=$mw batch 4" >> $O/mw_ab.txt
W2E_TUNE_MW=$mw timeout -k 10 200 python3 tools/layer_bench.py --batch 4 --warm 1.0 --iters 30 --only 12,14,16 2>&1 | grep -v amdgpu >> $O/mw_ab.txt
done
cat $O/mw_ab.txt
