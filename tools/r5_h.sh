cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5h
rm -rf $O; mkdir -p $O
for st in 0 1 2 0 1; do
echo "== W2E_TUNE_STAGGER=$st batch 8" >> $O/stagger_ab.txt
W2E_TUNE_STAGGER=$st timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 30 --only 12,14,16 2>&1 | grep -v amdgpu >> $O/stagger_ab.txt
done
for st in 0 1; do
echo "== W2E_TUNE_STAGGER=$st batch 4" >> $O/stagger_ab.txt
W2E_TUNE_STAGGER=$st timeout -k 10 200 python3 tools/layer_bench.py --batch 4 --warm 1.0 --iters 30 --only 12,14,16 2>&1 | grep -v amdgpu >> $O/stagger_ab.txt
done
cat $O/stagger_ab.txt
