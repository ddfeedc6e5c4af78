cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5m
rm -rf $O; mkdir -p $O
W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so W2E_HIPCC_FLAGS=-DW2E_TUNING python3 -m where2edit_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
for sk in 0 1 32 33 4 12 5 2; do
echo "== tuning build, W2E_TUNE_SKIP=$sk (1 no stores, 2 no K loop, 4 stage only the first chunk, 8 no wait/barrier after the first chunk, 32 no UP border)" >> $O/up_probe.txt
W2E_TUNE_SKIP=$sk W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 0.5 --iters 20 --only 9,11,13,15 2>&1 | grep -v "amdgpu\|total\|layer" >> $O/up_probe.txt
done
cat $O/up_probe.txt
