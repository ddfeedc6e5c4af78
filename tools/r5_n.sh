cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5n
rm -rf $O; mkdir -p $O
W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so W2E_HIPCC_FLAGS=-DW2E_TUNING python3 -m where2edit_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
W2E_TUNE_CLOCK=1 W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 70 --only 9,11,13,15 > $O/clock.txt 2>&1
grep -v amdgpu $O/clock.txt | tail -40
