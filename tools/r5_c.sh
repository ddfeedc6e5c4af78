cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5c
rm -rf $O; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_step.py -m gpu -x -q > $O/t_step.log 2>&1; echo "step rc=$?"; tail -75 $O/t_step.log
