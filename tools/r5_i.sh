cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5i
rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 tools/irse_shapes.py 2>&1 | grep -v amdgpu > $O/irse_shapes.txt; cat $O/irse_shapes.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_attention.py -m gpu -x -q -k "amp" > $O/t_amp.log 2>&1; echo "amp rc=$?"; tail -5 $O/t_amp.log
