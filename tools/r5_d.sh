cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
rm -rf $O; mkdir -p $O
timeout -k 10 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 SIGUSR2 nostop noprint" -ex run -ex "bt 40" -ex "info threads" --args python3 -m pytest tests/test_gpu_step.py -m gpu -x -q -k "workload2 and 4" > $O/gdb.log 2>&1
echo "gdb rc=$?"; grep -n "SIGSEGV\|^#" $O/gdb.log | head -60
