set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5b
rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value tools/issue_share_probe.hip -o /tmp/isp 2> /dev/null
/tmp/isp > $O/issue_share_probe.txt 2>&1
cat $O/issue_share_probe.txt
echo probe done
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1 || { tail -60 $O/tests_all.log; echo FULL SUITE FAILED; }
tail -70 $O/tests_all.log
