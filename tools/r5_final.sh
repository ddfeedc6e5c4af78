cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5final
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py --gpus 4 --dist-backend gloo --batch 1 --steps 3 --warmup 2 > $O/bench_line_gloo4.json 2> $O/bench_line_gloo4.err; echo "gloo4 rc=$?"
timeout -k 10 120 python3 bench.py --gpus 8 --rehearse --steps 5 --warmup 2 > $O/bench_rehearse8.json 2> $O/bench_rehearse8.err; echo "rehearse8 rc=$?"; cat $O/bench_rehearse8.json
timeout -k 10 300 python3 __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
