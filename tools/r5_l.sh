cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5l
rm -rf $O; mkdir -p $O
timeout -k 10 120 python3 bench.py --gpus 8 --rehearse --steps 5 --warmup 2 > $O/bench_rehearse8.json 2> $O/bench_rehearse8.err; echo "rehearse8 rc=$?"; cat $O/bench_rehearse8.json; grep -c amdgpu.ids $O/bench_rehearse8.err
