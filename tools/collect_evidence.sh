# Round evidence on one GPU box: the default bench line FIRST (rocprofv3 counter passes leave the GPU at the profiler's
# pinned clocks for a while), then the rocprofv3 --stats pass, the two PMC passes, the per-layer / per-kernel benches.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ev_r1
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
echo bench done
timeout -k 10 300 python3 bench.py --conv-precision bf16x3 --no-cpu-baseline > $O/bench_line_bf16x3.json 2> $O/bench_line_bf16x3.err || true
W2E_CONV_PRECISION=bf16x3 timeout -k 10 200 python3 tools/layer_bench.py --warm 1.5 --iters 50 > $O/layer_bench_bf16x3.txt 2>&1 || true
echo bf16x3 done
timeout -k 10 200 python3 tools/layer_bench.py --warm 1.5 --iters 50 > $O/layer_bench.txt 2>&1
echo layer done
timeout -k 10 200 python3 tools/mem_bench.py > $O/mem_bench.txt 2>&1
echo mem done
timeout -k 10 60 tools/bin/mfma_peak > $O/mfma_peak.txt 2>&1 || true
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --spinup 0 --no-preview --no-cpu-baseline > $O/bench_stats.log 2>&1
echo stats done
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --spinup 0 --no-preview --no-cpu-baseline --no-kernel-timing > $O/bench_fetch.log 2>&1
echo fetch done
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --spinup 0 --no-preview --no-cpu-baseline --no-kernel-timing > $O/bench_write.log 2>&1
echo write done
