# Round evidence on one GPU box: the default bench line FIRST (rocprofv3 counter passes leave the GPU at the profiler's
# pinned clocks for a while), then the rocprofv3 --stats pass, the two PMC passes, the per-layer / per-kernel benches.
# Raw rocprofv3 output goes to /tmp (the kernel traces are tens of MB); only the stats CSVs and summaries come back.
#   usage: bash tools/collect_evidence.sh [tag]        (writes gpurun_out/ev_<tag>/)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r4}
O=$GRAFT_REPO_ROOT/gpurun_out/ev_$TAG
R=/tmp/w2e_prof_$TAG
rm -rf $O $R; mkdir -p $O $R
SECONDS=0; timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "default bench.py run: $SECONDS s wall" > $O/bench_line_wall.txt
echo bench done
timeout -k 10 300 python3 bench.py --workload 3 --batch 8 --no-cpu-baseline --no-preview --no-config3 --no-config5 --no-n1-b8 > $O/bench_line_w3_b8.json 2> $O/bench_line_w3_b8.err || true
timeout -k 10 300 python3 bench.py --workload 2 --batch 8 --no-cpu-baseline --no-preview --no-config3 --no-config5 --no-n1-b8 > $O/bench_line_w2_b8.json 2> $O/bench_line_w2_b8.err || true
echo config-3 done
timeout -k 10 200 python3 tools/layer_bench.py --warm 1.5 --iters 50 > $O/layer_bench.txt 2>&1 || true
timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 30 > $O/layer_bench_b8.txt 2>&1 || true
timeout -k 10 300 python3 bench.py --workload 5 --batch 8 > $O/bench_line_w5_b8.json 2> $O/bench_line_w5_b8.err || true
timeout -k 10 300 python3 bench.py --workload 5 --batch 4 > $O/bench_line_w5_b4.json 2> $O/bench_line_w5_b4.err || true
timeout -k 10 200 python3 tools/mem_bench.py > $O/mem_bench.txt 2>&1 || true
echo micro done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/stats -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline > $O/bench_stats.log 2>&1
cp $(find $R/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $O/kernel_stats.csv auto > $O/kernel_stats_summary.txt
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/stats3 -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 3 --batch 8 --steps 10 --warmup 3 --no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline > $O/bench_stats_w3.log 2>&1
cp $(find $R/stats3 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_w3_b8.csv
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $O/kernel_stats_w3_b8.csv auto > $O/kernel_stats_w3_b8_summary.txt
echo stats3 done
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-stabilise --graph off --no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing > $O/bench_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o write --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-stabilise --graph off --no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing > $O/bench_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find $R/fetch -name '*counter_collection.csv' | head -1) $(find $R/write -name '*counter_collection.csv' | head -1) $TAG $O > $O/pmc.log 2>&1 || true
echo pmc done
# SQ counter pass (its own run: --pmc must not be combined with the tracing domains other than --kernel-trace): matrix-pipe occupancy
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/sq -o sq --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-stabilise --graph off --no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing > $O/bench_sq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_mfma.py $(find $R/sq -name '*counter_collection.csv' | head -1) > $O/pmc_mfma.txt 2>&1 || true
echo sq done
# the Winograd-domain contraction kernel alone (layers 512 @ 64^2 and 256 @ 128^2, batch 8): L2 hit rate, HBM fetch, matrix-pipe occupancy
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/wg1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py --batch 8 --iters 3 --only 8,10 > $O/wg_pmc1.log 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/wg2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py --batch 8 --iters 3 --only 8,10 > $O/wg_pmc2.log 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/wg3 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py --batch 8 --iters 3 --only 8,10 > $O/wg_pmc3.log 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/wino_gemm_pmc.py $(find $R/wg1 -name '*counter_collection.csv' | head -1) $(find $R/wg2 -name '*counter_collection.csv' | head -1) $(find $R/wg3 -name '*counter_collection.csv' | head -1) > $O/wino_gemm_pmc.txt 2>&1 || true
echo wino-gemm pmc done
cd $GRAFT_REPO_ROOT
# the Winograd forms against the direct kernels, layer by layer (same-resolution layers; batch 8 = the merged forward, batch 4 = the backward)
for wg in 0 auto; do for bb in 8 4; do
echo "== W2E_WINOGRAD=$wg, batch $bb" >> $O/winograd.txt
W2E_WINOGRAD=$wg timeout -k 10 200 python3 tools/layer_bench.py --batch $bb --warm 1.0 --iters 20 2>&1 | grep "layer\|same\|total" >> $O/winograd.txt || true
done; done
timeout -k 10 300 python3 tools/irse_shapes.py 2>&1 | grep -v amdgpu > $O/irse_shapes.txt || true
echo winograd done
timeout -k 10 200 python3 tools/cfg_selections.py $O/cfg_selections.txt > /dev/null 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/issue_probe.hip -o /tmp/issue_probe 2> /dev/null && /tmp/issue_probe > $O/issue_probe.txt 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/dma_probe.hip -o /tmp/dma_probe 2> /dev/null && /tmp/dma_probe > $O/dma_probe.txt 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/pk_probe.hip -o /tmp/pk_probe 2> /dev/null && /tmp/pk_probe > $O/pk_probe.txt 2>&1 || true
echo extras done
