# Round evidence on one GPU box: the default bench line FIRST (rocprofv3 counter passes leave the GPU at the profiler's
# pinned clocks for a while), then the rocprofv3 --stats pass, the two PMC passes, the per-layer / per-kernel benches.
# Raw rocprofv3 output goes to /tmp (the kernel traces are tens of MB); only the stats CSVs and summaries come back.
#   usage: bash tools/collect_evidence.sh [tag]        (writes gpurun_out/ev_<tag>/)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
#   usage: bash tools/collect_evidence.sh [tag] [part]     part 1 = bench lines + micro-benches, 2 = rocprofv3 passes + probes (a gpurun call is
#   limited to 20 minutes: the two parts are two calls), default both
TAG=${1:-r5}
PART=${2:-all}
O=$GRAFT_REPO_ROOT/gpurun_out/ev_$TAG
R=/tmp/w2e_prof_$TAG
rm -rf $R; mkdir -p $O $R
if [ "$PART" != "2" ]; then
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
fi
if [ "$PART" = "1" ]; then exit 0; fi
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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $GRAFT_REPO_ROOT/tools/fetch_calib.hip -o /tmp/fetch_calib 2> /dev/null && /tmp/fetch_calib > $O/fetch_calib_table.txt 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fc1 -o p --output-format csv -- /tmp/fetch_calib > $O/fc1.log 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d $R/fc2 -o p --output-format csv -- /tmp/fetch_calib > $O/fc2.log 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/fetch_calib.py $O/fetch_calib_table.txt $(find $R/fc1 $R/fc2 -name '*counter_collection.csv') > $O/fetch_calibration.txt 2>&1 || true
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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value tools/issue_share_probe.hip -o /tmp/isp 2> /dev/null && /tmp/isp > $O/issue_share_probe.txt 2>&1 || true
echo extras done
# round 5: the strong-scaling mode at N = 1 (global batch 64 as 8 micro-batches of 8), the 4-rank launcher run on this one GPU (gloo; a GPU box
# allows 6 processes on its card -- the 8-rank control flow is rehearsed without GPU work by `bench.py --gpus 8 --rehearse`, tests/test_dist_cpu.py),
# the same A/B of the fused kernel's two round-5 switches
timeout -k 10 300 python3 bench.py --scaling strong --no-cpu-baseline > $O/bench_line_strong_n1.json 2> $O/bench_line_strong_n1.err || true
timeout -k 10 400 python3 bench.py --gpus 4 --dist-backend gloo --batch 1 --steps 3 --warmup 2 > $O/bench_line_gloo4.json 2> $O/bench_line_gloo4.err || true
timeout -k 10 120 python3 bench.py --gpus 8 --rehearse --steps 5 --warmup 2 > $O/bench_rehearse8.json 2> $O/bench_rehearse8.err || true
for sw in "W2E_TUNE_MW=4 W2E_TUNE_XCD=0" "W2E_TUNE_MW=4" ""; do
echo "== ${sw:-default (64-channel workgroups, XCD-contiguous blocks)}, batch 8" >> $O/fused_ab.txt
env $sw timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 30 --only 12,14,16 2>&1 | grep -v amdgpu >> $O/fused_ab.txt || true
done
echo round5 done
