# Round 5, first GPU call: new tests, the FETCH_SIZE calibration probe, the XCD-map A/B.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5a
rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench.py tests/test_gpu_step.py -m gpu -x -q -k "bench or accumulated" > $O/tests_new.log 2>&1 || { tail -40 $O/tests_new.log; echo NEW TESTS FAILED; }
tail -5 $O/tests_new.log
echo new-tests done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value tools/fetch_calib.hip -o /tmp/fetch_calib 2> /dev/null
/tmp/fetch_calib > $O/fetch_calib_table.txt 2>&1
cat $O/fetch_calib_table.txt
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc1 -o p --output-format csv -- /tmp/fetch_calib > $O/fc1.log 2>&1 || echo fc1 failed
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d /tmp/fc2 -o p --output-format csv -- /tmp/fetch_calib > $O/fc2.log 2>&1 || echo fc2 failed
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/fc3 -o p --output-format csv -- /tmp/fetch_calib > $O/fc3.log 2>&1 || echo fc3 failed
cd $GRAFT_REPO_ROOT
python3 tools/fetch_calib.py $O/fetch_calib_table.txt $(find /tmp/fc1 /tmp/fc2 /tmp/fc3 -name '*counter_collection.csv') > $O/fetch_calibration.txt 2>&1 || true
cat $O/fetch_calibration.txt
echo calib done
for x in 0 1 3; do
echo "== W2E_TUNE_XCD=$x batch 8" >> $O/xcd_ab.txt
W2E_TUNE_XCD=$x timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 30 --only 9,11,12,13,14,15,16 2>&1 | grep -v amdgpu >> $O/xcd_ab.txt
done
for x in 0 1 3; do
echo "== W2E_TUNE_XCD=$x batch 4" >> $O/xcd_ab.txt
W2E_TUNE_XCD=$x timeout -k 10 200 python3 tools/layer_bench.py --batch 4 --warm 1.0 --iters 30 --only 9,11,12,13,14,15,16 2>&1 | grep -v amdgpu >> $O/xcd_ab.txt
done
cat $O/xcd_ab.txt
echo ab done
cd /tmp
for x in 0 1 3; do
W2E_TUNE_XCD=$x timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/lf$x -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py --batch 8 --iters 3 --only 11,12,14,15,16 > $O/lf$x.log 2>&1 || echo lf$x failed
python3 - <<PY > $O/layer_fetch_xcd$x.txt
import csv, collections, glob
d = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob('/tmp/lf$x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            k = r['Kernel_Name'].split('(')[0][:70] + ' grid ' + r.get('Grid_Size', '?')
            d[k][0] += 1; d[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(d.items(), key=lambda kv: -kv[1][1]):
    if v / n > 1000: print(f'{k}, launches {n}, FETCH_SIZE {v / n / 1024:.1f} MiB/launch')
PY
echo "== xcd $x"; cat $O/layer_fetch_xcd$x.txt
done
echo fetch-ab done
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1 || { tail -60 $O/tests_all.log; echo FULL SUITE FAILED; }
tail -8 $O/tests_all.log
