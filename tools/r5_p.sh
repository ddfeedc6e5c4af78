cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5p
rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "styled or modconv or upsample or generator" > $O/t_conv.log 2>&1; echo "conv tests rc=$?"; tail -3 $O/t_conv.log
for kc in 0 1 0 1; do
echo "== W2E_TUNE_KC16=$kc batch 8" >> $O/kc_ab.txt
W2E_TUNE_KC16=$kc timeout -k 10 200 python3 tools/layer_bench.py --batch 8 --warm 1.0 --iters 40 --only 5,7,9,11,13,15 2>&1 | grep -v "amdgpu\|layer" >> $O/kc_ab.txt
done
cat $O/kc_ab.txt
Q="--no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2 3; do
for opt in "" "--lib-option tune_kc16=0"; do
timeout -k 10 200 python3 bench.py $Q $opt > $O/line.json 2> $O/line.err
python3 -c "
import json; d=json.load(open('$O/line.json')); print('rep $rep', d['config']['lib_options'], 'b4', round(d['value'],1), round(d['ms_per_step'],3))" >> $O/ab.txt
done
done
cat $O/ab.txt
