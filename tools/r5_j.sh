cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5j
rm -rf $O; mkdir -p $O
Q="--no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2; do
for opt in "" "--lib-option tune_mw=4" "--lib-option tune_mw=4 --lib-option tune_xcd=0"; do
timeout -k 10 200 python3 bench.py $Q $opt > $O/line.json 2> $O/line.err
python3 -c "
import json; d=json.load(open('$O/line.json')); print('rep $rep', d['config']['lib_options'], 'b4', round(d['value'],1), round(d['ms_per_step'],3))" >> $O/ab.txt
timeout -k 10 200 python3 bench.py --batch 8 $Q $opt > $O/line.json 2> $O/line.err
python3 -c "
import json; d=json.load(open('$O/line.json')); print('rep $rep', d['config']['lib_options'], 'b8', round(d['value'],1), round(d['ms_per_step'],3))" >> $O/ab.txt
done
done
cat $O/ab.txt
cd /tmp
for tag in def mw4; do
opt=""; [ $tag = mw4 ] && opt="--lib-option tune_mw=4 --lib-option tune_xcd=0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$tag -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $Q $opt > $O/stats_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/st_$tag -name '*kernel_stats.csv' | head -1) auto | grep "fused3\|total kernel" > $O/stats_$tag.txt
echo "== $tag"; cat $O/stats_$tag.txt
done
