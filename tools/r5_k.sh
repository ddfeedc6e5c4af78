cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5k
rm -rf $O; mkdir -p $O
Q="--no-preview --no-config3 --no-config5 --no-n1-b8 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2 3; do
for opt in "" "--lib-option tune_wide=0"; do
timeout -k 10 200 python3 bench.py $Q $opt > $O/line.json 2> $O/line.err
python3 -c "
import json; d=json.load(open('$O/line.json')); print('rep $rep', d['config']['lib_options'], 'b4', round(d['value'],1), round(d['ms_per_step'],3))" >> $O/ab.txt
done
done
cat $O/ab.txt
