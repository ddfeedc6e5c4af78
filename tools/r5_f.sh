cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], "dom", d["roofline"]["dominant_kernel"]["frac"], "wino", d["roofline"]["winograd"]["frac"])
for k in ("n1_b8", "config3", "config5"):
    print(k, d[k]["value"], d[k]["ms_per_step"])
print("stack", d["stack_mfma_frac_of_step"], "hbm", d["roofline_hbm"]["frac"])
PY
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_all.log
