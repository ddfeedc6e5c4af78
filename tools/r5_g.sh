cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5g
rm -rf $O; mkdir -p $O
W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so W2E_HIPCC_FLAGS=-DW2E_TUNING python3 -m where2edit_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
for mw in 8 4; do
echo "== tuning build, W2E_TUNE_MW=$mw" >> $O/fused_probe.txt
W2E_TUNE_MW=$mw W2E_LIB_PATH=$GRAFT_REPO_ROOT/where2edit_amd/lib/libw2e_tuning.so timeout -k 10 300 python3 tools/fused_probe.py 2>&1 | grep -v amdgpu >> $O/fused_probe.txt
done
cat $O/fused_probe.txt
