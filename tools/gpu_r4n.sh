set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4n_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/r4n_tests.log
timeout -k 10 400 python tools/calib/inf_variants/phases.py > gpurun_out/r4n_phases.log 2>&1; echo rc=$?
tail -8 gpurun_out/r4n_phases.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4n_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4n_prof -- python $GRAFT_REPO_ROOT/tools/calib/inf_variants/phases.py two > $GRAFT_REPO_ROOT/gpurun_out/r4n_prof.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/r4n_prof -name "*_kernel_trace.csv" -size +1k -exec python tools/summarize_trace_by_grid.py {} \; > gpurun_out/r4n_by_grid.txt 2>&1
cat gpurun_out/r4n_by_grid.txt
