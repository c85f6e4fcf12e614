set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/calib/inf_variants/phases.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4n_prof -- python $GRAFT_REPO_ROOT/tools/calib/inf_variants/phases.py two > $GRAFT_REPO_ROOT/gpurun_out/r4n_phases.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r4n_phases.log
find gpurun_out/r4n_prof -name "*kernel_stats*" | head
for f in $(find gpurun_out/r4n_prof -name "*kernel_stats.csv"); do head -8 $f; done
find gpurun_out/r4n_prof -name "*_kernel_trace.csv" -size +1k -exec sh -c 'python tools/summarize_trace_by_grid.py {} ' \; 2>/dev/null | head -40
