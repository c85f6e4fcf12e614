#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh NAME — the rocprofv3 passes profiles/README.md lists, into
# gpurun_out/NAME_*; summarise afterwards (here) with tools/summarize_rocprof.py.  Build first: bench.py runs with --no-build.
set -e
name=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${name}_trace -- $B --steps 5 > gpurun_out/${name}_trace.json 2> gpurun_out/${name}_trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${name}_fetch -- $B --steps 2 --skip-sorted-leg > gpurun_out/${name}_fetch.json 2> gpurun_out/${name}_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${name}_write -- $B --steps 2 --skip-sorted-leg > gpurun_out/${name}_write.json 2> gpurun_out/${name}_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${name}_c5_trace -- $B --config C5 --steps 3 --skip-sorted-leg > gpurun_out/${name}_c5_trace.json 2> gpurun_out/${name}_c5_trace.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${name}_c3g_trace -- $B --config C3G --steps 3 --skip-sorted-leg > gpurun_out/${name}_c3g_trace.json 2> gpurun_out/${name}_c3g_trace.err
echo done
