#!/bin/bash
# The abort at the end of `rocprofv3 --pmc ... -- python3 tools/calib/inf_variants/pmc_deflate.py` (round 3's last GPU call): the same
# command once, and the variations that tell whose teardown it is — no profiler, the profiler without counters, fewer blocks,
# the compressor left open.  Every line of the result: the variant, the exit code, the last lines the process wrote.
# usage (GPU box): tools/calib/abort_probe.sh <output directory>
out=${1:-gpurun_out/abort_probe}
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/bgzf_bench.py --blocks 256 > "$out/prepare.txt" 2>&1 || exit 1		# (makes /tmp/pvc3_bgzf_bench/in.bam)
run() {	# name, environment, command...
	local name=$1; shift
	( "$@" ) > "$out/$name.txt" 2>&1
	echo "$name: exit $? :: $(grep -v '^$' "$out/$name.txt" | tail -n 3 | tr '\n' '|' | cut -c1-400)" >> "$out/summary.txt"
}
: > "$out/summary.txt"
run plain              python3 tools/calib/inf_variants/pmc_deflate.py
run trace_only         rocprofv3 --kernel-trace -d "$out/trace_only" -- python3 tools/calib/inf_variants/pmc_deflate.py
run pmc_fetch          rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -- python3 tools/calib/inf_variants/pmc_deflate.py
PMC_BLOCKS=512 run pmc_fetch_512 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_512" -- python3 tools/calib/inf_variants/pmc_deflate.py
PMC_KEEP=1 run pmc_fetch_kept rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_kept" -- python3 tools/calib/inf_variants/pmc_deflate.py
run pmc_inflate        rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_inflate" -- python3 tools/calib/inf_variants/pmc.py
cat "$out/summary.txt"
