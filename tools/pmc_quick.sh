#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_quick.sh OUTDIR "COUNTER LIST" [bench args]
# Runs bench.py under rocprofv3 with one PMC pass and prints per-kernel averages.  bench.py gets --no-build: under the profiler it
# must never start a compiler (build with `python3 -m panvc3_amd.build` before; the libraries travel to the GPU box).
set -e
out=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err
python3 - "$out" <<'PY'
import csv, glob, collections, re, sys
f = glob.glob("gpurun_out/%s/**/*_counter_collection.csv" % sys.argv[1], recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
	m = re.search(r"(k_\w+)(?:<[^>]*>)?\(", r["Kernel_Name"])
	if m:
		acc[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
	print("%-20s %-22s %.5g" % (k[0], k[1], sum(v) / len(v)))
PY
