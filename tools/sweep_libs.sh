#!/bin/bash
# usage (on the GPU box, from the repo root): tools/sweep_libs.sh "bench args" NAME [NAME ...] — bench.py once per diagnostic build
# panvc3_amd/_build/diag_NAME (tools/diag_build.py), kernel times of each on one line.  All runs on one box: an A/B that holds.
args=$1; shift
for n in "$@"; do
	lib=panvc3_amd/_build/diag_$n/libpvc3_hip.so
	[ "$n" = main ] && lib=panvc3_amd/libpvc3_hip.so
	PVC3_LIB_PATH=$lib python3 bench.py --no-build --cpu-baseline-seconds 0 --skip-host-buffers $args > gpurun_out/sweep_$n.json 2> gpurun_out/sweep_$n.err || { echo "$n failed"; tail -3 gpurun_out/sweep_$n.err; continue; }
	python3 - "$n" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/sweep_%s.json" % sys.argv[1]).read().strip().split("\n")[-1])
s = d.get("sorted_input")
print("%-10s step %.3f ms  %s  frac %.3f  checksum %s%s" % (sys.argv[1], d["ms_per_step"], " ".join("%s %.3f" % kv for kv in d["kernel_ms"].items()), d["roofline"]["frac"], d["batch"]["checksum"],
	("  | sorted: fast %.3f frac %.3f" % (s["kernel_ms"]["project"], s["roofline"]["frac"])) if s else ""))
PY
done
