#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_c5_ab.sh NAME [NAME ...] — bench.py --config C5 under rocprofv3 --kernel-trace
# --stats once per diagnostic build (tools/diag_build.py; "main" = the product library): the realignment kernels' averages side by side.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  lib=panvc3_amd/_build/diag_$n/libpvc3_hip.so; [ "$n" = main ] && lib=panvc3_amd/libpvc3_hip.so
  export PVC3_LIB_PATH=$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_$n -- python3 bench.py --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --config C5 --steps 3 --skip-sorted-leg > gpurun_out/c5_$n.json 2> gpurun_out/c5_$n.err
  f=$(find gpurun_out/c5_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "realign" in r["Name"] or "run_prefix" in r["Name"]:
        print("%-60s calls %s avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
