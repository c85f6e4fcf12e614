cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --no-build --steps 5 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg > gpurun_out/kt.json 2> gpurun_out/kt.err
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/kt/**/*_kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
	if "scan" in r["Name"] or "gather" in r["Name"]: print("%-60s %6s %12.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
