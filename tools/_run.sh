cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt5 -- python3 bench.py --config C5 --no-build --steps 3 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg > gpurun_out/kt5.json 2> gpurun_out/kt5.err
python3 - <<'PY'
import csv, glob, json
f = sorted(glob.glob("gpurun_out/kt5/**/*_kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
	if "pvc3" in r["Name"]: print("%-60s %6s %12.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
d = json.loads(open("gpurun_out/kt5.json").readline()); print(d["ms_per_step"], d["kernel_ms"])
PY
