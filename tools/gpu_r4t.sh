set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4t_c5 -- python3 bench.py --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --config C5 --steps 3 --skip-sorted-leg --other-configs "" > gpurun_out/r4t_c5.json 2> gpurun_out/r4t_c5.err; echo rc=$?
f=$(find gpurun_out/r4t_c5 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-4 | sed 's/pvc3::(anonymous namespace):://' | cut -c1-140
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4t_c5.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["kernel_ms"])
PY
