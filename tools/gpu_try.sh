set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; echo rc=$?
tail -4 gpurun_out/full_tests.log
python3 bench.py --warmup 2 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --steps 10 --skip-sorted-leg --other-configs "C5,C3G" > gpurun_out/try_c3.json 2> gpurun_out/try_c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/try_c3.json").read().strip().split("\n")[-1])
print("C3", d["ms_per_step"], d["kernel_ms"], d["parity_sample"])
for k, v in d["other_configs"].items(): print(k, v["ms_per_step"], v["kernel_ms"], v["parity_sample"])
PY
