set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; echo rc=$?
tail -4 gpurun_out/full_tests.log
for w in 8192 16384; do
python3 bench.py --warmup 2 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --steps 10 --skip-sorted-leg --other-configs "" --option realign_waves=$w > gpurun_out/r4t_c3_$w.json 2> gpurun_out/r4t_c3_$w.err
python - $w <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4t_c3_%s.json" % sys.argv[1]).read().strip().split("\n")[-1])
print("C3 realign_waves", sys.argv[1], d["ms_per_step"], d["kernel_ms"])
PY
done
