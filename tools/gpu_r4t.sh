set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q > gpurun_out/r4t_tests.log 2>&1; echo rc=$?
tail -3 gpurun_out/r4t_tests.log
for w in 8192 16384 4096; do
python3 bench.py --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --config C5 --steps 4 --skip-sorted-leg --other-configs "" --option realign_waves=$w > gpurun_out/r4t_c5_$w.json 2> gpurun_out/r4t_c5_$w.err
python - $w <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4t_c5_%s.json" % sys.argv[1]).read().strip().split("\n")[-1])
print("realign_waves", sys.argv[1], d["ms_per_step"], d["kernel_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4t_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4t_c5 -- python3 bench.py --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --no-build --config C5 --steps 3 --skip-sorted-leg --other-configs "" > gpurun_out/r4t_c5.json 2> gpurun_out/r4t_c5.err; echo rc=$?
f=$(find gpurun_out/r4t_c5 -name "*kernel_stats.csv" | head -1); grep "k_realign" $f | cut -d, -f1-4 | sed 's/pvc3::(anonymous namespace):://' | cut -c1-140
