set -x
timeout -k 10 600 python tools/cli_inflight_sweep.py > gpurun_out/r4l_sweep.log 2>&1; echo "rc=$?"
cat gpurun_out/r4l_sweep.log | tail -30
