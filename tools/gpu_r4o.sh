set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_host_tools.py tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4o_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/r4o_tests.log
timeout -k 10 600 python tools/cli_inflight_sweep.py > gpurun_out/r4o_sweep.log 2>&1; echo rc=$?
cat gpurun_out/r4o_sweep.log
