set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_host_tools.py tests/test_bgzf_gpu.py tests/test_parity_kit.py -m gpu -x -q > gpurun_out/r4p_tests.log 2>&1; echo rc=$?
tail -3 gpurun_out/r4p_tests.log
timeout -k 10 300 python tools/cli_once.py --repeat 3 > gpurun_out/r4p_once.log 2>&1; echo rc=$?
grep "M records\|Timeline\|Stages" gpurun_out/r4p_once.log
