set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/full_tests.log
