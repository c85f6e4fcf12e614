set -x
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4e_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r4e_tests.log
