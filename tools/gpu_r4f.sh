set -x
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4f_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r4f_tests.log
timeout -k 10 600 python tools/cli_records_bench.py --skip-digest > gpurun_out/r4f_cli_bench.log 2>&1; echo "rc=$?"
grep -v "Timeline" gpurun_out/r4f_cli_bench.log | head -24
