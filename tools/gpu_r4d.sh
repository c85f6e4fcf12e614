set -x
timeout -k 10 600 python -m pytest tests/test_host_tools.py -m gpu -x -q -k "cli_matches" > gpurun_out/r4d_cli.log 2>&1; echo "cli rc=$?"
tail -30 gpurun_out/r4d_cli.log
timeout -k 10 600 python -m pytest tests/test_workloads_gpu.py -m gpu -x -q -k "tiebreak or indel_run_shapes" > gpurun_out/r4d_tb.log 2>&1; echo "tb rc=$?"
tail -30 gpurun_out/r4d_tb.log
timeout -k 10 900 python tools/cli_records_bench.py --skip-digest > gpurun_out/r4d_cli_bench.log 2>&1; echo "rc=$?"
grep -v "Timeline" gpurun_out/r4d_cli_bench.log | tail -40
