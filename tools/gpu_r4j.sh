set -x
timeout -k 10 600 python -m pytest tests/test_host_tools.py tests/test_parity_kit.py -m gpu -x -q > gpurun_out/r4j_cli.log 2>&1; echo "cli rc=$?"
tail -5 gpurun_out/r4j_cli.log
bash tools/profile_cli.sh r04b > gpurun_out/r4j_profcli.log 2>&1; echo "profcli rc=$?"
grep -E "Record path|Stages" gpurun_out/r4j_profcli.log
timeout -k 10 400 python tools/cli_records_bench.py --skip-digest > gpurun_out/r4j_cli_bench.log 2>&1; echo "rc=$?"
grep -v "Timeline" gpurun_out/r4j_cli_bench.log | head -12
