set -x
timeout -k 10 600 python -m pytest tests/test_host_tools.py tests/test_workloads_gpu.py -m gpu -x -q -k "cli_matches or pack or workspace or gap_score or shipped" > gpurun_out/r4k_t.log 2>&1; echo "t rc=$?"
tail -5 gpurun_out/r4k_t.log
bash tools/profile_cli.sh r04c > gpurun_out/r4k_profcli.log 2>&1; echo "profcli rc=$?"
grep -E "Record path|Stages" gpurun_out/r4k_profcli.log
timeout -k 10 400 python tools/cli_records_bench.py --skip-digest > gpurun_out/r4k_cli_bench.log 2>&1; echo "rc=$?"
grep -v "Timeline" gpurun_out/r4k_cli_bench.log | head -12
