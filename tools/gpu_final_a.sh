set -x
mkdir -p gpurun_out
timeout -k 10 600 python tools/cli_records_bench.py > gpurun_out/fa_cli_records.log 2>&1; echo rc=$?
cat gpurun_out/fa_cli_records.log | grep -v "Timeline" | cut -c1-420
timeout -k 10 300 python tools/bgzf_bench.py > gpurun_out/fa_bgzf_bench.log 2>&1; echo rc=$?
cat gpurun_out/fa_bgzf_bench.log
