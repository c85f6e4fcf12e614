set -x
timeout -k 10 900 python tools/cli_records_bench.py > gpurun_out/r4c_cli_bench.log 2>&1; echo "rc=$?"
cat gpurun_out/r4c_cli_bench.log | tail -60
timeout -k 10 200 python tools/host_buffers_probe.py > gpurun_out/r4c_probe.log 2>&1; echo "probe rc=$?"
tail -8 gpurun_out/r4c_probe.log
