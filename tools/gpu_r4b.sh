set -x
python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q -k "device_resident" > gpurun_out/r4b_bgzf.log 2>&1; echo "bgzf rc=$?"
tail -15 gpurun_out/r4b_bgzf.log
timeout -k 10 600 python -m pytest tests/test_host_tools.py -m gpu -x -q -k "cli_matches" > gpurun_out/r4b_cli.log 2>&1; echo "cli rc=$?"
tail -40 gpurun_out/r4b_cli.log
timeout -k 10 300 python tools/host_buffers_probe.py > gpurun_out/r4b_probe.log 2>&1; echo "probe rc=$?"
tail -12 gpurun_out/r4b_probe.log
