set -x
timeout -k 10 300 python -m pytest tests/test_workloads_gpu.py -m gpu -x -q -k "destinations_named" > gpurun_out/r4h_dest.log 2>&1; echo "dest rc=$?"
tail -5 gpurun_out/r4h_dest.log
timeout -k 10 300 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4h_bgzf.log 2>&1; echo "bgzf rc=$?"
tail -3 gpurun_out/r4h_bgzf.log
bash tools/profile_cli.sh r04 > gpurun_out/r4h_profcli.log 2>&1; echo "profcli rc=$?"
tail -45 gpurun_out/r4h_profcli.log
