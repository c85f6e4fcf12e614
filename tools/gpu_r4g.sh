set -x
timeout -k 10 600 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4g_bgzf.log 2>&1; echo "bgzf rc=$?"
tail -8 gpurun_out/r4g_bgzf.log
timeout -k 10 900 python tools/calib/inf_variants/ring.py > gpurun_out/r4g_ring.log 2>&1; echo "ring rc=$?"
tail -8 gpurun_out/r4g_ring.log
