set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4m_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/r4m_tests.log
timeout -k 10 400 python tools/calib/inf_variants/phases.py > gpurun_out/r4m_phases.log 2>&1; echo rc=$?
cat gpurun_out/r4m_phases.log | tail -8
