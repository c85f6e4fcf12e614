set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_bgzf_gpu.py -m gpu -x -q > gpurun_out/r4r_tests.log 2>&1; echo rc=$?
tail -3 gpurun_out/r4r_tests.log
timeout -k 10 400 python tools/calib/inf_variants/phases.py > gpurun_out/r4r_phases.log 2>&1; echo rc=$?
tail -2 gpurun_out/r4r_phases.log
timeout -k 10 500 python tools/calib/inf_variants/cycles.py > gpurun_out/r4r_cycles.log 2>&1; echo rc=$?
tail -3 gpurun_out/r4r_cycles.log
