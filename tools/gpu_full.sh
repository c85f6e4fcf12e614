set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; echo rc=$?
tail -4 gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke rc=$?
tail -2 gpurun_out/smoke.log
