set -x
python -m pytest tests/test_host_tools.py tests/test_workloads_gpu.py -m gpu -x -q -k "cli or pipeline or malformed" > gpurun_out/r4a_tests.log 2>&1; echo "tests rc=$?" 
tail -3 gpurun_out/r4a_tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r4a_bench.err
tools/calib/abort_probe.sh gpurun_out/abort_probe
