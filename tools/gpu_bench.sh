set -x
mkdir -p gpurun_out
( time timeout -k 10 1100 python bench.py ) > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; echo rc=$?
tail -c 6000 gpurun_out/bench_now.json
tail -5 gpurun_out/bench_now.err
