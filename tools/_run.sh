cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2 &&
for c in C3 C5 C3G; do
  timeout -k 10 300 python3 bench.py --config $c --no-build --steps 5 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg 2>gpurun_out/run_$c.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$c', d['ms_per_step'], d.get('kernel_ms'))" || echo "$c failed"
done
