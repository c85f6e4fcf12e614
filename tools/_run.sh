cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2 &&
for c in C3G C3; do for n in main; do
  timeout -k 10 300 python3 bench.py --config $c --no-build --steps 5 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg 2>gpurun_out/run_$c.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$c $n', d['ms_per_step'], d['kernel_ms']['mid'], d['kernel_ms']['project'])" || echo "$c failed"
done; done
