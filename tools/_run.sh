cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2 &&
for n in main nowin; do
  if [ $n = main ]; then O=; else O="--option text_window=0"; fi
  timeout -k 10 200 python3 bench.py --no-build --steps 10 --warmup 2 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg $O 2>gpurun_out/run_$n.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d.get('kernel_ms'), d['roofline']['frac'])" || echo "$n failed"
done
