cd $GRAFT_REPO_ROOT
PVC3_LIB_PATH=panvc3_amd/_build/diag_pf/libpvc3_hip.so timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2 &&
for n in main pf main pf; do
  if [ $n = main ]; then L=; else L=panvc3_amd/_build/diag_$n/libpvc3_hip.so; fi
  PVC3_LIB_PATH=$L timeout -k 10 200 python3 bench.py --no-build --steps 10 --warmup 2 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg 2>gpurun_out/run_$n.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d['kernel_ms']['project'])" || echo "$n failed"
done
