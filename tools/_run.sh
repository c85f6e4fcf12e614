cd $GRAFT_REPO_ROOT
for c in C3G; do for n in main mnocmp mnone; do
  if [ $n = main ]; then L=; else L=panvc3_amd/_build/diag_$n/libpvc3_hip.so; fi
  PVC3_LIB_PATH=$L timeout -k 10 300 python3 bench.py --config $c --no-build --steps 3 --warmup 1 --cpu-baseline-seconds 0 --skip-host-buffers --skip-sorted-leg 2>gpurun_out/run_$c.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$c $n', d['ms_per_step'], d['kernel_ms']['mid'], d['kernel_ms']['project'])" || echo "$c failed"
done; done
