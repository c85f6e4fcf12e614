cd $GRAFT_REPO_ROOT
true &&
for n in main nopair; do
  if [ $n = main ]; then O=; else O="--option paired_blocks=0"; fi
  timeout -k 10 200 python3 bench.py --no-build --steps 10 --warmup 2 --cpu-baseline-seconds 0 --skip-host-buffers $O 2>gpurun_out/run_$n.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d.get('kernel_ms'), d['roofline']['frac'], d.get('sorted_input',{}).get('ms_per_step'), d.get('sorted_input',{}).get('kernel_ms',{}).get('project'))" || echo "$n failed"
done
