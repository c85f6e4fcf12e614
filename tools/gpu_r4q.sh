set -x
mkdir -p gpurun_out
( PVC3_PROBE_SMALL_FIRST=1 python tools/calib/rss_probe.py; date +%s.%N ) > gpurun_out/r4q_rss.log 2>&1
cat gpurun_out/r4q_rss.log
