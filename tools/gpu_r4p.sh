set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/cli_once.py --repeat 3 > gpurun_out/r4p_once.log 2>&1; echo rc=$?
timeout -k 10 300 python tools/cli_once.py --repeat 3 PVC3_WRITE_PARTS=16 >> gpurun_out/r4p_once.log 2>&1; echo rc=$?
timeout -k 10 300 python tools/cli_once.py --repeat 3 PVC3_WRITE_PARTS=4 >> gpurun_out/r4p_once.log 2>&1; echo rc=$?
grep "M records\|Stages" gpurun_out/r4p_once.log
