set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/cli_once.py --repeat 4 > gpurun_out/r4p_once.log 2>&1; echo rc=$?
timeout -k 10 300 python tools/cli_once.py --repeat 3 PVC3_BGZF_CHUNK=16777216 PVC3_RECORDS_STREAM=56623104 >> gpurun_out/r4p_once.log 2>&1; echo rc=$?
timeout -k 10 300 python tools/cli_once.py --repeat 3 PVC3_BGZF_CHUNK=33554432 PVC3_RECORDS_STREAM=113246208 >> gpurun_out/r4p_once.log 2>&1; echo rc=$?
grep "M records\|Stages" gpurun_out/r4p_once.log
bash tools/profile_cli.sh r4p > gpurun_out/r4p_prof.log 2>&1; echo rc=$?
