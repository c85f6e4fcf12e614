set -x
mkdir -p gpurun_out
timeout -k 10 800 python tools/calib/deflate_hash_bits.py > gpurun_out/r4s_hash.log 2>&1; echo rc=$?
cat gpurun_out/r4s_hash.log
