set -x
mkdir -p gpurun_out
python -c "
from panvc3_amd import build; build.build(); build.build_bgzf(); build.build_host()" > gpurun_out/r04_build.log 2>&1
bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1; echo rc=$?
tail -3 gpurun_out/r04_profile.log
ls gpurun_out | grep r04
