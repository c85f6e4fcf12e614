set -x
bash tools/profile_cli.sh r04 > gpurun_out/r4i_profcli.log 2>&1; echo "profcli rc=$?"
tail -45 gpurun_out/r4i_profcli.log
