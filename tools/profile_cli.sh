#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_cli.sh NAME — the command line BAM -> BAM (16 M records of config C2's
# reads, the record path on the GPU) under rocprofv3 --kernel-trace --stats: what the kernels of the record path and of the two
# BGZF codecs take per piece of the file.  Output in gpurun_out/NAME_cli/.
set -e
name=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tools")
import cli_e2e
P = cli_e2e.prepare("/tmp/pvc3_prof_cli", scale=0.25, repeat=1)
cli_e2e.write_repeated(P, "/tmp/pvc3_prof_cli/in_large.bam", 64)
PY
cd /tmp/pvc3_prof_cli
export PVC3_EXIT_PROFILE=1		# the tool then ends with exit(): rocprofv3 writes its output in an exit handler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${name}_cli" -- "$GRAFT_REPO_ROOT/bin/panvc3_project_alignments" -a in_large.bam -m msa-index.dat -r genome.fa -d REF -s / -o out.bam -b --debugging-output > "$GRAFT_REPO_ROOT/gpurun_out/${name}_cli.out" 2> "$GRAFT_REPO_ROOT/gpurun_out/${name}_cli.err"
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/${name}_cli -name "*kernel_stats.csv" | head -1)
echo "== $f"; cut -d, -f1-6 "$f" | head -40
grep -E "Record path|Stages" gpurun_out/${name}_cli.err
