cd /tmp/pvc3_e2e 2>/dev/null || { cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 python3 tools/cli_e2e.py --scale 1 --repeat 4 > /dev/null 2>&1; cd /tmp/pvc3_e2e; }
P="$GRAFT_REPO_ROOT/bin/panvc3_project_alignments"
for i in 1 2 3; do
  s=$(date +%s.%N)
  $P -a in.bam -m msa-index.dat -r genome.fa -d REF -s / -b -o out.bam --debugging-output --batch-size=131072 2> err.txt
  e=$(date +%s.%N)
  echo "wall $(echo "$e - $s" | bc) | $(grep 'Done' err.txt) | $(grep Stages err.txt | cut -c1-260)"
done
