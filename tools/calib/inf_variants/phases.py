"""The two kernels of the BGZF inflater (k_inflate_decode, k_inflate_resolve_lanes) timed on zlib level-6 streams of BAM records,
256 to 4 096 blocks a launch.  profiles/r04/inflate_phases.txt is this script at the commit that still held the earlier forms
(one kernel with 32 KB of output in LDS; the first decode and resolve kernels) behind environment switches.
usage (on the GPU box): python tools/calib/inf_variants/phases.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
code = r'''
import sys, zlib, gzip, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from panvc3_amd import bgzf
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam","rb").read()
have = [raw[i*65280:(i+1)*65280] for i in range(len(raw)//65280)]
cache = {}
def z(p):
    if p not in cache:
        c = zlib.compressobj(6, zlib.DEFLATED, -15); cache[p] = c.compress(p) + c.flush()
    return cache[p]
n_max = 4096
pieces = [have[i %% len(have)] for i in range(n_max)]
streams = [z(p) for p in pieces]
inf = bgzf.BlockInflater(0, max_in=n_max*30000, max_out=n_max*65280, max_blocks=n_max, n_slots=1)
L = bgzf.lib()
line = [os.environ.get("PVC3_VARIANT", ""), "in flight %%d" %% L.pvc3_bgzf_inflater_blocks_in_flight(inf._h)]
for count in (256, 1024, 2048, 2560, 4096):
    got, status, ms = inf.inflate(streams[:count], [len(p) for p in pieces[:count]], repeats=3)
    ok = (not status.any()) and got == pieces[:count]
    line.append("%%d blocks %%.2f ms = %%.1f GB/s%%s" %% (count, ms, sum(len(p) for p in pieces[:count]) / ms / 1e6, "" if ok else " MISMATCH"))
print("; ".join(line), flush=True)
''' % (root, root)
sys.path.insert(0, os.path.join(root, "tools"))
import cli_e2e
cli_e2e.prepare("/tmp/pvc3_bgzf_bench", 0.05, 1)
subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PVC3_VARIANT="decode + resolve"))
