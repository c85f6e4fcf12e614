"""Where a pass of k_inflate_decode spends its time: the kernel built with -DPVC3_INF_CYCLES reads the shader clock at the seams of a
pass (each reading costs some tens of cycles itself: the sum is larger than a pass of the plain kernel) and leaves the sums per
block where the block's output would be.  One wave per CU (256 blocks) and a full chip (4 096).
usage (on the GPU box): python tools/calib/inf_variants/cycles.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
lib = os.path.join(here, "libcycles.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-shared", "-DPVC3_INF_CYCLES", "-o", lib,
					   os.path.join(root, "panvc3_amd", "bgzf", "deflate.hip"), os.path.join(root, "panvc3_amd", "bgzf", "inflate.hip")])
code = r'''
import sys, zlib, gzip, os, struct
sys.path.insert(0, %r)
from panvc3_amd import bgzf
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam","rb").read()
have = [raw[i*65280:(i+1)*65280] for i in range(len(raw)//65280)]
def z(p):
    c = zlib.compressobj(6, zlib.DEFLATED, -15); return c.compress(p) + c.flush()
zs = [z(p) for p in have]
names = ["bits into the lanes", "first look-up, lengths", "second look-up, distances", "walk", "prefix sum", "stores", "one-symbol path", "loop top"]
for count in (256, 4096):
    pieces = [have[i %% len(have)] for i in range(count)]
    streams = [zs[i %% len(have)] for i in range(count)]
    inf = bgzf.BlockInflater(0, max_in=count*30000, max_out=count*65280, max_blocks=count, n_slots=1)
    got, status, ms = inf.inflate(streams, [len(p) for p in pieces], repeats=2)
    sums = [0] * 9
    for g in got:
        v = struct.unpack("<9Q", g[:72])
        for k in range(9): sums[k] += v[k]
    passes = sums[8]
    print("%%d blocks, %%.2f ms: %%.0f passes a block; shader-clock ticks per pass: " %% (count, ms, passes / count) + ", ".join("%%s %%.0f" %% (names[k], sums[k] / passes) for k in range(8)) + "; all %%.0f" %% (sum(sums[:8]) / passes), flush=True)
    inf.close()
''' % (root,)
sys.path.insert(0, os.path.join(root, "tools"))
import cli_e2e
cli_e2e.prepare("/tmp/pvc3_bgzf_bench", 0.05, 1)
subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PVC3_BGZF_LIB=lib))
