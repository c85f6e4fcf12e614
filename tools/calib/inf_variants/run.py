"""Where k_inflate_blocks' time goes, by elimination: the kernel built without its match copies, without its literal stores and
without both (the decoding itself does not depend on them), each timed on the same 219 blocks of zlib level 6 streams.
pmc.py next to this file is the same launch for a rocprofv3 --pmc pass (SQ_INSTS_SALU / _VALU / _LDS, SQ_WAVE_CYCLES).
usage (on the GPU box): python tools/calib/inf_variants/run.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
for name, flags in (("base", []), ("no_copy", ["-DPVC3_INF_SKIP_COPY"]), ("no_literals", ["-DPVC3_INF_SKIP_LIT"]), ("neither", ["-DPVC3_INF_SKIP_COPY", "-DPVC3_INF_SKIP_LIT"]), ("zcount", ["-DPVC3_INF_COUNT"])):
	subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-shared"] + flags
						  + ["-o", os.path.join(here, "lib%s.so" % name), os.path.join(root, "panvc3_amd", "bgzf", "deflate.hip"), os.path.join(root, "panvc3_amd", "bgzf", "inflate.hip")])
code = r'''
import sys, zlib, gzip, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from panvc3_amd import bgzf
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam","rb").read()
pieces = [raw[i*65280:(i+1)*65280] for i in range(min(512, len(raw)//65280))]
streams = []
for p in pieces:
    c = zlib.compressobj(6, zlib.DEFLATED, -15); streams.append(c.compress(p) + c.flush())
inf = bgzf.BlockInflater(0, max_in=len(pieces)*66000, max_out=len(pieces)*65280, max_blocks=len(pieces), n_slots=1)
got, status, ms = inf.inflate(streams, [len(p) for p in pieces], repeats=3)
print(os.environ.get("PVC3_BGZF_LIB","default").split("/")[-1], "%%d blocks %%.3f ms" %% (len(pieces), ms), "statuses", sorted(set(int(x) for x in status)))
if "zcount" in os.environ.get("PVC3_BGZF_LIB", ""):
    import struct
    tot = [0] * 5
    for g in got:
        for k, v in enumerate(struct.unpack("<5I", g[:20])): tot[k] += v
    print("per block: passes %%.0f, tokens the long way %%.0f, literals %%.0f, matches in the walk %%.0f, deflate blocks %%.1f" %% tuple(t / len(got) for t in tot))
''' % (root, root)
sys.path.insert(0, os.path.join(root, "tools"))
import cli_e2e
cli_e2e.prepare("/tmp/pvc3_bgzf_bench", 0.05, 1)
for name in sorted(os.listdir(here)):
    if name.endswith(".so"):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PVC3_BGZF_LIB=os.path.join(here, name)))
