# One inflate launch of 219 blocks for a PMC pass (run under rocprofv3; never starts a compiler or a child process).
import sys, zlib, gzip, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
from panvc3_amd import bgzf
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam", "rb").read()
pieces = [raw[i * 65280:(i + 1) * 65280] for i in range(min(512, len(raw) // 65280))]
streams = []
for p in pieces:
	c = zlib.compressobj(6, zlib.DEFLATED, -15); streams.append(c.compress(p) + c.flush())
inf = bgzf.BlockInflater(0, max_in=len(pieces) * 66000, max_out=len(pieces) * 65280, max_blocks=len(pieces), n_slots=1)
got, status, ms = inf.inflate(streams, [len(p) for p in pieces], repeats=2)
print("%d blocks %.3f ms" % (len(pieces), ms))
inf.close()
