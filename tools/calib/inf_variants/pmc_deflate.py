# One deflate launch of 2 048 blocks for a PMC pass (run under rocprofv3; never starts a compiler or a child process).
import sys, gzip, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
from panvc3_amd import bgzf
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam", "rb").read()
n_have = len(raw) // bgzf.BLOCK
data = (raw[:n_have * bgzf.BLOCK] * (-(-2048 // n_have)))[:2048 * bgzf.BLOCK]
comp = bgzf.BlockCompressor(0, max_blocks=2048, n_slots=1)
ms, size, n_in = comp.timed(data, repeats=2)
print("2048 blocks %.3f ms, %d bytes in, %d out" % (ms, n_in, size))
comp.close()
