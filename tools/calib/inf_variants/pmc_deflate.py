# One deflate launch of 2 048 blocks for a PMC pass (run under rocprofv3; never starts a compiler or a child process).
# PMC_BLOCKS: blocks of the launch; PMC_KEEP=1: the compressor is not closed before the interpreter ends (tools/calib/abort_probe.sh).
import sys, gzip, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
from panvc3_amd import bgzf
blocks = int(os.environ.get("PMC_BLOCKS", "2048"))
raw = gzip.open("/tmp/pvc3_bgzf_bench/in.bam", "rb").read()
n_have = len(raw) // bgzf.BLOCK
data = (raw[:n_have * bgzf.BLOCK] * (-(-blocks // n_have)))[:blocks * bgzf.BLOCK]
comp = bgzf.BlockCompressor(0, max_blocks=blocks, n_slots=1)
ms, size, n_in = comp.timed(data, repeats=2)
print("%d blocks %.3f ms, %d bytes in, %d out" % (blocks, ms, n_in, size), flush=True)
if "1" != os.environ.get("PMC_KEEP"):
	comp.close()
print("closed" if "1" != os.environ.get("PMC_KEEP") else "kept", flush=True)
