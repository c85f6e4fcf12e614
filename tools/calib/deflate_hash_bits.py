"""k_deflate_blocks with hash tables of 2^11 and 2^12 entries (-DPVC3_DEFLATE_HASH_BITS): the table is 16 of the 19.6 KB of LDS a block
being compressed holds, and LDS is what the command line's kernels share; against the size of the output.
usage (on the GPU box): python tools/calib/deflate_hash_bits.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
for bits in (12, 11):		# (the code construction's scratch needs 2^11 dwords of the table)
	lib = os.path.join(here, "libhash%d.so" % bits)
	subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-shared", "-DPVC3_DEFLATE_HASH_BITS=%d" % bits, "-o", lib,
						   os.path.join(root, "panvc3_amd", "bgzf", "deflate.hip"), os.path.join(root, "panvc3_amd", "bgzf", "inflate.hip")])
	print("hash bits %d:" % bits, flush=True)
	p = subprocess.run([sys.executable, os.path.join(root, "tools", "bgzf_bench.py")], env=dict(os.environ, PVC3_BGZF_LIB=lib), capture_output=True)
	print("\n".join("    " + l for l in p.stdout.decode().split("\n") if l.startswith("GPU:") or "round trip" in l), flush=True)
	if p.returncode: print(p.stderr.decode()[-2000:])
