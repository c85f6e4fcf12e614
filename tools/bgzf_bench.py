#!/usr/bin/env python3
"""Throughput of the GPU's BGZF block compressor on BAM records (the records of tools/cli_e2e.py, uncompressed):
kernel time of one group of blocks, the size of its output, and zlib / the host's encoder on the same bytes for scale.
usage: tools/bgzf_bench.py [--blocks 2048] [--scale 0.05]"""
import argparse
import gzip
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--blocks", type=int, default=2048)
	ap.add_argument("--scale", type=float, default=0.05)
	ap.add_argument("--workdir", default="/tmp/pvc3_bgzf_bench")
	args = ap.parse_args()
	import cli_e2e
	from panvc3_amd import bgzf, build
	cli_e2e.prepare(args.workdir, args.scale, 1)
	raw = gzip.open(os.path.join(args.workdir, "in.bam"), "rb").read()
	n_have = len(raw) // bgzf.BLOCK
	data = (raw[:n_have * bgzf.BLOCK] * (-(-args.blocks // n_have)))[:args.blocks * bgzf.BLOCK]
	comp = bgzf.BlockCompressor(0, max_blocks=args.blocks, n_slots=1)
	ms, size, n_in = comp.timed(data, repeats=5)
	print("GPU: %d blocks, %.1f MB in %.3f ms = %.1f GB/s, ratio %.4f" % (args.blocks, n_in / 1e6, ms, n_in / ms / 1e6, size / n_in))
	for blocks in (256, 512, 1024):
		if blocks < args.blocks:
			ms_b, _, n_b = comp.timed(data[:blocks * bgzf.BLOCK], repeats=3)
			print("GPU: %d blocks: %.3f ms = %.1f GB/s" % (blocks, ms_b, n_b / ms_b / 1e6))
	t = time.perf_counter()
	streams = comp.deflate(data[:256 * bgzf.BLOCK])
	dt = time.perf_counter() - t
	ok = all(zlib.decompress(s, -15) == data[i * bgzf.BLOCK:(i + 1) * bgzf.BLOCK] for i, s in enumerate(streams))
	print("round trip of 256 blocks through zlib's inflate: %s (host buffers in and out: %.1f ms)" % ("ok" if ok else "MISMATCH", 1e3 * dt))
	# the other direction: the same blocks inflated on the GPU, from zlib's streams (what samtools writes) and from the GPU's own
	n_inf = min(args.blocks, 2048)
	pieces = [data[i * bgzf.BLOCK:(i + 1) * bgzf.BLOCK] for i in range(n_inf)]
	unique = {}
	def level6(p):
		if p not in unique:
			c = zlib.compressobj(6, zlib.DEFLATED, -15)
			unique[p] = c.compress(p) + c.flush()
		return unique[p]
	z_streams = [level6(p) for p in pieces]
	own = comp.deflate(data[:n_inf * bgzf.BLOCK])
	inf = bgzf.BlockInflater(0, max_in=n_inf * 66000, max_out=n_inf * bgzf.BLOCK, max_blocks=n_inf, n_slots=1)
	for name, streams in (("zlib level 6", z_streams), ("the GPU's encoder", own)):
		for count in (n_inf, 512, 128):
			got, status, ms = inf.inflate(streams[:count], [len(p) for p in pieces[:count]], repeats=3)
			good = (not status.any()) and got == pieces[:count]
			n_out = sum(len(p) for p in pieces[:count])
			print("GPU inflate, streams of %s: %d blocks, %.1f MB out of %.1f MB in %.3f ms = %.1f GB/s %s"
				  % (name, count, n_out / 1e6, sum(len(x) for x in streams[:count]) / 1e6, ms, n_out / ms / 1e6, "ok" if good else "MISMATCH"))
			ok = ok and good
	inf.close()
	sample = data[:64 * bgzf.BLOCK]
	t = time.perf_counter()
	for x in z_streams[:64]:
		zlib.decompress(x, -15)
	print("zlib inflate, one core: %.1f MB/s" % (len(sample) / (time.perf_counter() - t) / 1e6))
	for level in (1, 6):
		t = time.perf_counter()
		n = sum(len(zlib.compress(sample[o:o + bgzf.BLOCK], level)) - 6 for o in range(0, len(sample), bgzf.BLOCK))
		dt = time.perf_counter() - t
		print("zlib level %d, one core: %.1f MB/s, ratio %.4f" % (level, len(sample) / dt / 1e6, n / len(sample)))
	build.build_cpp_tests()
	p = subprocess.run([os.path.join(ROOT, "bin", "test_fast_deflate_test")], input=sample, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
					   env=dict(os.environ, PVC3_DEFLATE_TIMING="1"))
	print("host encoder (fast_deflate.hpp), one core:", p.stderr.decode().strip())
	comp.close()
	if not ok:
		raise SystemExit(1)


if __name__ == "__main__":
	main()
