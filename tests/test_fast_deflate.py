"""panvc3_amd/host/fast_deflate.hpp (the BGZF writer's own DEFLATE encoder) against zlib's inflate: every stream must be a
valid raw DEFLATE stream that inflates to its input.  No reference counterpart (the reference writes BAM through zlib)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "test_fast_deflate_test")


@pytest.fixture(scope="module")
def tool():
	from panvc3_amd import build
	build.build_cpp_tests()
	assert os.path.exists(TOOL)
	return TOOL


def _round_trip(tool, data):
	out = subprocess.run([tool], input=data, stdout=subprocess.PIPE, check=True).stdout
	pos, got, sizes = 0, [], []
	while pos < len(out):
		(n,) = struct.unpack_from("<I", out, pos)
		d = zlib.decompressobj(-15)
		piece = d.decompress(out[pos + 4:pos + 4 + n]) + d.flush()
		assert d.eof and not d.unused_data		# one complete final block, nothing behind it
		got.append(piece); sizes.append(n)
		pos += 4 + n
	assert b"".join(got) == data
	return sum(sizes)


def test_edge_cases(tool):
	rng = np.random.default_rng(3)
	for data in (b"", b"a", b"ab", b"abcdefg", b"aaaaaaaa", b"a" * 70000, bytes(range(256)) * 3, b"abcd" * 20000,
				 bytes(rng.integers(0, 256, size=65280, dtype=np.uint8)), bytes(rng.integers(0, 256, size=65281, dtype=np.uint8)),
				 bytes(rng.integers(0, 256, size=9, dtype=np.uint8)), bytes(rng.integers(0, 2, size=200000, dtype=np.uint8))):
		size = _round_trip(tool, data)
		assert size <= len(data) + 80 * (1 + len(data) // 0xff00)		# never much larger than stored


def test_skewed_frequencies_need_length_limiting(tool):
	# Fibonacci-like symbol frequencies: an unrestricted Huffman code would be deeper than 15 bits.
	fib = [1, 1]
	while len(fib) < 24:
		fib.append(fib[-1] + fib[-2])
	rng = np.random.default_rng(5)
	sym = np.repeat(np.arange(24, dtype=np.uint8), fib)
	rng.shuffle(sym)
	data = bytes(sym[:65000])
	_round_trip(tool, data)
	# the same skew in the match distances and lengths: repeated text at growing distances
	words = [bytes(rng.integers(97, 123, size=int(rng.integers(4, 40)), dtype=np.uint8)) for _ in range(200)]
	text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=6000) % 200)
	assert _round_trip(tool, text) < len(text) // 2


def test_bam_like_records_compress(tool):
	rng = np.random.default_rng(9)
	recs = []
	for i in range(3000):
		name = b"read_%07d/1\0" % i
		seq = bytes(rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=75))
		qual = bytes(rng.choice(np.array([2, 11, 25, 37], dtype=np.uint8), size=150, p=[0.02, 0.08, 0.2, 0.7]))
		recs.append(struct.pack("<iiBBHHHiiii", 0, 1000 + 37 * i, len(name), 60, 4681, 1, 99, 150, 0, 1300 + 37 * i, 300) + name + struct.pack("<I", 150 << 4) + seq + qual + b"NMC\0ASC\x96")
	data = b"".join(recs)
	size = _round_trip(tool, data)
	ref = len(zlib.compress(data, 6))
	assert size < 1.25 * ref		# within a quarter of zlib's default level
