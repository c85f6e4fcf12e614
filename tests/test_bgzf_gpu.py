"""The GPU's BGZF block compressor (panvc3_amd/bgzf/deflate.hip through include/pvc3_bgzf.h) against zlib's inflate: every
stream must be one complete raw DEFLATE stream that inflates to its input, the same bytes on every run.  No reference
counterpart (the reference writes BAM through seqan3 / zlib), so what is pinned here is the format, by zlib."""
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_the_header():
	"""CPU: libpvc3_bgzf.so loads and exports every function include/pvc3_bgzf.h declares."""
	import re
	from panvc3_amd import bgzf, build
	build.build_bgzf()
	declared = set(re.findall(r"PVC3_BGZF_API [^;]*?\b(pvc3_bgzf_\w+)\(", open(os.path.join(ROOT, "include", "pvc3_bgzf.h")).read()))
	assert declared == set(bgzf.SYMBOLS)
	L = bgzf.lib()
	for name in declared:
		assert hasattr(L, name), name


def _inflate(stream):
	d = zlib.decompressobj(-15)
	out = d.decompress(stream) + d.flush()
	assert d.eof and not d.unused_data		# one complete final block, nothing behind it
	return out


def _bam_like(rng, n_records):
	"""Records shaped like BAM's: a fixed part, a name, CIGAR, packed bases, qualities in runs, tags."""
	out = []
	pos = 0
	for i in range(n_records):
		pos += int(rng.integers(1, 400))
		l_seq = 150
		qual = np.repeat(rng.choice(np.array([2, 11, 25, 37], dtype=np.uint8), size=40), rng.integers(1, 12, size=40))[:l_seq]
		qual = np.pad(qual, (0, l_seq - len(qual)), constant_values=37)
		rec = (np.array([0, pos, 0x1e0a, 0x00630001, l_seq, -1, -1, 0], dtype="<i4").tobytes() + b"read%07d\0" % i
			   + np.array([150 << 4], dtype="<u4").tobytes() + rng.integers(0, 256, size=75, dtype=np.uint8).tobytes() + qual.tobytes()
			   + b"ASc\xfbXSc\xecYTZUU\0")
		out.append(np.array([len(rec)], dtype="<i4").tobytes() + rec)
	return b"".join(out)


@pytest.fixture(scope="module")
def gpu():
	from panvc3_amd import bgzf, build
	build.build_bgzf()
	c = bgzf.BlockCompressor(0, max_blocks=64, n_slots=2)
	yield c
	c.close()


@pytest.mark.gpu
def test_edge_cases(gpu):
	rng = np.random.default_rng(3)
	cases = (b"", b"a", b"ab", b"abc", b"abcd", b"abcdefg", b"aaaaaaaa", b"a" * 70000, bytes(range(256)) * 3, b"abcd" * 20000,
			 bytes(rng.integers(0, 256, size=65280, dtype=np.uint8)), bytes(rng.integers(0, 256, size=65281, dtype=np.uint8)),
			 bytes(rng.integers(0, 256, size=9, dtype=np.uint8)), bytes(rng.integers(0, 2, size=200000, dtype=np.uint8)),
			 bytes(rng.integers(0, 256, size=63, dtype=np.uint8)), bytes(rng.integers(0, 256, size=64, dtype=np.uint8)),
			 bytes(rng.integers(0, 256, size=65, dtype=np.uint8)), b"ab" * 129 + b"c", b"x" * 258, b"x" * 259, b"x" * 260, b"x" * 65280)
	for data in cases:
		streams = gpu.deflate(data)
		assert len(streams) == max(1, -(-len(data) // 65280))
		assert b"".join(_inflate(s) for s in streams) == data
		assert sum(len(s) for s in streams) <= len(data) + 5 * len(streams)		# never larger than stored
	# a block of incompressible bytes is stored: five bytes more than the input
	assert len(gpu.deflate(cases[10])[0]) == 65285


@pytest.mark.gpu
def test_every_small_piece_size(gpu):
	"""Pieces of 1 .. 300 bytes of the same text: the last window of a block ends at every offset of a wave."""
	rng = np.random.default_rng(11)
	text = bytes(rng.choice(np.frombuffer(b"ACGTN!#5F", dtype=np.uint8), size=40000))
	for piece in list(range(1, 70)) + [127, 128, 129, 255, 256, 257, 258, 259, 300]:
		data = text[:piece * 40]
		streams = gpu.deflate(data, piece=piece)
		assert len(streams) == 40
		assert b"".join(_inflate(s) for s in streams) == data


@pytest.mark.gpu
def test_skewed_frequencies_need_length_limiting(gpu):
	fib = [1, 1]
	while len(fib) < 24:
		fib.append(fib[-1] + fib[-2])
	rng = np.random.default_rng(5)
	sym = np.repeat(np.arange(24, dtype=np.uint8), fib)
	rng.shuffle(sym)
	data = bytes(sym[:65000])
	(s,) = gpu.deflate(data)
	assert _inflate(s) == data
	words = [bytes(rng.integers(97, 123, size=int(rng.integers(4, 40)), dtype=np.uint8)) for _ in range(200)]
	text = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=6000) % 200)
	streams = gpu.deflate(text)
	assert b"".join(_inflate(s) for s in streams) == text
	assert sum(len(s) for s in streams) < len(text) // 2


@pytest.mark.gpu
def test_bam_records_round_trip_and_are_deterministic(gpu):
	from panvc3_amd import bgzf
	rng = np.random.default_rng(7)
	data = _bam_like(rng, 20000)		# about 5 MB, more blocks than the handle's group holds at once
	streams = gpu.deflate(data)
	assert b"".join(_inflate(s) for s in streams) == data
	assert streams == gpu.deflate(data)
	total = sum(len(s) for s in streams)
	level1 = sum(len(zlib.compress(data[o:o + 65280], 1)) for o in range(0, len(data), 65280))
	assert total < 1.05 * level1		# not worse than zlib's fastest level
	# as BGZF members the blocks are a gzip file
	import gzip
	import io
	blocks = b"".join(bgzf.bgzf_member(s, data[i * 65280:(i + 1) * 65280]) for i, s in enumerate(streams))
	assert gzip.GzipFile(fileobj=io.BytesIO(blocks)).read() == data


@pytest.mark.gpu
def test_slots_are_independent(gpu):
	"""Two groups in flight through the slot interface."""
	import ctypes as C
	from panvc3_amd import bgzf
	L = bgzf.lib()
	rng = np.random.default_rng(13)
	datas = [_bam_like(rng, 700), _bam_like(rng, 900)]
	counts = []
	for slot, data in enumerate(datas):
		inp = np.ctypeslib.as_array(C.cast(L.pvc3_bgzf_slot_input(gpu._h, slot), C.POINTER(C.c_uint8)), shape=(64 * bgzf.IN_STRIDE,))
		lens = np.ctypeslib.as_array(C.cast(L.pvc3_bgzf_slot_lengths(gpu._h, slot), C.POINTER(C.c_uint32)), shape=(64,))
		n = -(-len(data) // 65280)
		for i in range(n):
			chunk = np.frombuffer(data[i * 65280:(i + 1) * 65280], dtype=np.uint8)
			inp[i * bgzf.IN_STRIDE + bgzf.IN_PAD:i * bgzf.IN_STRIDE + bgzf.IN_PAD + len(chunk)] = chunk
			lens[i] = len(chunk)
		assert 0 == L.pvc3_bgzf_submit(gpu._h, slot, n)
		counts.append(n)
	assert 0 != L.pvc3_bgzf_submit(gpu._h, 0, 1)		# busy until waited for
	for slot, data in enumerate(datas):
		out, out_len, ms = C.c_void_p(), C.c_void_p(), C.c_float()
		assert 0 == L.pvc3_bgzf_wait(gpu._h, slot, C.byref(out), C.byref(out_len), C.byref(ms))
		ol = np.ctypeslib.as_array(C.cast(out_len, C.POINTER(C.c_uint32)), shape=(counts[slot],))
		o = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(counts[slot] * bgzf.OUT_STRIDE,))
		got = b"".join(_inflate(o[i * bgzf.OUT_STRIDE:i * bgzf.OUT_STRIDE + int(ol[i])].tobytes()) for i in range(counts[slot]))
		assert got == data
		assert ms.value > 0


@pytest.mark.gpu
def test_random_structures(gpu):
	"""Blocks of many shapes — alphabets of 1 to 256 symbols, copies at every distance class up to and beyond 32 768, matches of
	every length up to and beyond 258, lengths that end a window anywhere — each stream inflated by zlib."""
	rng = np.random.default_rng(23)
	blocks = []
	for k in range(160):
		n = int(rng.integers(1, 65281))
		kind = k % 8
		if kind == 0:		# few symbols, long runs
			data = np.repeat(rng.integers(0, int(rng.integers(1, 5)), size=n // 3 + 2, dtype=np.uint8), rng.integers(1, 9, size=n // 3 + 2))[:n]
		elif kind == 1:		# uniform over a small alphabet
			data = rng.integers(0, int(rng.integers(2, 257)), size=n, dtype=np.uint8)
		elif kind == 2:		# a phrase book: copies at all distances
			words = [rng.integers(0, 256, size=int(rng.integers(3, 400)), dtype=np.uint8) for _ in range(int(rng.integers(2, 60)))]
			out, total = [], 0
			while total < n:
				w = words[int(rng.integers(0, len(words)))]
				out.append(w); total += len(w)
			data = np.concatenate(out)[:n]
		elif kind == 3:		# one period, any length
			period = rng.integers(0, 256, size=int(rng.integers(1, 40000)), dtype=np.uint8)
			data = np.tile(period, n // len(period) + 1)[:n]
		elif kind == 4:		# random bytes with a copy of the beginning exactly `d` back, d around the window limit
			data = rng.integers(0, 256, size=n, dtype=np.uint8)
			d = int(rng.choice([32766, 32767, 32768, 32769, 32770, 40000]))
			if n > d + 600:
				data[d:d + 600] = data[:600]
		elif kind == 5:		# matches of chosen lengths separated by fresh bytes
			piece = rng.integers(0, 256, size=300, dtype=np.uint8)
			out = [piece]
			for length in (3, 4, 5, 8, 9, 10, 11, 12, 130, 131, 226, 227, 257, 258, 259, 260, 300):
				out += [rng.integers(0, 256, size=7, dtype=np.uint8), piece[:length]]
			data = np.concatenate(out)[:n]
		elif kind == 6:		# skewed: geometric symbol frequencies
			data = np.minimum(rng.geometric(0.3, size=n), 255).astype(np.uint8)
		else:				# text-like
			data = rng.choice(np.frombuffer(b"etaoin shrdlu\n", dtype=np.uint8), size=n)
		blocks.append(np.ascontiguousarray(data, dtype=np.uint8).tobytes())
	for data in blocks:
		(s,) = gpu.deflate(data)
		assert _inflate(s) == data
		assert len(s) <= len(data) + 5


# ---- the other direction: pvc3_bgzf_inflater_* (panvc3_amd/bgzf/inflate.hip) against zlib-made streams -------------------------------

def _raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
	c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
	return c.compress(data) + c.flush()


@pytest.fixture(scope="module")
def gpu_inflater():
	from panvc3_amd import bgzf, build
	build.build_bgzf()
	c = bgzf.BlockInflater(0, max_in=24 << 20, max_out=48 << 20, max_blocks=2048, n_slots=2)
	yield c
	c.close()


def _inflate_cases():
	rng = np.random.default_rng(29)
	text = bytes(rng.choice(np.frombuffer(b"etaoin shrdlu\n", dtype=np.uint8), size=65536))
	cases = [b"", b"a", b"ab", b"abc" * 5, b"x" * 65536, bytes(range(256)) * 256, text, bytes(rng.integers(0, 256, size=65536, dtype=np.uint8)),
			 bytes(rng.integers(0, 4, size=40000, dtype=np.uint8)), _bam_like(rng, 200)[:65280], b"ab" * 129 + b"c", b"q" * 257, b"q" * 258, b"q" * 259]
	for size in (1, 2, 7, 8, 9, 100, 257, 258, 259, 273, 1000, 32768, 32769, 65280):
		cases.append(bytes(rng.integers(0, 256, size=size, dtype=np.uint8)))
		cases.append(bytes(rng.integers(97, 101, size=size, dtype=np.uint8)))
	# copies at the distance limit, lengths around every code boundary
	far = bytearray(rng.integers(0, 256, size=65536, dtype=np.uint8))
	far[32768:33400] = far[:632]
	far[40000:40258] = far[40000 - 32768:40000 - 32768 + 258]
	cases.append(bytes(far))
	return cases


@pytest.mark.gpu
def test_inflate_zlib_streams(gpu_inflater):
	"""Streams of every block type (stored, fixed, dynamic; several blocks per stream), made by zlib at levels 0, 1, 6, 9 with the
	default, fixed, Huffman-only and run-length strategies, at any alignment in the input."""
	streams, datas = [], []
	for data in _inflate_cases():
		for level in (0, 1, 6, 9):
			for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
				streams.append(_raw_deflate(data, level, strategy))
				datas.append(data)
	for gap in (0, 1, 2, 3, 5):
		for first in range(0, len(streams), 1024):
			part, want = streams[first:first + 1024], datas[first:first + 1024]
			got, status, _ = gpu_inflater.inflate(part, [len(d) for d in want], gap=gap)
			bad = [i for i in range(len(part)) if status[i] != 0 or got[i] != want[i]]
			assert not bad, (gap, first, bad[:5], [int(status[i]) for i in bad[:5]])


@pytest.mark.gpu
def test_inflate_matches_at_every_distance(gpu_inflater):
	"""zlib's level 9 finds a copy of a block of random bytes at every distance up to 32 KB: distances around powers of two, the
	longest match at the largest distance, copies that overlap their own output, several far copies in a row, in the pinned-output
	form and in the device-resident one.  (Written for an inflater that kept only the output's last 8 KB in LDS.)"""
	rng = np.random.default_rng(41)
	cases = []
	for dist in (4000, 4095, 4096, 4097, 7800, 8000, 8191, 8192, 8193, 8450, 12000, 16384, 20000, 32000, 32768):
		for length in (3, 50, 258, 700):
			block = bytes(rng.integers(0, 256, size=length, dtype=np.uint8))
			filler = bytes(rng.integers(0, 256, size=dist - length if dist > length else 1, dtype=np.uint8))
			head = bytes(rng.integers(0, 256, size=int(rng.integers(0, 5000)), dtype=np.uint8))
			cases.append(head + block + filler + block + bytes(rng.integers(0, 4, size=100, dtype=np.uint8)))
	# the same 300 bytes again and again, 9 KB apart, with text in between that compresses (literal runs and near matches mixed in)
	block = bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
	text = _bam_like(rng, 40)
	cases.append((block + text[:8900]) * 6)
	cases.append((block + text[:16000]) * 3 + block)
	streams = [_raw_deflate(c[:65536], 9) for c in cases]
	sizes = [len(c[:65536]) for c in cases]
	got, status, _ = gpu_inflater.inflate(streams, sizes, gap=1)
	bad = [i for i in range(len(cases)) if status[i] != 0 or got[i] != cases[i][:65536]]
	assert not bad, (bad[:5], [int(status[i]) for i in bad[:5]])
	dev, st2, crcs = gpu_inflater.inflate_to_device(streams, sizes, slot=1)
	assert not st2.any() and dev == b"".join(c[:65536] for c in cases)
	assert [int(x) for x in crcs] == [zlib.crc32(c[:65536]) & 0xffffffff for c in cases]


@pytest.mark.gpu
def test_inflate_matches_that_copy_matches(gpu_inflater):
	"""What the second kernel (k_inflate_resolve_lanes) has to order: matches whose sources are other matches' output a few bytes
	back.  Periods of 1 .. 24 bytes (every match overlaps its own output and copies the one before it), texts made of copies of
	their own last few hundred bytes (64 matches in a row that read each other), lengths around the 16 bytes a lane copies by itself,
	and symbol frequencies that give codes longer than the decoder's tables index (the one-symbol path)."""
	rng = np.random.default_rng(43)
	cases = []
	for period in range(1, 25):
		unit = bytes(rng.integers(0, 256, size=period, dtype=np.uint8))
		cases.append((unit * (3000 // period + 1))[:3000] + bytes(rng.integers(0, 256, size=20, dtype=np.uint8)) + unit * 40)
	for reach, longest in ((40, 12), (300, 40), (300, 17), (2000, 300), (64, 16)):
		out = bytearray(rng.integers(0, 256, size=64, dtype=np.uint8))
		while len(out) < 60000:
			back = int(rng.integers(1, min(reach, len(out)) + 1)); n = int(rng.integers(3, longest + 1))
			for _ in range(n): out.append(out[-back])		# (a copy that may overlap itself)
			if rng.random() < 0.3: out.extend(rng.integers(0, 256, size=int(rng.integers(1, 4)), dtype=np.uint8).tobytes())
		cases.append(bytes(out[:65536]))
	ranks = np.arange(1, 257, dtype=np.float64)
	skewed = rng.choice(256, size=65536, p=(1 / ranks ** 1.6) / (1 / ranks ** 1.6).sum()).astype(np.uint8)
	cases.append(bytes(skewed))
	cases.append(bytes(skewed[:30000]) + bytes(skewed[100:30100]))
	streams, sizes, want = [], [], []
	for c in cases:
		for level in (1, 6, 9):
			streams.append(_raw_deflate(c, level)); sizes.append(len(c)); want.append(c)
	got, status, _ = gpu_inflater.inflate(streams, sizes, gap=3)
	bad = [i for i in range(len(want)) if status[i] != 0 or got[i] != want[i]]
	assert not bad, (bad[:5], [int(status[i]) for i in bad[:5]])
	dev, st2, crcs = gpu_inflater.inflate_to_device(streams, sizes, slot=1)
	assert not st2.any() and dev == b"".join(want)
	assert [int(x) for x in crcs] == [zlib.crc32(c) & 0xffffffff for c in want]


@pytest.mark.gpu
def test_inflater_slots_at_once(gpu_inflater):
	"""Both slots submitted before either is waited for: each slot has a stream, device input, token space and status of its own, so
	two groups of blocks are decoded at the same time — different data, different numbers of blocks, several rounds."""
	rng = np.random.default_rng(47)
	for round_ in range(3):
		groups, want = [], []
		for slot in range(2):
			pieces = []
			for _ in range(int(rng.integers(40, 300))):
				kind = int(rng.integers(0, 3))
				size = int(rng.integers(1, 65281))
				pieces.append(_bam_like(rng, 260)[:size] if 0 == kind else bytes(rng.integers(0, 4 if 1 == kind else 256, size=size, dtype=np.uint8)))
			groups.append(([_raw_deflate(p, int(rng.choice([1, 6, 9]))) for p in pieces], [len(p) for p in pieces]))
			want.append(pieces)
		for slot, (got, status) in enumerate(gpu_inflater.inflate_slots(groups)):
			bad = [i for i in range(len(want[slot])) if status[i] != 0 or got[i] != want[slot][i]]
			assert not bad, (round_, slot, bad[:5], [int(status[i]) for i in bad[:5]])


@pytest.mark.gpu
def test_inflate_what_the_gpu_deflated(gpu, gpu_inflater):
	rng = np.random.default_rng(31)
	data = _bam_like(rng, 6000)
	streams = gpu.deflate(data)
	pieces = [data[i * 65280:(i + 1) * 65280] for i in range(len(streams))]
	got, status, ms = gpu_inflater.inflate(streams, [len(p) for p in pieces], slot=1)
	assert not status.any() and got == pieces and ms > 0


@pytest.mark.gpu
def test_inflate_refuses_what_is_wrong(gpu_inflater):
	"""A damaged stream gets a status (or inflates to different bytes, which the caller's CRC check finds); its neighbours are not
	touched.  Truncated input, a wrong ISIZE in either direction, flipped bytes, garbage."""
	rng = np.random.default_rng(37)
	data = _bam_like(rng, 200)[:60000]
	good = _raw_deflate(data, 6)
	damaged = [(good[:len(good) // 2], len(data)), (good[:-3], len(data)), (good, len(data) - 1), (good, len(data) + 1),
			   (bytes(rng.integers(0, 256, size=5000, dtype=np.uint8)), 20000), (b"\x07" + good[1:], len(data)), (b"", 10)]
	for k in range(20):
		b = bytearray(good)
		for _ in range(3):
			b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
		damaged.append((bytes(b), len(data)))
	streams, sizes = [], []
	for s, n in damaged:
		streams += [good, s]
		sizes += [len(data), n]
	streams.append(good); sizes.append(len(data))
	got, status, _ = gpu_inflater.inflate(streams, sizes, gap=3)
	for i in range(0, len(streams), 2):
		assert status[i] == 0 and got[i] == data
	for i, (s, n) in enumerate(damaged):
		j = 2 * i + 1
		try:
			d = zlib.decompressobj(-15)
			ref = d.decompress(s) + d.flush()
			ok = d.eof and len(ref) == n and not d.unused_data
		except zlib.error:
			ok = False
		if ok:
			assert status[j] == 0 and got[j] == ref		# (a flipped bit may leave a valid stream of the same length)
		else:
			assert status[j] != 0 or got[j] != data[:n] or n != len(data)
	assert status[1] != 0 and status[3] != 0 and status[5] != 0 and status[7] != 0


@pytest.mark.gpu
def test_device_resident_compressor_makes_the_file_bytes():
	"""pvc3_bgzf_dev_*: the input lies in device memory (the record path's kernels write it there), and what comes back is the BGZF
	file — every block with its gzip / BC header, the stream, the CRC-32 computed on the device and ISIZE.  Python's gzip reads
	multi-member files and verifies both trailers, so a wrong checksum, size or header fails here."""
	import gzip
	import struct
	from panvc3_amd import bgzf
	rng = np.random.default_rng(21)
	comp = bgzf.DeviceCompressor(0, max_blocks=40, n_slots=2)
	try:
		for slot, n in ((0, 1), (1, 65279), (0, 65280), (1, 65281), (0, 3 * 65280 + 17), (1, 40 * 65280)):
			data = (_bam_like(rng, n // 280 + 2) * 2)[:n] if n > 1000 else bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
			data = (data * (n // len(data) + 1))[:n]
			blob, ms = comp.deflate(data, slot=slot)
			assert gzip.decompress(blob) == data
			# block by block: header fields, BSIZE, ISIZE
			p, total = 0, 0
			while p < len(blob):
				assert blob[p:p + 4] == b"\x1f\x8b\x08\x04" and blob[p + 10:p + 16] == b"\x06\x00BC\x02\x00"
				bsize = struct.unpack_from("<H", blob, p + 16)[0] + 1
				crc, isize = struct.unpack_from("<II", blob, p + bsize - 8)
				assert isize == min(65280, n - total) and crc == (zlib.crc32(data[total:total + isize]) & 0xffffffff)
				total += isize; p += bsize
			assert p == len(blob) and total == n
	finally:
		comp.close()
	# bytes that do not shrink: the blocks are stored, the file is larger than the page-locked memory it comes back through, and is
	# fetched in more than one piece
	big = bgzf.DeviceCompressor(0, max_blocks=600, n_slots=1)
	try:
		data = bytes(rng.integers(0, 256, size=600 * 65280 - 1234, dtype=np.uint8))
		blob, _ = big.deflate(data)
		assert len(blob) > len(data) and gzip.decompress(blob) == data
	finally:
		big.close()


@pytest.mark.gpu
def test_device_resident_inflater_and_checksums():
	"""pvc3_bgzf_inflater_submit_device: the blocks' bytes land in device memory of the caller, and every block's CRC-32 comes back
	as the device computed it (lengths of every residue modulo 64 and 4, the empty range, a refused block)."""
	from panvc3_amd import bgzf
	rng = np.random.default_rng(22)
	sizes = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1020, 4093, 65279, 65280, 65536, 30000]
	pieces = []
	for n in sizes:
		d = _bam_like(rng, n // 250 + 2)
		pieces.append((d * (n // len(d) + 1))[:n])
	streams = []
	for p in pieces:
		c = zlib.compressobj(6, zlib.DEFLATED, -15)
		streams.append(c.compress(p) + c.flush())
	streams[-1] = streams[-1][:len(streams[-1]) // 2]		# a stream that ends early: refused, whatever lies in its place
	inf = bgzf.BlockInflater(0, max_in=2 << 20, max_out=1, max_blocks=64, n_slots=2)
	try:
		for slot in (0, 1):
			got, status, crcs = inf.inflate_to_device(streams, sizes, slot=slot)
			assert list(status[:-1]) == [0] * (len(sizes) - 1) and status[-1] != 0
			o = 0
			for i, p in enumerate(pieces[:-1]):
				assert got[o:o + len(p)] == p, i
				assert int(crcs[i]) == (zlib.crc32(p) & 0xffffffff), (i, len(p))
				o += len(p)
	finally:
		inf.close()
