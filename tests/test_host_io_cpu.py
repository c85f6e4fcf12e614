"""Host-side I/O pieces that need no GPU: per-contig FASTA access through .fai / .gzi (the reference requires both files,
libpanvc3/compressed_fasta_reader.cc:163-248) and the BGZF writer's behaviour when the last bytes cannot be written."""
import os
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "test_fasta_access_test")


@pytest.fixture(scope="module")
def tool():
	from panvc3_amd import build
	build.build_cpp_tests()
	assert os.path.exists(TOOL)
	return TOOL


def _fasta(contigs, width, final_newline=True):
	text, fai = b"", []
	for name, seq in contigs:
		text += b">" + name.encode() + b"\n"
		off = len(text)
		for i in range(0, len(seq), width):
			text += seq[i:i + width].encode() + b"\n"
		fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + 1))
	if not final_newline:
		text = text[:-1]
	return text, "".join(fai)


def _bgzf_block(data, extra_first=b""):
	"""One BGZF block; extra_first: other subfields placed BEFORE the BC subfield."""
	comp = zlib.compressobj(6, zlib.DEFLATED, -15)
	body = comp.compress(data) + comp.flush()
	xlen = len(extra_first) + 6
	bsize = 12 + xlen + len(body) + 8
	return (struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, xlen) + extra_first + b"BC" + struct.pack("<HH", 2, bsize - 1)
			+ body + struct.pack("<II", zlib.crc32(data), len(data)))


CONTIGS = [("chrA", "ACGT" * 30), ("chrB", "TTGACCA" * 9 + "GG"), ("chrC", "G" * 120)]		# chrA and chrC: whole lines only (width 60)


@pytest.mark.parametrize("final_newline", [True, False])
def test_contigs_through_fai(tool, tmp_path, final_newline):
	text, fai = _fasta(CONTIGS, 60, final_newline)
	p = tmp_path / "ref.fa"
	p.write_bytes(text)
	(tmp_path / "ref.fa.fai").write_text(fai)
	out = subprocess.run([tool, "contigs", str(p), "chrC", "chrA", "nosuch", "chrB"], capture_output=True, text=True)
	assert out.returncode == 0, out.stderr
	assert out.stdout == "".join(">%s\n%s\n" % (n, dict(CONTIGS)[n]) for n in ("chrC", "chrA", "chrB"))


@pytest.mark.parametrize("extra_first", [b"", b"XY" + struct.pack("<H", 3) + b"abc"])
def test_contigs_through_gzi(tool, tmp_path, extra_first):
	text, fai = _fasta(CONTIGS, 60, final_newline=False)
	# blocks of 100 uncompressed bytes, .gzi = pairs (compressed offset, uncompressed offset) of every block but the first
	blocks, gzi, c_off = [], [], 0
	for u in range(0, len(text), 100):
		b = _bgzf_block(text[u:u + 100], extra_first)
		if u: gzi.append((c_off, u))
		blocks.append(b); c_off += len(b)
	blocks.append(_bgzf_block(b""))
	p = tmp_path / "ref.fa.gz"
	p.write_bytes(b"".join(blocks))
	(tmp_path / "ref.fa.gz.fai").write_text(fai)
	(tmp_path / "ref.fa.gz.gzi").write_bytes(struct.pack("<Q", len(gzi)) + b"".join(struct.pack("<QQ", c, u) for c, u in gzi))
	out = subprocess.run([tool, "contigs", str(p), "chrB", "chrC"], capture_output=True, text=True)
	assert out.returncode == 0, out.stderr
	assert out.stdout == "".join(">%s\n%s\n" % (n, dict(CONTIGS)[n]) for n in ("chrB", "chrC"))


def test_writer_reports_a_failed_close(tool):
	if not os.path.exists("/dev/full"):
		pytest.skip("no /dev/full")
	out = subprocess.run([tool, "full-device"], capture_output=True, text=True)
	assert out.returncode == 3, (out.returncode, out.stderr)


def test_fast_crc32_equals_zlib(tool):
	out = subprocess.run([os.path.join(ROOT, "bin", "test_crc32_test")], capture_output=True, text=True)
	assert out.returncode == 0, out.stdout


def test_fast_inflate_against_zlib_streams(tool):
	out = subprocess.run([os.path.join(ROOT, "bin", "test_inflate_test")], capture_output=True, text=True)
	assert out.returncode == 0, out.stdout


def test_fast_inflate_fuzz_under_sanitizers(tmp_path):
	"""fast_inflate.hpp parses bytes of the input file before zlib does: damaged, cut and random streams, from and into heap buffers
	of exactly the sizes given, with AddressSanitizer and UndefinedBehaviorSanitizer (CPU only: the GPU boxes refuse sanitizers)."""
	exe = str(tmp_path / "inflate_fuzz")
	cmd = [os.environ.get("CXX", "g++"), "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
		   os.path.join(ROOT, "tests", "cpp", "inflate_test.cpp"), "-o", exe, "-lz"]
	subprocess.check_call(cmd)
	out = subprocess.run([exe, "fuzz", "6000"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
	assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
	assert "0 failures" in out.stdout


def test_writer_with_a_group_deflater(tmp_path):
	"""byte_writer's group path (the GPU compressor's slot interface) with a host stand-in: the same bytes, cut into the same
	blocks, whether the groups are attached from the start, after some output, or not at all; one slot or several."""
	import gzip
	import numpy as np
	from panvc3_amd import build
	build.build_cpp_tests()
	tool = os.path.join(ROOT, "bin", "test_group_writer_test")
	rng = np.random.default_rng(17)
	for size in (0, 1, 65279, 65280, 65281, 3 * 65280, 3 * 65280 + 1, 6 * 65280, 1_500_000):
		data = bytes(rng.integers(0, 7, size=size, dtype=np.uint8))
		cuts = {}
		for mode, slots in (("plain", 1), ("group", 1), ("group", 2), ("group", 3), ("late", 2)):
			out = tmp_path / ("%s%d.bgzf" % (mode, slots))
			subprocess.run([tool, str(out), mode, str(slots)], input=data, check=True, timeout=120)
			raw = out.read_bytes()
			assert gzip.decompress(raw) == data
			assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
			sizes, pos = [], 0
			while pos < len(raw):
				(bsize,) = struct.unpack_from("<H", raw, pos + 16)
				sizes.append(struct.unpack_from("<I", raw, pos + bsize + 1 - 4)[0])
				pos += bsize + 1
			cuts[(mode, slots)] = sizes
		assert len(set(map(tuple, cuts.values()))) == 1


def test_reader_with_a_group_inflater(tmp_path):
	"""byte_reader's group path (the GPU inflater's slot interface) with a host stand-in: the same bytes whether the pieces are
	inflated by the group from the start, from some point on, or not at all; blocks the group refuses are inflated by the
	reader, blocks it returns wrong bytes for are found by their checksum and inflated again; small pieces make records and
	blocks straddle them; a damaged block is an error whoever inflates."""
	import re
	import sys
	import numpy as np
	from panvc3_amd import build
	build.build_cpp_tests()
	tool = os.path.join(ROOT, "bin", "test_group_reader_test")
	sys.path.insert(0, os.path.join(ROOT, "tests"))
	import bamio
	rng = np.random.default_rng(19)
	data = bytes(np.repeat(rng.integers(0, 20, size=400000, dtype=np.uint8), rng.integers(1, 6, size=400000)))[:1_000_003]
	blocks, pos = [], 0
	while pos < len(data):
		n = int(rng.integers(1, 65281))
		blocks.append(bamio._bgzf_block(data[pos:pos + n]))
		pos += n
		if 0 == len(blocks) % 7:
			blocks.append(bamio._bgzf_block(b""))		# empty blocks in the middle of the file are legal
	path = tmp_path / "in.bgzf"
	path.write_bytes(b"".join(blocks) + bamio._bgzf_block(b""))
	for chunk in (None, "70000", "200000"):
		env = dict(os.environ, PVC3_BGZF_CHUNK=chunk) if chunk else None
		for mode in ("plain", "group", "late", "refuse", "garble"):
			p = subprocess.run([tool, str(path), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
			assert 0 == p.returncode, p.stderr
			assert p.stdout == data, (chunk, mode)
			through, here = (int(x) for x in re.findall(r"(\d+)", p.stderr.decode())[:2])
			if "plain" == mode:
				assert 0 == through
			elif chunk is None:
				pass		# (pieces of 32 MB do not fit the stand-in's input: the reader keeps inflating itself)
			elif "late" != mode:		# (attached late, the pieces read ahead by then may have been the whole file)
				assert through > 0 and (here > 0) == (mode in ("refuse", "garble"))
	# one flipped bit in the middle of a block's data: an error in every mode
	raw = bytearray(path.read_bytes())
	raw[len(blocks[0]) + len(blocks[1]) // 2] ^= 4
	bad = tmp_path / "bad.bgzf"
	bad.write_bytes(bytes(raw))
	for mode in ("plain", "group", "refuse"):
		p = subprocess.run([tool, str(bad), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PVC3_BGZF_CHUNK="200000"), timeout=120)
		assert 0 != p.returncode and (b"corrupt" in p.stderr or b"checksum" in p.stderr)
