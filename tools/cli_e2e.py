#!/usr/bin/env python3
"""End-to-end timing of the command line tools (SURVEY.md §8 d iii): synthetic MSA -> panvc3_index_msa, synthetic BAM ->
panvc3_project_alignments -> BAM, wall clock.  usage: tools/cli_e2e.py [--scale 0.2] [--workdir DIR]
The reads are config C2 of BASELINE.json scaled (REF + 2 founders, 150 bp single-end)."""
import argparse
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from tests.synth import synth		# noqa: E402
import bamio						# noqa: E402


def prepare(workdir, scale=0.2, repeat=1):
	"""Synthetic inputs of the command line tools in `workdir`: the MSA rows, input.tsv, genome.fa, msa-index.dat (built with
	panvc3_index_msa) and in.bam (config C2 scaled, the records written `repeat` times).  Returns a dict with the tool paths,
	the number of records and what cli_e2e's report needs."""
	os.makedirs(workdir, exist_ok=True)
	os.chdir(workdir)
	from panvc3_amd import build
	build.build()
	tools = build.build_host()
	index_msa = [t for t in tools if t.endswith("panvc3_index_msa")][0]
	project = [t for t in tools if t.endswith("panvc3_project_alignments")][0]

	gp, rp, desc = synth.config("C2", scale=scale)
	genome = synth.Genome(gp)
	reads = synth.Reads(genome, rp)
	b = reads.batch()
	n = int(reads.n)
	names = ["chr1/REF"] + ["chr1/%d" % k for k in range(1, genome.n_entries)]
	lens = []
	with open("input.tsv", "w") as tsv:
		for e in range(genome.n_entries):
			words, n_cols, text = genome.entry(e)
			bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_cols]
			row = np.full(n_cols, ord("-"), dtype=np.uint8)
			row[bits == 0] = text
			row.tofile("row%d.txt" % e)
			lens.append(len(text))
			tsv.write("chr1\t%s\t%s\n" % (names[e].split("/")[1], os.path.abspath("row%d.txt" % e)))
			if 0 == e:
				with open("genome.fa", "wb") as f:
					f.write(b">chr1\n")
					t = text.tobytes()
					for i in range(0, len(t), 1 << 20):
						f.write(t[i:i + (1 << 20)])		# one long line per MiB is fine for the reader
						f.write(b"\n")
	t0 = time.perf_counter()
	subprocess.check_call([index_msa, "--build-index", "--sequence-inputs=input.tsv", "--msa-index-output=msa-index.dat"], stdout=subprocess.DEVNULL)
	t_index = time.perf_counter() - t0

	# ---- BAM straight from the SoA batch (CIGAR and SEQ are already in BAM encoding) --------------------------------------
	header = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, ln) for nm, ln in zip(names, lens)) + "@PG\tID:bowtie2\tPN:bowtie2\tVN:2.5.1\n"
	raw = [b"BAM\1", struct.pack("<i", len(header)), header.encode(), struct.pack("<i", len(names))]
	for nm, ln in zip(names, lens):
		raw += [struct.pack("<i", len(nm) + 1), nm.encode(), b"\0", struct.pack("<i", ln)]
	cig_off, seq_off, qual_off = b["cigar_off"].astype(np.int64), b["seq_off"].astype(np.int64), b["qual_off"].astype(np.int64)
	cig_bytes, seq_bytes, qual_bytes = b["cigar"].tobytes(), b["seq4"].tobytes(), b["qual"].tobytes()
	src_pos, src_entry, l_seq = b["src_pos"], b["src_entry"], b["l_seq"]
	tags = b"ASc\xfbXSc\xecYTZUU\0"
	for i in range(n):
		qname = b"r%d\0" % i
		n_cig = int(cig_off[i + 1] - cig_off[i])
		body = struct.pack("<iiBBHHHIiii", int(src_entry[i]), int(src_pos[i]), len(qname), 30, 4680, n_cig, 0, int(l_seq[i]), -1, -1, 0)
		rec = body + qname + cig_bytes[4 * cig_off[i]:4 * cig_off[i + 1]] + seq_bytes[seq_off[i]:seq_off[i + 1]] + qual_bytes[qual_off[i]:qual_off[i + 1]] + tags
		raw.append(struct.pack("<i", len(rec)))
		raw.append(rec)
	n_head = 4 + 4 * len(names)
	# BGZF blocks are independent: the header is compressed on its own and the blocks of the records are written `repeat` times
	# (a larger input for the price of compressing the records once).
	head_bytes, body_bytes = b"".join(raw[:n_head]), b"".join(raw[n_head:])
	head_blocks = b"".join(bamio._bgzf_block(head_bytes[i:i + 0xff00]) for i in range(0, len(head_bytes), 0xff00))
	body_blocks = b"".join(bamio._bgzf_block(body_bytes[i:i + 0xff00]) for i in range(0, len(body_bytes), 0xff00))
	n *= repeat
	with open("in.bam", "wb") as f:
		f.write(head_blocks)
		for _ in range(repeat):
			f.write(body_blocks)
		f.write(bamio._bgzf_block(b""))
	with open("empty.bam", "wb") as f:		# the same header, no records: what the fixed part of a run costs
		f.write(head_blocks)
		f.write(bamio._bgzf_block(b""))
	return dict(project=project, index_msa=index_msa, n=n, lens=lens, n_entries=genome.n_entries, t_index=t_index, in_size=os.path.getsize("in.bam"), workdir=workdir,
				head_bytes=len(head_blocks), body_bytes=len(body_blocks), n_unique=n // repeat)


def write_repeated(P, path, repeat):
	"""A BAM file with in.bam's header and its records `repeat` times (in.bam's own blocks, copied); returns the number of records."""
	with open(os.path.join(P["workdir"], "in.bam"), "rb") as f:
		head = f.read(P["head_bytes"])
		body = f.read(P["body_bytes"])
	with open(path, "wb") as f:
		f.write(head)
		for _ in range(repeat):
			f.write(body)
		f.write(bamio._bgzf_block(b""))
	return P["n_unique"] * repeat


def project_command(P, alignments="in.bam", output="out.bam", bam=True):
	cmd = [P["project"], "-a", alignments, "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-o", output, "--debugging-output"]
	return cmd + (["-b"] if bam else [])


def fresh_output(cmd, cwd=None):
	if "-o" in cmd:
		path = os.path.join(cwd or ".", cmd[cmd.index("-o") + 1])
		if os.path.exists(path): os.remove(path)


def timed(cmd, cwd=None):
	"""(seconds, the tool's --debugging-output stage line or None).  The output is a new file every time: a file that exists is
	truncated, and closing a truncated and rewritten file makes the file system write it out there and then (ext4's
	auto_da_alloc) — 0.19 s of the 1.24 of a run with 2 GB of output, none of it the tool's."""
	fresh_output(cmd, cwd)
	t = time.perf_counter()
	p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=cwd)
	dt = time.perf_counter() - t
	if p.returncode:
		raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), p.stderr.decode()[-2000:]))
	stages = [line.strip() for line in p.stderr.decode().split("\n") if "Stages" in line]
	return dt, (stages[0] if stages else None)


def records_digest(path):
	"""(number of records, SHA-256 of the uncompressed records) of a BAM file: everything behind the header, which holds the
	command line.  Inflated block by block with zlib, CRC32 and ISIZE checked."""
	import hashlib
	h = hashlib.sha256()
	raw = open(path, "rb").read()
	pos, out, header_left, n_records, carry = 0, [], None, 0, b""
	while pos < len(raw):
		(bsize,) = struct.unpack_from("<H", raw, pos + 16)
		d = zlib.decompressobj(-15)
		piece = d.decompress(raw[pos + 18:pos + bsize + 1 - 8]) + d.flush()
		crc, isize = struct.unpack_from("<II", raw, pos + bsize + 1 - 8)
		if not d.eof or len(piece) != isize or (zlib.crc32(piece) & 0xffffffff) != crc:
			raise RuntimeError("%s: bad BGZF block at %d" % (path, pos))
		pos += bsize + 1
		carry += piece
		if header_left is None:		# magic, l_text, text, n_ref, (l_name, name, l_ref) *
			if len(carry) < 12:
				continue
			(l_text,) = struct.unpack_from("<i", carry, 4)
			if len(carry) < 12 + l_text:
				continue
			(n_ref,) = struct.unpack_from("<i", carry, 8 + l_text)
			q, ok = 12 + l_text, True
			for _ in range(n_ref):
				if len(carry) < q + 4:
					ok = False; break
				(l_name,) = struct.unpack_from("<i", carry, q)
				q += 4 + l_name + 4
			if not ok or len(carry) < q:
				continue
			carry = carry[q:]
			header_left = 0
		# whole records of what is at hand
		q = 0
		while len(carry) - q >= 4:
			(bs,) = struct.unpack_from("<i", carry, q)
			if len(carry) - q < 4 + bs:
				break
			q += 4 + bs
			n_records += 1
		h.update(carry[:q])
		carry = carry[q:]
	if carry:
		raise RuntimeError("%s: %d bytes behind the last record" % (path, len(carry)))
	return n_records, h.hexdigest()


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--scale", type=float, default=0.2)
	ap.add_argument("--workdir", default="/tmp/pvc3_e2e")
	ap.add_argument("--repeat", type=int, default=1, help="write the records this many times (a larger input without generating more reads)")
	args = ap.parse_args()
	P = prepare(args.workdir, args.scale, args.repeat)
	project, n, lens, t_index, in_size = P["project"], P["n"], P["lens"], P["t_index"], P["in_size"]

	def run(cmd):
		t = time.perf_counter()
		p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
		if p.returncode:
			sys.stderr.write(p.stderr.decode())
			raise SystemExit("command failed: " + " ".join(cmd))
		dt = time.perf_counter() - t
		for line in p.stderr.decode().split("\n"):
			if "Stages" in line or "Timeline" in line or "BGZF on the GPU" in line:
				print("   ", " ".join(cmd[-3:]), "|", line.strip())
		return dt

	t_io = run([project, "--io-only", "-a", "in.bam", "-b", "-o", "io.bam"])
	full = [project, "-a", "in.bam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-b", "-o", "out.bam", "--debugging-output"]
	run(full)		# first run pays the page cache and the GPU start-up
	t_full = run(full)
	gpu_size = os.path.getsize("out.bam")
	digest_gpu = records_digest("out.bam")
	t_host = run([c if c != "out.bam" else "out_host.bam" for c in full] + ["--bgzf-deflate=host"])
	digest_host = records_digest("out_host.bam")
	print("records of out.bam (blocks compressed on the GPU, %.1f MB) and of out_host.bam (host threads, %.1f MB): %d and %d, %s"
		  % (gpu_size / 1e6, os.path.getsize("out_host.bam") / 1e6, digest_gpu[0], digest_host[0], "identical" if digest_gpu == digest_host else "DIFFERENT"))
	if digest_gpu != digest_host or digest_gpu[0] != n:
		raise SystemExit("the two outputs differ")
	os.remove("out_host.bam")
	t_small = run(full + ["--batch-size=131072"])
	t_sam = run([project, "-a", "in.bam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-o", "out.sam"])
	t_thr = {}
	for thr in (16, 64, 128):
		t_thr[thr] = run(full + ["--batch-size=262144", "--threads=%d" % thr])
	t_l1 = run(full + ["--batch-size=262144", "--threads=64", "--compression-level=1"])
	t_l6 = run(full + ["--batch-size=262144", "--threads=64", "--compression-level=6"])
	l6_size = os.path.getsize("out.bam")
	# the fixed part (index file, FASTA, GPU start-up, index upload): the same command on an input without records (empty.bam)
	run(full)
	out_size = os.path.getsize("out.bam")
	t_empty = run([project, "-a", "empty.bam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-b", "-o", "empty_out.bam"])
	print("records %d, genome %d bp x %d rows, in.bam %.1f MB, out.bam (built-in deflate) %.1f MB" % (n, lens[0], P["n_entries"], in_size / 1e6, out_size / 1e6))
	print("index_msa --build-index: %.2f s" % t_index)
	print("project_alignments --io-only BAM->BAM: %.2f s = %.2f M records/s" % (t_io, n / t_io / 1e6))
	print("project_alignments BAM->BAM: %.2f s = %.2f M records/s" % (t_full, n / t_full / 1e6))
	print("project_alignments BAM->BAM, BGZF blocks compressed by the host threads (--bgzf-deflate=host): %.2f s = %.2f M records/s" % (t_host, n / t_host / 1e6))
	print("project_alignments BAM->BAM, batches of 131072 (reading overlaps the GPU and the writing): %.2f s = %.2f M records/s" % (t_small, n / t_small / 1e6))
	print("project_alignments BAM->SAM: %.2f s = %.2f M records/s" % (t_sam, n / t_sam / 1e6))
	for thr, t in t_thr.items():
		print("project_alignments BAM->BAM, batches of 262144, %d threads: %.2f s = %.2f M records/s" % (thr, t, n / t / 1e6))
	print("project_alignments BAM->BAM, batches of 262144, 64 threads, zlib level 1: %.2f s = %.2f M records/s" % (t_l1, n / t_l1 / 1e6))
	print("project_alignments BAM->BAM, batches of 262144, 64 threads, zlib level 6: %.2f s = %.2f M records/s (out.bam %.1f MB)" % (t_l6, n / t_l6 / 1e6, l6_size / 1e6))
	print("the same command on an input without records (index file, FASTA, GPU start-up): %.2f s" % t_empty)
	best = min(t_thr.values())
	print("best run without that fixed part: %.2f M records/s" % (n / max(1e-9, best - t_empty) / 1e6))


if __name__ == "__main__":
	main()
