"""The C++ host pipeline: panvc3_index_msa, the SAM/BAM I/O layer and the panvc3_project_alignments
command line.  The index tool and the I/O layer are tested on the CPU; the projecting command line needs a GPU.

Expected outputs come from oracle/project_records.py, a restatement of the reference's record / header rewriting
(parity against the real binary is unpinned: the reference has no test and no checked-in output for this layer)."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, load_test_msa

import bamio

BIN = os.path.join(ROOT, "bin")


@pytest.fixture(scope="module")
def tools():
	from panvc3_amd import build
	build.build_host()
	return {n: os.path.join(BIN, n) for n in ("panvc3_index_msa", "panvc3_project_alignments", "panvc3_convert_bed_positions", "panvc3_rewrite_cigar")}


def _run(cmd, cwd=None, check=True, env=None):
	p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=None if env is None else dict(os.environ, **env), timeout=600)		# (a tool that hangs must fail the test, not the run)
	if check and p.returncode != 0:
		raise AssertionError("%s failed (%d): %s" % (cmd[0], p.returncode, p.stderr.decode()[-2000:]))
	return p


def _read_index_file(path):
	raw = open(path, "rb").read()
	assert raw[:8] == b"PVC3IDX\x01"
	p = 8
	n_chr, = struct.unpack_from("<I", raw, p); p += 4
	out = {}
	for _ in range(n_chr):
		l, = struct.unpack_from("<I", raw, p); p += 4
		chr_id = raw[p:p + l].decode(); p += l
		n_seq, = struct.unpack_from("<I", raw, p); p += 4
		for _ in range(n_seq):
			l, = struct.unpack_from("<I", raw, p); p += 4
			seq_id = raw[p:p + l].decode(); p += l
			n_cols, = struct.unpack_from("<Q", raw, p); p += 8
			nw = (n_cols + 63) // 64
			words = np.frombuffer(raw, dtype="<u8", count=nw, offset=p).copy(); p += 8 * nw
			out[(chr_id, seq_id)] = (n_cols, words)
	assert p == len(raw)
	return out


def _setup_msa_dir(tmp_path):
	d = tmp_path / "index-input" / "sequences"
	d.mkdir(parents=True)
	for f in os.listdir(os.path.join(GOLDEN, "test_msa")):
		if f.endswith(".bz2"):
			shutil.copy(os.path.join(GOLDEN, "test_msa", f), d / f)
	shutil.copy(os.path.join(GOLDEN, "test_msa", "input.tsv"), tmp_path / "index-input" / "input.tsv")


@pytest.fixture()
def built_index(tools, tmp_path):
	"""The test-workflow's build_msa_index rule (test-workflow/Snakefile:36) on the shipped MSA."""
	_setup_msa_dir(tmp_path)
	pipe = "bzip2 -d -c" if shutil.which("bzip2") else "%s -c \"import bz2,sys;sys.stdout.buffer.write(bz2.open(sys.argv[1]).read())\"" % sys.executable
	p = _run([tools["panvc3_index_msa"], "--build-index", "--sequence-inputs=index-input/input.tsv", "--msa-index-output=msa-index.dat", "--output-fasta", "--pipe-input=" + pipe], cwd=tmp_path)
	return tmp_path, p.stdout.decode()


def test_index_msa_on_shipped_msa(tools, built_index):
	tmp_path, fasta = built_index
	rows = load_test_msa()
	# FASTA: ">chr/seq" + the non-gap bytes on one line (index-msa/index_vector_builder.cc:15-58), in input order
	expect = "".join(">test/%s\n%s\n" % (k, rows[k].replace(b"-", b"").decode()) for k in ("REF", "1", "2", "3", "4"))
	assert fasta == expect
	idx = _read_index_file(tmp_path / "msa-index.dat")
	assert sorted(idx) == [("test", k) for k in sorted(rows)]		# sorted by id like the reference
	for k, row in rows.items():
		n_cols, words = idx[("test", k)]
		assert n_cols == len(row) == 12636
		gaps = np.frombuffer(row, dtype=np.uint8) == ord("-")
		bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_cols].astype(bool)
		assert np.array_equal(bits, gaps)
	out = _run([tools["panvc3_index_msa"], "--list-contents", "--msa-index-input=msa-index.dat"], cwd=tmp_path).stdout.decode()
	assert out == "test (5 sequences of 12636 characters)\n" + "".join("\t%s\n" % k for k in sorted(rows))
	# --fasta-line-width
	p = _run([tools["panvc3_index_msa"], "-B", "-i", "index-input/input.tsv", "-m", "x.dat", "-f", "-w", "60", "-p", "bzip2 -d -c" if shutil.which("bzip2") else "bzcat"], cwd=tmp_path, check=False)
	if p.returncode == 0:
		lines = p.stdout.decode().split("\n")
		assert lines[0] == ">test/REF" and all(len(x) == 60 for x in lines[1:167]) and len(lines[167]) == 10000 - 166 * 60


def test_index_msa_query_mode(tools, built_index, oracle_mod):
	"""index-msa/main.cc:88-186: interactive project_position queries, 's' / 'd' switch the sequences."""
	tmp_path, _ = built_index
	rows = load_test_msa()
	o = oracle_mod.Oracle()
	ent = {k: o.add_aligned(rows[k]) for k in ("REF", "1", "2", "3", "4")}
	n1, n3 = o.zeros(ent["1"]), o.zeros(ent["3"])
	script = "nosuch\n1\nREF\n0\n17\nabc\n%d\n%d\nd\n2\n100\ns\n3\n%d\n" % (n1 - 1, n1, n3 - 1)
	p = subprocess.run([tools["panvc3_index_msa"], "--query", "--msa-index-input=msa-index.dat", "--chr-id=test"], cwd=tmp_path, input=script.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
	assert p.returncode == 0 and b"No entry for sequence 'nosuch'." in p.stderr
	out = p.stdout.decode()
	numbers = [int(x) for x in __import__("re").findall(r"\)\) (\d+)\n", out)]
	pp = o.project_position
	assert numbers == [pp(ent["1"], ent["REF"], 0), pp(ent["1"], ent["REF"], 17), pp(ent["1"], ent["REF"], n1 - 1), pp(ent["1"], ent["2"], 100), pp(ent["3"], ent["2"], n3 - 1)]
	assert "Source sequence identifier? " in out and "Destination sequence identifier? " in out
	assert "[1 \u2192 REF] Source co-ordinate or 's' or 'd' to switch sequence? ([0, %d)) " % n1 in out and "[3 \u2192 2] " in out
	bad = _run([tools["panvc3_index_msa"], "--query", "--msa-index-input=msa-index.dat", "--chr-id=other"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and b"ERROR: No entry for chromosome 'other'." in bad.stderr


def test_index_msa_a2m_and_errors(tools, tmp_path):
	(tmp_path / "in.a2m").write_text(">chrB s2\nAC-GT\nAC\n>chrA\tREF\nACGT--A\n>chrB REF\nA--GTAC\n")
	_run([tools["panvc3_index_msa"], "--build-index", "--sequences=in.a2m", "--msa-index-output=o.dat"], cwd=tmp_path)
	idx = _read_index_file(tmp_path / "o.dat")
	assert list(idx) == [("chrA", "REF"), ("chrB", "REF"), ("chrB", "s2")]
	assert idx[("chrB", "s2")][0] == 7 and int(idx[("chrB", "s2")][1][0]) == 0b0000100
	(tmp_path / "bad.a2m").write_text(">c a\nACGT\n>c b\nAC\n")
	p = _run([tools["panvc3_index_msa"], "--build-index", "--sequences=bad.a2m", "--msa-index-output=o2.dat"], cwd=tmp_path, check=False)
	assert p.returncode != 0 and b"expected 4" in p.stderr
	p = _run([tools["panvc3_project_alignments"], "--alignments=nonexistent.sam", "--msa-index=o.dat", "--reference=x.fa", "--reference-msa-id=REF"], cwd=tmp_path, check=False)
	assert p.returncode != 0
	p = _run([tools["panvc3_project_alignments"], "--original-pos-tag=AB", "--alignments=x", "--msa-index=o.dat", "--reference=x.fa", "--reference-msa-id=REF"], cwd=tmp_path, check=False)
	assert p.returncode != 0 and b"does not match the expected format" in p.stderr


def _make_records(synth_mod, rows, n=600, seed=5):
	"""Reads sampled from the shipped MSA as bowtie2-like SAM records (pairs, tags, a few records that must be dropped)."""
	order = ["REF", "1", "2", "3", "4"]
	genome = synth_mod.Genome(aligned_rows=[rows[k] for k in order])
	gp, rp, _ = synth_mod.config("C3", scale=1e-6)
	rp.n_records = n; rp.read_len = 100; rp.seed = seed; rp.indel_read_frac = 0.2; rp.clip_read_frac = 0.2
	reads = synth_mod.Reads(genome, rp)
	b = reads.batch(copy=True)
	rng = np.random.default_rng(seed)
	names = ["test/REF", "test/1", "test/2", "test/3", "test/4"]		# entry e of the genome <-> names[e]
	recs = []
	for i in range(reads.n):
		cig = "".join("%d%s" % (int(c) >> 4, "MIDNSHP=X"[int(c) & 15]) for c in b["cigar"][int(b["cigar_off"][i]):int(b["cigar_off"][i + 1])])
		s4 = b["seq4"][int(b["seq_off"][i]):int(b["seq_off"][i + 1])]
		L = int(b["l_seq"][i])
		seq = "".join("=ACMGRSVTWYHKDBN"[(int(s4[k >> 1]) >> (4 if k % 2 == 0 else 0)) & 15] for k in range(L))
		qual = "".join(chr(33 + int(q)) for q in b["qual"][int(b["qual_off"][i]):int(b["qual_off"][i + 1])])
		first = i % 2 == 0
		flag = (99 if first else 147)
		tags = [("AS", "i", -int(rng.integers(0, 30))), ("XS", "i", -int(rng.integers(0, 60))), ("NM", "i", int(rng.integers(0, 5))), ("MD", "Z", "100"), ("YS", "i", -3), ("YT", "Z", "CP")]
		if i % 7 == 0:
			tags.append(("OA", "Z", "old/1,5,+,3M,2,0;"))
		if i % 11 == 0:
			tags.append(("ZB", "B", ("S", [1, 2, 3])))
		if i % 13 == 0:
			tags = [t for t in tags if t[0] != "NM"]
		rec = bamio.Record("read%d" % (i // 2), flag, names[int(b["src_entry"][i])], int(b["src_pos"][i]) + 1, int(rng.integers(0, 43)), cig, "=", int(b["mate_pos"][i]) + 1,
						   300 if first else -300, seq, qual, tags)
		if i % 17 == 0:
			rec.seq = rec.seq[:10] + "R" + rec.seq[11:]			# IUPAC code: N after dna5
		if i % 19 == 0:
			rec.qual = "".join(chr(min(126, ord(c) + 20)) for c in rec.qual)	# beyond phred 41: saturates
		recs.append(rec)
		if i % 23 == 0:
			recs.append(bamio.Record("unmapped%d" % i, 77, "*", 0, 0, "*", "*", 0, 0, "ACGT", "IIII", [("YT", "Z", "UP")]))
		if i % 29 == 0:
			recs.append(bamio.Record("dup%d" % i, flag | 0x400, rec.rname, rec.pos, rec.mapq, rec.cigar, "=", rec.pnext, rec.tlen, rec.seq, rec.qual, []))
		if i % 31 == 0:
			recs.append(bamio.Record("nomate%d" % i, 0, rec.rname, rec.pos, 7, rec.cigar, "*", 0, 0, rec.seq, "*", [("XS", "i", 5)]))
		if i % 37 == 0:
			recs.append(bamio.Record("othermate%d" % i, 65, rec.rname, rec.pos, 7, rec.cigar, "test/3", 17, 0, rec.seq, rec.qual, []))
	return genome, recs, names


def test_io_round_trip_sam_bam(tools, tmp_path):
	from tests.synth import synth
	rows = load_test_msa()
	genome, recs, names = _make_records(synth, rows, n=200)
	header = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, 10000 + i) for i, n in enumerate(names)) + "@RG\tID:g1\tSM:s\n@PG\tID:bowtie2\tPN:bowtie2\tVN:2.5.1\tCL:\"bowtie2 -x i\"\n@CO\thello\n"
	bamio.write_sam(tmp_path / "in.sam", header, recs)
	_run([tools["panvc3_project_alignments"], "--io-only", "--alignments=in.sam", "--output-bam", "--output-path=x.bam"], cwd=tmp_path)
	text, refs, got = bamio.read_bam(tmp_path / "x.bam")
	assert refs == [(n, 10000 + i) for i, n in enumerate(names)]
	assert len(got) == len(recs)
	ref_index = {n: i for i, n in enumerate(names)}
	for g, r in zip(got, recs):
		assert g["qname"] == r.qname and g["flag"] == r.flag and g["pos"] == r.pos - 1 and g["cigar"] == r.cigar and g["seq"] == r.seq.upper() and g["qual"] == r.qual
		assert g["ref_id"] == (-1 if r.rname == "*" else ref_index[r.rname])
		assert [(t[0], t[1], t[2]) for t in g["tags"]] == [tuple(t) for t in r.tags]
	# BAM -> SAM reproduces the original text (header in seqan3's field order)
	_run([tools["panvc3_project_alignments"], "--io-only", "--alignments=x.bam", "--output-path=back.sam"], cwd=tmp_path)
	back = open(tmp_path / "back.sam").read().split("\n")
	body = [l for l in back if l and not l.startswith("@")]
	assert body == [r.sam_line() for r in recs]
	assert [l for l in back if l.startswith("@SQ")] == ["@SQ\tSN:%s\tLN:%d" % (n, 10000 + i) for i, n in enumerate(names)]
	# the BGZF layer cuts blocks at fixed offsets: the file does not depend on the number of deflate threads
	_run([tools["panvc3_project_alignments"], "--io-only", "--threads=1", "-a", "in.sam", "-b", "-o", "t1.bam"], cwd=tmp_path)
	_run([tools["panvc3_project_alignments"], "--io-only", "--threads=7", "-a", "t1.bam", "-b", "-o", "t7.bam"], cwd=tmp_path)
	# the BGZF reader in pieces far smaller than its 32 MB (even than a block): the records and the header straddle them
	for chunk in ("700", "5000"):
		_run([tools["panvc3_project_alignments"], "--io-only", "-a", "t1.bam", "-o", "chunk%s.sam" % chunk], cwd=tmp_path, env={"PVC3_BGZF_CHUNK": chunk})
		assert open(tmp_path / ("chunk%s.sam" % chunk)).read() == open(tmp_path / "back.sam").read()
	assert open(tmp_path / "t1.bam", "rb").read() == open(tmp_path / "t7.bam", "rb").read() == open(tmp_path / "x.bam", "rb").read()
	# a truncated BGZF file is an error, not a short read
	raw = open(tmp_path / "x.bam", "rb").read()
	open(tmp_path / "cut.bam", "wb").write(raw[:len(raw) // 2])
	p = _run([tools["panvc3_project_alignments"], "--io-only", "-a", "cut.bam", "-o", "cut.sam"], cwd=tmp_path, check=False)
	assert p.returncode != 0 and b"ERROR" in p.stderr
	# a record whose l_seq (or n_cigar_op, l_read_name) points past its block_size is an error, not an out-of-bounds read
	import struct
	ref_index2 = {n: i for i, n in enumerate(names)}
	refs2 = [(n, 10000 + i) for i, n in enumerate(names)]
	for field_off, fmt, value in ((16, "<I", 100000), (12, "<H", 60000), (8, "<B", 255)):
		rawbam = b"BAM\1" + struct.pack("<i", len(header)) + header.encode() + struct.pack("<i", len(refs2))
		for n, l in refs2:
			rawbam += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
		good = bamio.bam_bytes(recs[0], ref_index2)
		bad = bytearray(bamio.bam_bytes(recs[1], ref_index2))
		struct.pack_into(fmt, bad, 4 + field_off, value)
		rawbam += good + bytes(bad)
		with open(tmp_path / "bad.bam", "wb") as f:
			f.write(bamio._bgzf_block(rawbam))
			f.write(bamio._bgzf_block(b""))
		p = _run([tools["panvc3_project_alignments"], "--io-only", "-a", "bad.bam", "-o", "bad.sam"], cwd=tmp_path, check=False)
		assert p.returncode != 0 and b"malformed BAM record" in p.stderr, (field_off, p.stderr)
	# python-written BAM is read too (bgzf blocks written by another implementation)
	bamio.write_bam(tmp_path / "py.bam", header, [(n, 10000 + i) for i, n in enumerate(names)], recs)
	_run([tools["panvc3_project_alignments"], "--io-only", "-a", "py.bam", "-o", "py.sam"], cwd=tmp_path)
	assert [l for l in open(tmp_path / "py.sam").read().split("\n") if l and not l.startswith("@")] == [r.sam_line() for r in recs]


def _rewrite_cigar_case(built_index):
	"""rc_in.bam and ref.fa in the index fixture's directory; the records, the lines -S has to print and their CIGARs (oracle/rewrite_cigar_tool.py)."""
	from oracle import rewrite_cigar_tool as rc
	from tests.synth import synth
	tmp_path, fasta = built_index
	rows = load_test_msa()
	genome, recs, names = _make_records(synth, rows, n=400, seed=8)
	seqs = {n: rows[k].replace(b"-", b"").decode() for n, k in zip(names, ["REF", "1", "2", "3", "4"])}
	# a reference with N and lower-case stretches: only an upper-case match counts, only 'N' is marked
	for n in seqs:
		t = list(seqs[n])
		for i in range(50, len(t), 97): t[i] = "N"
		for i in range(300, len(t), 211): t[i] = t[i].lower()
		seqs[n] = "".join(t)
	with open(tmp_path / "ref.fa", "w") as f:
		for n in names:
			f.write(">%s extra words\n" % n)
			for i in range(0, len(seqs[n]), 60): f.write(seqs[n][i:i + 60] + "\n")
	header = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(seqs[n])) for n in names) + "@PG\tID:bowtie2\tPN:bowtie2\tVN:2.5.1\n"
	bamio.write_bam(tmp_path / "rc_in.bam", header, [(n, len(seqs[n])) for n in names], recs)

	def expected_line(r, cigar, extra_tag=None):
		tags = [(t, ty, v) for t, ty, v in r.tags]
		if extra_tag: tags.append(extra_tag)
		tags.sort(key=lambda t: (ord(t[0][0]), ord(t[0][1])))
		seq = "".join(rc.dna5_char(c) for c in r.seq)
		qual = r.qual if r.qual == "*" else "".join(min(c, "J") for c in r.qual)
		rnext = "=" if r.rnext == r.rname and r.rname != "*" else r.rnext		# the SAM writer abbreviates an equal RNEXT
		return bamio.Record(r.qname, r.flag, r.rname, r.pos, r.mapq, cigar, rnext, r.pnext, r.tlen, seq, qual, tags).sam_line()

	want_s, cig_s = [], []
	for r in recs:
		if r.rname == "*" or r.pos <= 0:
			want_s.append(expected_line(r, r.cigar)); cig_s.append(r.cigar)
			continue
		cig, nb = rc.sequence_match(r.cigar, r.seq, seqs[r.rname], r.pos - 1)
		cig_s.append(cig)
		want_s.append(expected_line(r, cig, ("XN", "B", ("C", list(nb))) if nb else None))
	return tmp_path, recs, want_s, cig_s


def test_rewrite_cigar_tool(tools, built_index):
	"""bin/panvc3_rewrite_cigar against oracle/rewrite_cigar_tool.py: M -> = / X with the N bit map, and back to M."""
	from oracle import rewrite_cigar_tool as rc
	tmp_path, recs, want_s, cig_s = _rewrite_cigar_case(built_index)
	p = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "--output-sequence-match-ops", "--reference=ref.fa"], cwd=tmp_path)
	out = p.stdout.decode().split("\n")
	assert [l for l in out if l and not l.startswith("@")] == want_s
	assert any("\tXN:B:C," in l for l in want_s) and any("X" in c for c in cig_s)
	assert [l for l in out if l.startswith("@PG")][-1].startswith("@PG\tID:panvc3.rewrite-cigar.1\tPN:PanVC 3 rewrite_cigar\tCL:")
	# back: = / X -> M, merged; BAM in and out
	_run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "-r", "ref.fa", "-b", "-o", "rc_s.bam"], cwd=tmp_path)
	q = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_s.bam", "--output-alignment-match-ops"], cwd=tmp_path)
	body = [l for l in q.stdout.decode().split("\n") if l and not l.startswith("@")]
	assert [l.split("\t")[5] for l in body] == [rc.alignment_match(c) if c != "*" else "*" for c in cig_s]
	assert [l.split("\t")[5] for l in body] == [rc.alignment_match(r.cigar) if r.cigar != "*" else "*" for r in recs]
	bad = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and b"--reference is required" in bad.stderr


@pytest.mark.gpu
def test_rewrite_cigar_tool_on_the_gpu(tools, built_index):
	"""--device=0: the M operations split and the N positions found by pvc3_split_match_ops; the same lines as the restatement's, and
	the same bytes as the host's run but for the command line in @PG; the records the tool refuses are refused with the same words."""
	tmp_path, recs, want_s, cig_s = _rewrite_cigar_case(built_index)
	host = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "--reference=ref.fa"], cwd=tmp_path).stdout.decode().split("\n")
	gpu = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "--reference=ref.fa", "--device=0"], cwd=tmp_path).stdout.decode().split("\n")
	assert [l for l in gpu if l and not l.startswith("@")] == want_s
	assert [l for l in gpu if not l.startswith("@PG\tID:panvc3.rewrite-cigar")] == [l for l in host if not l.startswith("@PG\tID:panvc3.rewrite-cigar")]
	_run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "-r", "ref.fa", "-b", "-o", "rc_gpu.bam", "--device=0"], cwd=tmp_path)
	_run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "-r", "ref.fa", "-b", "-o", "rc_host.bam"], cwd=tmp_path)
	assert bamio.read_bam(tmp_path / "rc_gpu.bam")[1:] == bamio.read_bam(tmp_path / "rc_host.bam")[1:]
	# an alignment past the end of its sequence: both refuse it
	short = tmp_path / "short.fa"
	text = open(tmp_path / "ref.fa").read().split(">")
	with open(short, "w") as f:
		for part in text[1:]:
			lines = part.split("\n")
			f.write(">" + lines[0] + "\n" + "".join(lines[1:])[:40] + "\n")
	for extra in ([], ["--device=0"]):
		bad = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-S", "-r", "short.fa"] + extra, cwd=tmp_path, check=False)
		assert bad.returncode != 0 and b"extends past the end of its reference sequence" in bad.stderr, extra
	bad = _run([tools["panvc3_rewrite_cigar"], "-a", "rc_in.bam", "-M", "--device=0"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and b"--device has an effect" in bad.stderr


@pytest.mark.gpu
def test_project_alignments_cli_matches_restatement(tools, built_index, oracle_mod):
	"""The workflow's command line (test-workflow/Snakefile:76-86) end to end: BAM in, SAM on stdout; and BAM out, ranges TSV."""
	from oracle import project_records as pr
	from tests.synth import synth
	tmp_path, fasta = built_index
	rows = load_test_msa()
	genome, recs, names = _make_records(synth, rows, n=1200)
	ref_text = rows["REF"].replace(b"-", b"").decode()
	with open(tmp_path / "genome.fa", "w") as f:
		f.write(">test some description\n")
		for i in range(0, len(ref_text), 70):
			f.write(ref_text[i:i + 70] + "\n")
	subprocess.check_call(["gzip", "-k", "genome.fa"], cwd=tmp_path)
	(tmp_path / "order.txt").write_text("nosuch\ntest\t10000\n")
	lens = {n: len(rows[k].replace(b"-", b"")) for n, k in zip(names, ["REF", "1", "2", "3", "4"])}
	in_refs = [(n, lens[n], "M5:abc" if n == "test/REF" else "") for n in ["test/2", "test/REF", "test/1", "test/4", "test/3"]]		# header order != entry order
	header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d%s\n" % (n, l, ("\t" + e) if e else "") for n, l, e in in_refs)
	header += "@RG\tID:g1\tSM:s\n@PG\tID:bowtie2\tPN:bowtie2\tVN:2.5.1\tCL:\"bowtie2 -x i\"\n@PG\tID:panvc3.project-alignments.3\tPN:x\tPP:bowtie2\n@CO\thello\n"
	bamio.write_bam(tmp_path / "alignments.sorted.bam", header, [(n, l) for n, l, e in in_refs], recs)
	bamio.write_sam(tmp_path / "alignments.sam", header, recs)

	o = oracle_mod.Oracle()
	seq_of = {"test/REF": "REF", "test/1": "1", "test/2": "2", "test/3": "3", "test/4": "4"}
	ent = {n: o.add_aligned(rows[seq_of[n]]) for n in names}
	o.set_reference(ent["test/REF"], ref_text)
	out_refs, mapping, chr_seq = pr.process_headers(in_refs, "/", "REF", ["nosuch", "test"])
	assert out_refs == [("test", 10000, "M5:abc")]
	src_entries = [ent[n] for n, _, _ in in_refs]
	dst_entries = [ent["test/REF"]]
	opts = dict(preserve={"XS", "YS"}, rname_tag="XR", pos_tag="XP", rnext_tag="YR", pnext_tag="YP", query_ranges_tag="XB", ref_ranges_tag="YB", rec_idx_tag="XI")
	expect, ranges = pr.project_records(recs, in_refs, o, src_entries, dst_entries, mapping, out_refs, opts)
	assert len(expect) > 1000 and any("\tXB:B:I," in l for l in expect)

	cmd = [tools["panvc3_project_alignments"], "--alignments=alignments.sorted.bam", "--msa-index=msa-index.dat", "--reference=genome.fa.gz", "--reference-msa-id=REF",
		   "--ref-id-separator=/", "--reference-order-input=order.txt", "--record-index-tag=XI", "--preserve-tag=XS", "--preserve-tag=YS",
		   "--output-realigned-ranges=ranges.tsv", "--batch-size=256"]
	p = _run(cmd, cwd=tmp_path)
	out = p.stdout.decode().split("\n")
	head = [l for l in out if l.startswith("@")]
	body = [l for l in out if l and not l.startswith("@")]
	assert body == expect
	assert head[0] == "@HD\tVN:1.6" and head[1] == "@SQ\tSN:test\tLN:10000\tM5:abc" and head[2] == "@RG\tID:g1\tSM:s"
	assert head[3:5] == ["@PG\tID:bowtie2\tPN:bowtie2\tCL:\"bowtie2 -x i\"\tVN:2.5.1", "@PG\tID:panvc3.project-alignments.3\tPN:x\tPP:bowtie2"]
	assert head[5].startswith("@PG\tID:panvc3.project-alignments.4\tPN:PanVC 3 project_alignments\tCL:") and "\tPP:panvc3.project-alignments.3\t" in head[5]
	assert head[6] == "@CO\thello"
	assert open(tmp_path / "ranges.tsv").read() == "Location\tLength\n" + "".join("%d\t%d\n" % r for r in ranges)
	assert b"Matched reads:" in p.stderr

	# Several GPUs (here: the one GPU twice, each with a context, an index replica and a pipeline of its own), batches of 64
	# records going to them in turn and coming back in input order; and one thread instead of the default.
	for extra in (["--devices=0,0", "--batch-size=64"], ["--devices=0,0,0", "--batch-size=100", "--threads=3"], ["--threads=1"]):
		pd = _run(cmd + extra, cwd=tmp_path)
		assert [l for l in pd.stdout.decode().split("\n") if l and not l.startswith("@")] == expect, extra
		assert open(tmp_path / "ranges.tsv").read() == "Location\tLength\n" + "".join("%d\t%d\n" % r for r in ranges)

	# Results that outgrow the arrays they were to land in (here every batch's: PVC3_CLI_TINY_RESULTS): the batch is projected again
	# on its own — after the same GPU's next batch, which is already queued and shares the context's alignment scratch and trace
	# buffers, has run (pvc3_pipeline_synchronize).
	for extra in (["--batch-size=64"], ["--devices=0,0", "--batch-size=100"]):
		pt = _run(cmd + extra, cwd=tmp_path, env={"PVC3_CLI_TINY_RESULTS": "1"})
		assert [l for l in pt.stdout.decode().split("\n") if l and not l.startswith("@")] == expect, extra

	# A BGZF-compressed multi-contig FASTA with its .gzi and .fai (what the reference requires: compressed_fasta_reader.cc:163-248):
	# only the destination contig is read, through the indexes; the output does not change.  Also the uncompressed file + .fai.
	decoy = "".join("ACGT"[x] for x in np.random.default_rng(1).integers(0, 4, size=30011))
	fasta_text, fai_lines = "", []
	for name, seq, width in (("decoy1", decoy, 61), ("test", ref_text, 70), ("decoy2", decoy[:999], 50)):
		fasta_text += ">%s some description\n" % name
		fai_lines.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), len(fasta_text), width, width + 1))
		fasta_text += "".join(seq[i:i + width] + "\n" for i in range(0, len(seq), width))
	raw = fasta_text.encode()
	(tmp_path / "multi.fa").write_bytes(raw)
	(tmp_path / "multi.fa.fai").write_text("".join(fai_lines))
	blocks, gzi, c_off = [], [], 0
	for u_off in range(0, len(raw), 4093):
		if u_off: gzi.append((c_off, u_off))
		blk = bamio._bgzf_block(raw[u_off:u_off + 4093])
		blocks.append(blk); c_off += len(blk)
	blocks.append(bamio._bgzf_block(b""))
	(tmp_path / "multi.fa.gz").write_bytes(b"".join(blocks))
	(tmp_path / "multi.fa.gz.gzi").write_bytes(struct.pack("<Q", len(gzi)) + b"".join(struct.pack("<QQ", a, b) for a, b in gzi))
	(tmp_path / "multi.fa.gz.fai").write_text("".join(fai_lines))
	for ref in ("multi.fa.gz", "multi.fa"):
		p2 = _run([c.replace("--reference=genome.fa.gz", "--reference=" + ref) for c in cmd] + ["--debugging-output"], cwd=tmp_path)
		assert [l for l in p2.stdout.decode().split("\n") if l and not l.startswith("@")] == expect
		assert b"1 sequences read through the .fai index" in p2.stderr

	# SAM in (gz), BAM out, one big batch: same records
	subprocess.check_call(["gzip", "-k", "alignments.sam"], cwd=tmp_path)
	_run([tools["panvc3_project_alignments"], "-a", "alignments.sam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-u", "order.txt", "--record-index-tag=XI", "-t", "XS", "-t", "YS", "-b", "-o", "out.bam"], cwd=tmp_path)
	_run([tools["panvc3_project_alignments"], "--io-only", "-a", "out.bam", "-o", "out.sam"], cwd=tmp_path)
	assert [l for l in open(tmp_path / "out.sam").read().split("\n") if l and not l.startswith("@")] == expect
	# out.bam's blocks were compressed on the GPU (the default for BAM output).  The same with groups of two blocks (many groups,
	# both slots in use), on the host's threads, and by zlib: the same records (that the blocks are cut at the same bytes
	# whoever compresses them is tests/test_host_io_cpu.py's).
	import gzip
	to_bam = [tools["panvc3_project_alignments"], "-a", "alignments.sam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-u", "order.txt",
			  "--record-index-tag=XI", "-t", "XS", "-t", "YS", "-b", "--debugging-output"]
	base = (tmp_path / "out.bam").read_bytes()
	for name, extra, env in (("groups.bam", [], {"PVC3_BGZF_GROUP_BLOCKS": "2"}), ("host.bam", ["--bgzf-deflate=host"], None), ("zlib.bam", ["--compression-level=6"], None)):
		pg = _run(to_bam + ["-o", name] + extra, cwd=tmp_path, env=env)
		raw = (tmp_path / name).read_bytes()
		# (the @PG line holds the command line: compare behind the header)
		assert gzip.decompress(raw)[-200000:] == gzip.decompress(base)[-200000:], name
		assert (b"BGZF on the GPU" in pg.stderr) == (name == "groups.bam"), name
		if name == "groups.bam":
			assert b" groups" in pg.stderr and int(pg.stderr.split(b"BGZF on the GPU: ")[1].split(b"blocks in ")[1].split()[0]) >= 2
	# BAM in (the records are framed in place inside the inflated pieces of the file; pieces of 3 000 compressed bytes make
	# them straddle), small batches, SAM out: the same records again
	_run([tools["panvc3_project_alignments"], "--io-only", "-a", "alignments.sam", "-b", "-o", "alignments.bam"], cwd=tmp_path)
	for env, extra in ((None, []), ({"PVC3_BGZF_CHUNK": "3000"}, []), ({"PVC3_BGZF_CHUNK": "3000"}, ["--bgzf-inflate=gpu"]), ({"PVC3_BGZF_CHUNK": "70000"}, ["--debugging-output", "--bgzf-inflate=gpu"])):
		p3 = _run([tools["panvc3_project_alignments"], "-a", "alignments.bam", "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-u", "order.txt",
				   "--record-index-tag=XI", "-t", "XS", "-t", "YS", "--batch-size=64"] + extra, cwd=tmp_path, env=env)
		assert [l for l in p3.stdout.decode().split("\n") if l and not l.startswith("@")] == expect
		assert (b"BGZF input on the GPU" in p3.stderr) == bool(extra and extra[0] == "--debugging-output")
	# ---- the record path on the GPU (--record-path=gpu: the default for a BAM file in and BAM out on one GPU) ----------------------
	# The file's blocks are inflated on the GPU, the records found in the inflated bytes, filtered, scattered into the batch
	# arrays, projected, written anew and compressed without leaving it (include/pvc3_records.h, device_records.hpp).  Behind the
	# header (whose @PG line holds the command line) the uncompressed output must be the bytes the host's threads make
	# (--record-path=host) — for a file whose records straddle its blocks (our writer's, bamio's default), for one written the way
	# htslib writes (no record straddles a block), with pieces of the file so small that records straddle the pieces as well and
	# every piece holds less than four records that pass the filters, and with the blocks inflated by the host's threads.
	def records_behind_header(path):
		raw = gzip.decompress(open(path, "rb").read())
		l_text, = struct.unpack_from("<i", raw, 4)
		q = 8 + l_text
		n_ref, = struct.unpack_from("<i", raw, q); q += 4
		for _ in range(n_ref):
			l_name, = struct.unpack_from("<i", raw, q); q += 4 + l_name + 4
		return raw[q:]
	def summary(stderr):
		text = stderr.decode()
		return [l for l in text.split("\n") if l.startswith("\t") and not l.startswith("\tStages") and not l.startswith("\tTimeline") and not l.startswith("\tRecord path") and not l.startswith("\tBGZF") and not l.startswith("\tteardown")]
	bamio.write_bam(tmp_path / "htslib_style.bam", header, [(n, l) for n, l, e in in_refs], recs, aligned=True, block=20000)
	bamio.write_bam(tmp_path / "small_blocks.bam", header, [(n, l) for n, l, e in in_refs], recs, block=700)		# records of ~ 300 bytes: most blocks hold no record start
	base_cmd = [tools["panvc3_project_alignments"], "-m", "msa-index.dat", "-r", "genome.fa", "-d", "REF", "-s", "/", "-u", "order.txt", "--record-index-tag=XI", "-t", "XS", "-t", "YS", "-b",
				"--output-realigned-ranges=ranges_dev.tsv", "--debugging-output"]
	for name in ("alignments.sorted.bam", "htslib_style.bam", "small_blocks.bam"):
		ph = _run(base_cmd + ["-a", name, "-o", "host_path.bam", "--record-path=host"], cwd=tmp_path)
		want_bytes = records_behind_header(tmp_path / "host_path.bam")
		want_ranges = open(tmp_path / "ranges_dev.tsv").read()
		assert b"Record path on the GPU" not in ph.stderr
		for env, extra in ((None, []), ({"PVC3_BGZF_CHUNK": "3000", "PVC3_RECORDS_STREAM": "65536"}, []), ({"PVC3_BGZF_CHUNK": "70000"}, ["--record-path=gpu", "--bgzf-inflate=gpu"]), (None, ["--bgzf-inflate=host"]),
						   ({"PVC3_RECORDS_GPU_BLOCKS": "3", "PVC3_RECORDS_BLOCKS": "7"}, []), ({"PVC3_RECORDS_GPU_BLOCKS": "2", "PVC3_BGZF_CHUNK": "9000"}, [])):		# (the last two: the host's threads inflate a share of every piece)
			pd = _run(base_cmd + ["-a", name, "-o", "dev_path.bam"] + extra, cwd=tmp_path, env=env)
			assert b"Record path on the GPU" in pd.stderr, (name, env, extra)
			got_bytes = records_behind_header(tmp_path / "dev_path.bam")
			if got_bytes != want_bytes:
				_, _, a = bamio.read_bam(tmp_path / "host_path.bam")
				_, _, b = bamio.read_bam(tmp_path / "dev_path.bam")
				assert len(a) == len(b), (name, env, extra, len(a), len(b))
				for x, y in zip(a, b):
					assert x == y, (name, env, extra, x, y)
				raise AssertionError("the record bytes differ although the parsed records agree: %s %s %s" % (name, env, extra))
			assert open(tmp_path / "ranges_dev.tsv").read() == want_ranges
			assert summary(pd.stderr) == summary(ph.stderr), (name, env, extra)
	_run([tools["panvc3_project_alignments"], "--io-only", "-a", "dev_path.bam", "-o", "dev_path.sam"], cwd=tmp_path)
	assert [l for l in open(tmp_path / "dev_path.sam").read().split("\n") if l and not l.startswith("@")] == expect
	bad = _run(base_cmd + ["-a", "alignments.sam", "-o", "x.bam", "--record-path=gpu"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and b"--record-path=gpu needs a BAM file" in bad.stderr

	text, refs, got = bamio.read_bam(tmp_path / "out.bam")
	assert refs == [("test", 10000)]
	for g in got[:50]:
		span = sum(n for n, op in bamio.parse_cigar(g["cigar"]) if op in "MDN=X")
		assert g["bin"] == bamio.reg2bin(g["pos"], g["pos"] + (span or 1))


@pytest.mark.gpu
def test_convert_bed_positions_cli(tools, built_index, oracle_mod):
	"""convert-bed-positions/main.cc:60-106: both bounds of every interval of the chosen chromosome through project_position."""
	tmp_path, _ = built_index
	rows = load_test_msa()
	o = oracle_mod.Oracle()
	ent = {k: o.add_aligned(rows[k]) for k in ("REF", "1", "2", "3", "4")}
	rng = np.random.default_rng(3)
	n1 = o.zeros(ent["1"])
	lines, expect = ["# a comment", "track name=x", ""], []
	for i in range(3000):
		a = int(rng.integers(0, n1 - 1)); b = int(rng.integers(a, n1))
		if i % 10 == 3:
			lines.append("other\t%d\t%d\tname" % (a, b))
			continue
		lines.append("test\t%d\t%d" % (a, b) + ("\tname%d\t0\t+" % i if i % 3 == 0 else ""))
		expect.append("test\t%d\t%d" % (o.project_position(ent["1"], ent["REF"], a), o.project_position(ent["1"], ent["REF"], b)))
	(tmp_path / "in.bed").write_text("\n".join(lines) + "\n")
	p = _run([tools["panvc3_convert_bed_positions"], "--msa-index=msa-index.dat", "--chr=test", "--src-seq=1", "--dst-seq=REF", "--bed=in.bed"], cwd=tmp_path)
	assert p.stdout.decode().split("\n")[:-1] == expect
	assert b"Chromosome ID matches: %d mismatches: 300." % len(expect) in p.stderr
	# stdin, the other direction, and the error paths
	q = subprocess.run([tools["panvc3_convert_bed_positions"], "-m", "msa-index.dat", "-c", "test", "-s", "REF", "-d", "3"], cwd=tmp_path, input=b"test\t5\t100\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
	assert q.returncode == 0 and q.stdout.decode() == "test\t%d\t%d\n" % (o.project_position(ent["REF"], ent["3"], 5), o.project_position(ent["REF"], ent["3"], 100))
	bad = _run([tools["panvc3_convert_bed_positions"], "-m", "msa-index.dat", "-c", "test", "-s", "nosuch", "-d", "3", "-b", "in.bed"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and "Did not find an entry for sequence ID".encode() in bad.stderr
	(tmp_path / "bad.bed").write_text("test\t5\n")
	bad = _run([tools["panvc3_convert_bed_positions"], "-m", "msa-index.dat", "-c", "test", "-s", "1", "-d", "3", "-b", "bad.bed"], cwd=tmp_path, check=False)
	assert bad.returncode != 0 and b"Parse error in BED input on line 1." in bad.stderr


@pytest.mark.gpu
def test_recalculate_mapq_cli(tools, tmp_path):
	"""bin/panvc3_recalculate_mapq (host I/O + the MAPQ kernels behind the C ABI) against the restatement of the reference
	tool (oracle/recalculate_mapq_tool.py): QNAME groups of paired, unpaired and mixed records with XR / XP / YR / YP tags as
	project_alignments writes them, with and without --rescore-alignments, SAM and BAM.  Parity with the reference binary is
	unpinned (no fixture exists for this tool)."""
	from oracle import recalculate_mapq_tool as tool
	rng = np.random.default_rng(99)
	names = ["chr1", "chr2", "chr3"]
	refs = [(n, 100000) for n in names]
	header = "@HD\tVN:1.6\tSO:unsorted\tGO:query\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "@PG\tID:panvc3.project-alignments.1\tPN:PanVC 3 project_alignments\n"
	recs = []
	for gi in range(600):
		kind = int(rng.integers(0, 10))
		name = "read%d" % gi
		seq1 = "".join("ACGTN"[x] for x in rng.choice(5, size=int(rng.integers(40, 120)), p=[0.245, 0.245, 0.245, 0.245, 0.02]))
		seq2 = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(40, 120))))

		def rec(flag, chrom, pos, seq, rnext, pnext, tags):
			# a CIGAR in = / X form with an indel run now and then, covering the whole read
			n = len(seq)
			a = int(rng.integers(1, n - 4)); b = n - a
			cigar = "%d=1X%d=" % (a, b - 1) if rng.random() < 0.6 else "%d=2I1D%d=" % (a, b - 2)
			ref_len = sum(int(c[:-1]) for c in __import__("re").findall(r"\d+[=XD]", cigar))
			bits = bytearray((ref_len + 7) // 8)
			if rng.random() < 0.3:
				bits[int(rng.integers(0, len(bits)))] |= 1 << int(rng.integers(0, 8))
			qual = "".join(chr(33 + int(q)) for q in rng.integers(2, 42, size=n))
			tags = sorted(tags + [("XN", "B", ("C", list(bits)))], key=lambda t: (ord(t[0][0]), ord(t[0][1])))
			return bamio.Record(name, flag, chrom, pos, int(rng.integers(0, 43)), cigar, rnext, pnext, 0, seq, qual, tags)
		k = int(rng.integers(1, 6))
		if kind >= 2:
			for a in range(k):
				chrom = names[rng.integers(3)]
				p1 = int(rng.integers(1, 400)) * 100; p2 = p1 + 300
				t1 = [("AS", "i", -int(rng.integers(0, 40))), ("XR", "i", 10 + a), ("XP", "i", p1 + a), ("YR", "i", 10 + a), ("YP", "i", p2 + a), ("XI", "i", gi)]
				t2 = [("AS", "i", -int(rng.integers(0, 40))), ("XR", "i", 10 + a), ("XP", "i", p2 + a), ("YR", "i", 10 + a), ("YP", "i", p1 + a), ("XI", "i", gi)]
				if rng.random() < 0.05: t1 = [t for t in t1 if t[0] != "AS"]
				recs.append(rec(99, chrom, p1, seq1, "=", p2, t1))
				if rng.random() < 0.9: recs.append(rec(147, chrom, p2, seq2, "=", p1, t2))
		else:
			for a in range(k):
				chrom = names[rng.integers(3)]
				p1 = int(rng.integers(1, 400)) * 100
				recs.append(rec(0, chrom, p1, seq1, "*", 0, [("AS", "i", -int(rng.integers(0, 40))), ("XR", "i", 10 + a), ("XP", "i", p1)]))
			if kind == 1:
				recs.append(rec(99, names[0], 500, seq1, "=", 800, [("AS", "i", 0), ("XR", "i", 1), ("XP", "i", 500), ("YR", "i", 1), ("YP", "i", 800)]))
		if rng.random() < 0.1:
			recs.append(bamio.Record(name, 4, "*", 0, 0, "*", "*", 0, 0, seq1, "*", []))		# unmapped: dropped
	bamio.write_bam(tmp_path / "in.bam", header, refs, recs)
	exe = os.path.join(BIN, "panvc3_recalculate_mapq")
	for rescore in (False, True):
		want, stats = tool.process(recs, names, rescore=rescore)
		args = [exe, "--alignments=in.bam"] + (["--rescore-alignments"] if rescore else [])
		p = _run(args + ["--output-path=out.sam"], cwd=tmp_path)
		got = [l for l in open(tmp_path / "out.sam").read().split("\n") if l and not l.startswith("@")]
		exp = []
		for r, mapq, tags in want:
			rnext = "=" if r.rnext == r.rname and r.rname != "*" else r.rnext
			exp.append(bamio.Record(r.qname, r.flag, r.rname, r.pos, mapq, r.cigar, rnext, r.pnext, r.tlen, r.seq, r.qual, tags).sam_line())
		assert len(got) == len(exp)
		for a, b in zip(got, exp):
			assert a == b
		err = p.stderr.decode()
		assert "Total alignments: %d" % stats["total_alignments"] in err and "Records with mate missing: %d" % stats["mate_not_found"] in err
		assert "Reads with and without a mate: %d" % stats["reads_with_and_without_mate"] in err and stats["reads_with_and_without_mate"] > 10
		head = [l for l in open(tmp_path / "out.sam").read().split("\n") if l.startswith("@PG")]
		assert head[-1].startswith("@PG\tID:panvc3.recalculate-mapq.1\tPN:PanVC 3 recalculate_mapq") and "PP:panvc3.project-alignments.1" in head[-1]
		# BAM out, small batches: the same records
		_run(args + ["--output-bam", "--output-path=out.bam", "--batch-records=100"], cwd=tmp_path)		# (the blocks compressed on the GPU)
		_run(args + ["--output-bam", "--output-path=out_host.bam", "--batch-records=100", "--bgzf-deflate=host"], cwd=tmp_path)
		import gzip
		assert gzip.decompress((tmp_path / "out.bam").read_bytes())[-20000:] == gzip.decompress((tmp_path / "out_host.bam").read_bytes())[-20000:]
		_, _, back = bamio.read_bam(tmp_path / "out.bam")
		assert [(b["qname"], b["mapq"]) for b in back] == [(r.qname, mapq) for r, mapq, _ in want]
