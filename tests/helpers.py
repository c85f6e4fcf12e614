"""Shared test helpers: fixture loaders, the reference's CIGAR comparator, seeded generators.

Nothing here is product code.  Generators restate the reference's RapidCheck generators
(tests/rewrite_cigar_arbitrary.cc:231-845, tests/indel_run_checker_arbitrary.cc:248-340) as seeded numpy code.
"""
import bz2
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CIGAR_OPS = "MIDNSHP=X"
CIGAR_OPS_PADDED = "MIDNSHP=X???????"


def cigar_encode(text):
	out, num = [], ""
	for ch in text:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num) << 4) | CIGAR_OPS.index(ch))
			num = ""
	assert num == ""
	return np.array(out, dtype=np.uint32)


def cigar_decode(arr):
	return "".join("%d%s" % (int(v) >> 4, CIGAR_OPS_PADDED[int(v) & 0xf]) for v in arr)


def cigar_items(c):
	"""-> list of (count, op_char)"""
	if isinstance(c, str):
		c = cigar_encode(c)
	return [(int(v) >> 4, CIGAR_OPS[int(v) & 0xf]) for v in c]


def cigar_eq(lhs, rhs, count_padding=False):
	"""The reference's comparator (include/panvc3/cigar.hh:46-152): within a run of indels only the total
	I and D counts are compared; with count_padding, P is transparent inside such runs."""
	lhs, rhs = cigar_items(lhs), cigar_items(rhs)

	def count_indels(items, i):
		ins = dele = 0
		while i < len(items):
			cnt, op = items[i]
			if op == "I":
				ins += cnt
			elif op == "D":
				dele += cnt
			elif count_padding and op == "P":
				pass
			else:
				return i, ins, dele, True
			i += 1
		return i, ins, dele, False

	li = ri = 0
	while True:
		li, lins, ldel, r1 = count_indels(lhs, li)
		ri, rins, rdel, r2 = count_indels(rhs, ri)
		if (lins, ldel) != (rins, rdel):
			return False
		if not r1 and not r2:
			return True
		if r1 != r2:
			return False
		if lhs[li] != rhs[ri]:
			return False
		li += 1
		ri += 1


def load_tsv(name):
	rows = []
	with open(os.path.join(GOLDEN, name)) as f:
		for ln, line in enumerate(f, 1):
			line = line.rstrip("\n")
			if line:
				rows.append((ln, line.split("\t")))
	return rows


def rewrite_cases(name):
	"""Yields (tsv_line, src_aligned, dst_aligned, query, src_pos, expected_dst_pos, cigar_in, cigar_expected)."""
	src = dst = None
	for ln, f in load_tsv(name):
		if f[0] == "S":
			src, dst = f[2], f[3]
		else:
			yield ln, src, dst, f[1], int(f[2]), int(f[3]), f[4], f[5]


def indel_run_cases():
	"""Yields (tsv_line, cigar, expected_runs [(cigar, q0, q1, r0, r1)], expected_pieces, has_tail, q_pos, r_pos)."""
	pieces = []
	for ln, f in load_tsv("indel_run_checker_inputs.tsv"):
		if f[0] == "N":
			pieces.append(("N", f[1]))
		elif f[0] == "I":
			pieces.append(("I", f[1], int(f[2]), int(f[3]), int(f[4]), int(f[5])))
		else:
			yield ln, f[1], pieces, f[2] == "T", f[3] == "T", int(f[4]), int(f[5])
			pieces = []


def load_test_msa():
	"""The reference's shipped 5-row test MSA (test-workflow/index-input/, copied as data)."""
	d = os.path.join(GOLDEN, "test_msa")
	rows = {}
	with open(os.path.join(d, "input.tsv")) as f:
		for line in f:
			chr_id, seq_id, path = line.rstrip("\n").split("\t")
			with bz2.open(os.path.join(d, os.path.basename(path)), "rb") as g:
				rows[seq_id] = g.read().rstrip(b"\n")
	return rows


# --- seeded restatement of tests/rewrite_cigar_arbitrary.cc ------------------------------------------

NUC = "ACGT"


def _next(c):
	return NUC[(NUC.index(c) + 1) % 4]


class _Buf:
	"""cigar_buffer restated for building expectations (cigar.cc:54-95)."""

	def __init__(self):
		self.items = []

	def push(self, op, n=1):
		if n == 0:
			return
		if self.items and self.items[-1][1] == op:
			self.items[-1][0] += n
		else:
			self.items.append([n, op])

	def text(self):
		return "".join("%d%s" % (n, op) for n, op in self.items)


def gen_rewrite_case(rng, size):
	"""One msa_builder sample: returns dict(src, dst aligned strings, query, src_pos, dst_pos, cigar_in, cigar_expected)."""
	def non_segment(n):
		src, dst = [], []
		sp = dp = 0
		for _ in range(n):
			cc = NUC[rng.integers(4)]
			op = rng.integers(4)
			if op == 0:
				src.append(cc); dst.append(cc); sp += 1; dp += 1
			elif op == 1:
				src.append(cc); dst.append(_next(cc)); sp += 1; dp += 1
			elif op == 2:
				src.append(cc); dst.append("-"); sp += 1
			else:
				src.append("-"); dst.append(cc); dp += 1
		return "".join(src), "".join(dst), sp, dp

	def boundary(n):
		buf, seg = _Buf(), []
		for _ in range(n):
			op = "SHP"[rng.integers(3)]
			buf.push(op)
			if op == "S":
				seg.append(NUC[rng.integers(4)])
		return "".join(seg), buf

	# segment section (12 op types, rewrite_cigar_arbitrary.cc:369-387)
	ops = [int(x) for x in rng.integers(0, 12, size=size)]
	MATCHLIKE = {0, 1, 2, 3}
	DELETION = {7, 8, 9}

	def has_del_on_boundary(seq):
		for op in seq:
			if op in MATCHLIKE:
				return False
			if op in DELETION:
				return True
		return False
	if has_del_on_boundary(ops):
		ops.insert(0, int(rng.integers(4)))
	if has_del_on_boundary(reversed(ops)):
		ops.append(int(rng.integers(4)))

	seg_q, seg_src, seg_dst = [], [], []
	cin, cexp = _Buf(), _Buf()
	for op in ops:
		if op == 11:	# padding
			cin.push("P"); cexp.push("P")
			continue
		cc = NUC[rng.integers(4)]
		if op == 0:   trip, i, e = (cc, cc, cc), "=", "="
		elif op == 1: trip, i, e = (cc, cc, _next(cc)), "=", "X"
		elif op == 2: trip, i, e = (cc, _next(cc), cc), "X", "="
		elif op == 3: trip, i, e = (cc, _next(cc), _next(_next(cc))), "X", "X"
		elif op == 4: trip, i, e = (cc, cc, "-"), "=", "I"
		elif op == 5: trip, i, e = (cc, _next(cc), "-"), "X", "I"
		elif op == 6: trip, i, e = (cc, "-", "-"), "I", "I"
		elif op == 7: trip, i, e = ("-", "-", cc), None, "D"		# dst_deletion
		elif op == 8: trip, i, e = ("-", cc, cc), "D", "D"
		elif op == 9: trip, i, e = ("-", cc, _next(cc)), "D", "D"
		else:         trip, i, e = ("-", cc, "-"), "D", None		# src_deletion (op 10)
		if trip[0] != "-":
			seg_q.append(trip[0])
		if not (trip[1] == "-" and trip[2] == "-"):	# alignment_without_segment drops all-gap columns
			seg_src.append(trip[1]); seg_dst.append(trip[2])
		if i: cin.push(i)
		if e: cexp.push(e)

	lsrc, ldst, sp, dp = non_segment(int(rng.integers(0, size + 1)))
	rsrc, rdst, _, _ = non_segment(int(rng.integers(0, size + 1)))
	lseg, lbuf = boundary(int(rng.integers(0, max(1, size // 4) + 1)))
	rseg, rbuf = boundary(int(rng.integers(0, max(1, size // 4) + 1)))

	def join(*bufs):
		# The reference concatenates the three sections' CIGARs without merging (rsv::concat); adjacent equal
		# operations across a section boundary are merged here, as cigar_buffer does on the actual output.
		merged = _Buf()
		for b in bufs:
			for n, op in b.items:
				merged.push(op, n)
		return merged.text()
	return dict(
		src=lsrc + "".join(seg_src) + rsrc,
		dst=ldst + "".join(seg_dst) + rdst,
		query=lseg + "".join(seg_q) + rseg,
		src_pos=sp, dst_pos=dp, seg_col=len(lsrc),
		cigar_in=join(lbuf, cin, rbuf), cigar_expected=join(lbuf, cexp, rbuf),
	)


def gen_indel_run_case(rng):
	"""Restates tests/indel_run_checker_arbitrary.cc: a chain of (indel | non-indel) runs separated by boundary
	runs; op counts 1..14.  Returns (cigar_text, [(first_op_idx, end_op_idx)] of the indel runs with both I and D)."""
	ops, expect = [], []

	def cnt():
		return int(rng.integers(1, 15))
	n_runs = int(rng.integers(0, 7))
	for r in range(n_runs):
		if rng.integers(2):		# indel run: both I and D, adjacent ops of distinct type
			k = int(rng.integers(2, 6))
			first = "ID"[rng.integers(2)]
			start = len(ops)
			for i in range(k):
				ops.append((cnt(), first if i % 2 == 0 else ("D" if first == "I" else "I")))
			expect.append((start, len(ops)))
		else:					# non-indel run: boundary ops and single-type indels
			k = int(rng.integers(1, 5))
			prev = None
			for i in range(k):
				choices = [c for c in "M=XSNHP" if c != prev] + [c for c in "ID" if prev not in ("I", "D")]
				op = choices[rng.integers(len(choices))]
				ops.append((cnt(), op))
				prev = op
			# make sure the run ends on a boundary op so that it does not fuse with a following indel run
			if ops[-1][1] in "ID":
				ops.append((cnt(), "M"))
		# boundary run
		ops.append((cnt(), "M=XSNHP"[rng.integers(7)]))
	# drop expectations that fused with a preceding single-type indel (cannot happen by construction) and merge adjacent equal ops
	text = "".join("%d%s" % (n, op) for n, op in ops)
	return text, expect


# --- batches ---------------------------------------------------------------------------------------------------------

_SEQ4 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


class BatchBuilder:
	"""Accumulates records into the SoA batch of include/pvc3.h (same dict for the oracle and the HIP path)."""

	def __init__(self):
		self.src_pos, self.src_entry, self.dst_entry, self.flags = [], [], [], []
		self.cigar_off, self.cigar = [0], []
		self.seq_off, self.l_seq, self.seq4 = [0], [], []
		self.qual_off, self.qual = [0], []
		self.mate_src_entry, self.mate_dst_entry, self.mate_pos = [], [], []

	def add(self, src_pos, src_entry, dst_entry, cigar, query, qual=None, mate=None, seq_codes=None):
		cig = cigar_encode(cigar) if isinstance(cigar, str) else np.asarray(cigar, dtype=np.uint32)
		self.src_pos.append(src_pos); self.src_entry.append(src_entry); self.dst_entry.append(dst_entry)
		self.cigar.extend(int(x) for x in cig)
		self.cigar_off.append(len(self.cigar))
		codes = list(seq_codes) if seq_codes is not None else [_SEQ4.get(c.upper(), 15) for c in query]
		n = len(codes)
		if n & 1:
			codes = codes + [0]
		self.seq4.extend((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
		self.seq_off.append(len(self.seq4))
		self.l_seq.append(n)
		fl = 0
		if qual is not None:
			assert len(qual) == n
			fl |= 1
			self.qual.extend(int(x) for x in qual)
		self.qual_off.append(len(self.qual))
		if mate is not None:
			fl |= 2
			self.mate_src_entry.append(mate[0]); self.mate_dst_entry.append(mate[1]); self.mate_pos.append(mate[2])
		else:
			self.mate_src_entry.append(0); self.mate_dst_entry.append(0); self.mate_pos.append(0)
		self.flags.append(fl)

	def build(self):
		u32, u64, u8 = np.uint32, np.uint64, np.uint8
		return dict(
			src_pos=np.array(self.src_pos, dtype=u32), src_entry=np.array(self.src_entry, dtype=u32), dst_entry=np.array(self.dst_entry, dtype=u32),
			flags=np.array(self.flags, dtype=u32), cigar_off=np.array(self.cigar_off, dtype=u64), cigar=np.array(self.cigar, dtype=u32),
			seq_off=np.array(self.seq_off, dtype=u64), l_seq=np.array(self.l_seq, dtype=u32), seq4=np.array(self.seq4, dtype=u8),
			qual_off=np.array(self.qual_off, dtype=u64), qual=np.array(self.qual, dtype=u8),
			mate_src_entry=np.array(self.mate_src_entry, dtype=u32), mate_dst_entry=np.array(self.mate_dst_entry, dtype=u32),
			mate_pos=np.array(self.mate_pos, dtype=u32),
		)


def describe_record(batch, res, i):
	c0, c1 = int(res["out_cigar_off"][i]), int(res["out_cigar_off"][i + 1])
	r0, r1 = int(res["run_off"][i]), int(res["run_off"][i + 1])
	return dict(
		status=int(res["status"][i]), dst_pos=int(res["dst_pos"][i]), mate=int(res["mate_dst_pos"][i]),
		cigar=cigar_decode(res["out_cigar"][c0:c1]), runs=[tuple(int(x) for x in r) for r in res["runs"][r0:r1]],
		cigar_in=cigar_decode(batch["cigar"][int(batch["cigar_off"][i]):int(batch["cigar_off"][i + 1])]), src_pos=int(batch["src_pos"][i]),
	)


def assert_results_equal(batch, got, want):
	"""Byte-equality of every output array (bit-exact parity); on a mismatch shows the first differing record."""
	n = batch["src_pos"].size
	keys = ("status", "dst_pos", "mate_dst_pos", "out_cigar_off", "run_off", "out_cigar", "runs")
	if all(np.array_equal(got[k], want[k]) for k in keys):
		return
	for i in range(n):
		a, b = describe_record(batch, got, i), describe_record(batch, want, i)
		assert a == b, "record %d differs:\n  hip    %r\n  oracle %r" % (i, a, b)
	raise AssertionError("arrays differ although every record compares equal")
