"""pvc3_split_match_ops (the rewrite_cigar tool's sequence-match mode on the GPU) against oracle/rewrite_cigar_tool.py, the restatement of
rewrite-cigar/main.cc:191-284 and :126-148.  Parity unpinned against the real tool: the reference holds no test or fixture for it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OPS = "MIDNSHP=X"
CODE = {"=": 0, "A": 1, "C": 2, "M": 3, "G": 4, "R": 5, "S": 6, "V": 7, "T": 8, "W": 9, "Y": 10, "H": 11, "K": 12, "D": 13, "B": 14, "N": 15}


def _pack(seq):
	codes = [CODE[c] for c in seq] + [0]
	return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))


def _ops(cigar):
	from oracle import rewrite_cigar_tool as rc
	return np.array([(n << 4) | OPS.index(op) for n, op in rc.parse(cigar)], dtype=np.uint32)


def _text(ops):
	return "".join("%d%s" % (int(c) >> 4, OPS[int(c) & 0xf]) for c in ops)


def test_split_match_ops_against_the_restatement():
	"""Random records over references with N, lower-case and IUPAC letters (the comparison is of raw bytes: dna5::to_char of the read
	against whatever the FASTA holds), every operation type, M runs of every pattern, alignments at the very end of a sequence,
	records without operations, and the ones the tool refuses: an unknown operation, an M past SEQ, an empty M, an alignment past
	the end of its reference, a sequence that does not exist."""
	import panvc3_amd
	from panvc3_amd import api
	from oracle import rewrite_cigar_tool as rc
	rng = np.random.default_rng(101)
	refs = ["".join(rng.choice(list("ACGTNacgtRY"), p=[.22, .22, .22, .22, .05, .01, .01, .01, .01, .015, .015]) for _ in range(L)) for L in (5000, 300, 1, 0)]
	ctx = panvc3_amd.Context(0)
	reference = api.Reference(ctx, refs)
	sequence, pos, cigars, seqs, lens, want = [], [], [], [], [], []
	for k in range(3000):
		which = int(rng.choice([0, 0, 0, 1]))
		ref = refs[which]
		n_ops = int(rng.integers(0, 9))
		parts, q_len, r_len = [], 0, 0
		for _ in range(n_ops):
			op = str(rng.choice(list("MMMMIDNSHP=X")))
			cnt = int(rng.integers(1, 40)) if op != "H" else int(rng.integers(0, 5))
			parts.append((cnt, op))
			if op in "MIS=X": q_len += cnt
			if op in "MDN=X": r_len += cnt
		if r_len > len(ref) or 0 == k % 50: start = int(rng.integers(0, len(ref) + 1))		# (mostly past the end)
		elif 0 == k % 11: start = len(ref) - r_len												# up to the very end
		else: start = int(rng.integers(0, len(ref) - r_len + 1))
		cigar = "".join("%d%s" % p for p in parts)
		seq = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04]) for _ in range(q_len + int(rng.integers(0, 3))))
		# most reads follow the reference so that = runs exist
		q = r = 0
		s = list(seq)
		for cnt, op in parts:
			if op in "M=X":
				for i in range(cnt):
					if start + r + i < len(ref) and rng.random() < 0.8: s[q + i] = ref[start + r + i].upper() if ref[start + r + i].upper() in "ACGTN" else "N"
			if op in "MIS=X": q += cnt
			if op in "MDN=X": r += cnt
		seq = "".join(s)
		sequence.append(which); pos.append(start); cigars.append(_ops(cigar)); seqs.append(_pack(seq)); lens.append(len(seq))
		try:
			if start + r_len > len(ref): raise IndexError
			want.append(rc.sequence_match(cigar, seq, ref, start))
		except IndexError:
			want.append(None)
	# the refused ones
	bad = [(0, 10, np.array([(5 << 4) | 9], dtype=np.uint32), "ACGTA", 1), (0, 10, _ops("4M3M"), "ACGTA", 2), (0, 10, _ops("2M0M2M"), "ACGTA", 3), (1, 298, _ops("5M"), "ACGTA", 4),
		   (7, 0, _ops("5M"), "ACGTA", 4), (2, 0, _ops("2D"), "", 4), (3, 0, _ops("1M"), "A", 4), (3, 0, _ops("3S2I1H"), "ACGTA", 0)]
	for which, p, ops, seq, st in bad:
		sequence.append(which); pos.append(p); cigars.append(ops); seqs.append(_pack(seq)); lens.append(len(seq)); want.append((st,))
	got_ops, got_map, status = api.split_match_ops(ctx, reference, sequence, pos, cigars, seqs, lens)
	for i, w in enumerate(want):
		if w is None:
			assert 4 == status[i] and 0 == len(got_ops[i]) and b"" == got_map[i], i
		elif 1 == len(w):
			assert w[0] == status[i], (i, int(status[i]))
			if w[0]: assert 0 == len(got_ops[i]) and b"" == got_map[i], i
			else: assert "3S2I1H" == _text(got_ops[i]) and b"" == got_map[i]
		else:
			assert 0 == status[i], (i, int(status[i]))
			assert w[0] == _text(got_ops[i]), (i, _text(cigars[i]), w[0], _text(got_ops[i]))
			assert w[1] == got_map[i], (i, w[1], got_map[i])
	# nothing to do
	assert ([], [], []) == tuple(list(x) for x in api.split_match_ops(ctx, reference, [], [], [], [], []))
	reference.close()
