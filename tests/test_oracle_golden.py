"""Pins the CPU oracle (oracle/pvc3_oracle.c) against the reference's own fixtures and property specs.

Fixtures are the reference's data files tests/{rewrite_cigar,alignment_projector,indel_run_checker}_inputs.tsv
(copied verbatim into tests/golden/).  Comparisons use the reference's own notion of equality (cigar_eq,
include/panvc3/cigar.hh:46-152) exactly as its tests do (tests/rewrite_cigar.cc:183, tests/alignment_projector.cc:237),
and additionally record which rows the literal restatement reproduces character for character.
"""
import numpy as np
import pytest

from helpers import (cigar_decode, cigar_encode, cigar_eq, cigar_items, gen_indel_run_case, gen_rewrite_case,
					 indel_run_cases, load_test_msa, rewrite_cases)
from model import column_walk

# Rows the literal restatement reproduces only modulo cigar_eq<true> (P transparent inside an indel run).
REWRITE_NON_LITERAL = {19}
# TSV line 10 expects 1=1X1=1X1= for the same sub-problem (ATG vs TTT) for which line 9 expects 1D1=1D2I:
# the two cannot both hold; line 10 only matches the QUAL-less 0/-1 scoring.
PROJECTOR_CONFLICTING = {10}


def _pair(orc, src, dst):
	o = orc.Oracle()
	s = o.add_aligned(src)
	d = o.add_aligned(dst)
	o.set_reference(d, dst.replace("-", ""))
	return o, s, d


@pytest.mark.parametrize("case", list(rewrite_cases("rewrite_cigar_inputs.tsv")), ids=lambda c: "line%d" % c[0])
def test_rewrite_cigar_fixture(oracle_mod, case):
	ln, src, dst, query, src_pos, exp_pos, cin, cexp = case
	o, s, d = _pair(oracle_mod, src, dst)
	pos, cig = o.rewrite_cigar(s, d, src_pos, cin, query)
	assert pos == exp_pos
	assert cigar_eq(cexp, cig, count_padding=True)	# tests/rewrite_cigar.cc:183
	if ln not in REWRITE_NON_LITERAL:
		assert cigar_decode(cig) == cexp
	# the column-walk formulation (what the HIP kernel implements) agrees literally with the per-base one
	mpos, mcig = column_walk(src, dst, dst.replace("-", ""), src_pos, cigar_items(cin), query)
	assert (mpos, mcig) == (pos, cigar_items(cig))


@pytest.mark.parametrize("score_mode", [0, 1])
@pytest.mark.parametrize("case", list(rewrite_cases("alignment_projector_inputs.tsv")), ids=lambda c: "line%d" % c[0])
def test_alignment_projector_fixture(oracle_mod, case, score_mode):
	ln, src, dst, query, src_pos, exp_pos, cin, cexp = case
	o, s, d = _pair(oracle_mod, src, dst)
	qual = [41] * len(query)	# tests/alignment_projector.cc:223 (max_letter<phred42>), gap costs 0 / -1 (:80-81)
	pos, cig, runs = o.project_alignment(s, d, src_pos, cin, query, qual=qual, gap_open=0, gap_ext=-1, score_mode=score_mode)
	assert pos == exp_pos
	if ln in PROJECTOR_CONFLICTING:
		assert not cigar_eq(cexp, cig, count_padding=True)
		pos2, cig2, _ = o.project_alignment(s, d, src_pos, cin, query, qual=None)
		assert cigar_decode(cig2) == cexp		# matches only seqan3's default 0/-1 nucleotide scoring
		pytest.xfail("fixture line 10 contradicts line 9 in the reference itself")
	assert cigar_eq(cexp, cig, count_padding=True)	# tests/alignment_projector.cc:237
	assert cigar_decode(cig) == cexp				# and literally, under the restated seqan3 tie-break


@pytest.mark.parametrize("case", list(indel_run_cases()), ids=lambda c: "line%d" % c[0])
def test_indel_run_checker_fixture(oracle_mod, case):
	ln, cigar, pieces, has_preceding, has_tail, exp_q, exp_r = case
	o = oracle_mod.Oracle()
	runs, ref_loc, first, second, fq, fr = o.indel_runs(cigar, 0)
	items = cigar_items(cigar)
	# Walk the expectation list the way tests/indel_run_checker.cc:170-227 does.
	idx = 0
	begin = 0
	for k in range(len(runs)):
		if has_preceding:
			kind, text = pieces[idx][0], pieces[idx][1]
			assert kind == "N" and cigar_items(text) == items[begin:first[k]]
			begin = int(first[k]); idx += 1
		p = pieces[idx]
		assert p[0] == "I" and cigar_items(p[1]) == items[begin:second[k]]
		assert (runs[k]["query_location"], runs[k]["query_location"] + runs[k]["query_length"]) == (p[2], p[3])
		assert (ref_loc[k], ref_loc[k] + runs[k]["ref_length"]) == (p[4], p[5])
		begin = int(second[k]); has_preceding = True; idx += 1
	if has_tail:
		assert pieces[idx][0] == "N" and cigar_items(pieces[idx][1]) == items[begin:]
		idx += 1
	assert idx == len(pieces)
	assert (fq, fr) == (exp_q, exp_r)


def test_rank_select_semantics(oracle_mod):
	rng = np.random.default_rng(7)
	for n in (1, 63, 64, 65, 127, 128, 1000):
		s = "".join("-" if x else "A" for x in rng.integers(0, 2, size=n))
		o = oracle_mod.Oracle()
		e = o.add_aligned(s)
		zeros = [i for i, c in enumerate(s) if c != "-"]
		assert o.zeros(e) == len(zeros)
		for i in range(n + 1):
			assert o.rank0(e, i) == o.rank0_naive(e, i) == sum(1 for z in zeros if z < i)
		for k in range(1, len(zeros) + 1):
			assert o.select0(e, k) == o.select0_naive(e, k) == zeros[k - 1]


def test_rewrite_property_restated_rapidcheck(oracle_mod):
	"""tests/rewrite_cigar_arbitrary.cc:1036-1090 restated with a seeded generator."""
	rng = np.random.default_rng(20241016)
	checked = 0
	for it in range(3000):
		c = gen_rewrite_case(rng, int(rng.integers(0, 40)))
		n_src = sum(1 for ch in c["src"] if ch != "-")
		if c["src_pos"] >= n_src:
			continue	# select0 past the end is undefined in the reference too
		# By construction dst_pos counts the dst characters left of the segment.  When the first source character at or
		# after the segment start is preceded by source-gap columns holding dst characters (only possible when the
		# segment section opens with columns that are gaps in the source), select0/rank0 legitimately counts those too.
		col = c["seg_col"]
		skip = False
		while c["src"][col] == "-":
			skip |= c["dst"][col] != "-"
			col += 1
		if skip:
			continue
		o, s, d = _pair(oracle_mod, c["src"], c["dst"])
		pos, cig = o.rewrite_cigar(s, d, c["src_pos"], c["cigar_in"], c["query"])
		assert pos == c["dst_pos"], c
		# The reference's property uses cigar_eq<false>; its by-construction expectation places a pending D before
		# an interleaved P whereas rewrite_cigar() itself emits the P first (as in fixture line 19, which the
		# reference compares with cigar_eq<true>), so P has to be transparent here as well.
		assert cigar_eq(c["cigar_expected"], cig, count_padding=True), (c, cigar_decode(cig))
		mpos, mcig = column_walk(c["src"], c["dst"], c["dst"].replace("-", ""), c["src_pos"], cigar_items(c["cigar_in"]), c["query"])
		assert (mpos, mcig) == (pos, cigar_items(cig)), c
		checked += 1
	assert checked > 2000


def test_indel_run_property_restated_rapidcheck(oracle_mod):
	"""tests/indel_run_checker_arbitrary.cc:487-536 restated (and strengthened to exact equality of the run list)."""
	rng = np.random.default_rng(99)
	o = oracle_mod.Oracle()
	for it in range(2000):
		text, expect = gen_indel_run_case(rng)
		if not text:
			continue
		runs, ref_loc, first, second, fq, fr = o.indel_runs(text, 0)
		assert [(int(a), int(b)) for a, b in zip(first, second)] == expect, text
		items = cigar_items(text)
		assert fq == sum(n for n, op in items if op in "M=XIS")
		assert fr == sum(n for n, op in items if op in "M=XDN")
		for k, (a, b) in enumerate(expect):
			assert runs[k]["query_length"] == sum(n for n, op in items[a:b] if op == "I")
			assert runs[k]["ref_length"] == sum(n for n, op in items[a:b] if op == "D")


def test_collapse_and_buffer(oracle_mod):
	o = oracle_mod.Oracle()
	assert cigar_decode(o.collapse("3=2=1X1X4I")) == "5=2X4I"
	assert cigar_decode(o.collapse("1M")) == "1M"
	assert cigar_decode(o.collapse("")) == ""


def test_align_global_restated_seqan3(oracle_mod):
	o = oracle_mod.Oracle()
	q41 = lambda s: [41] * len(s)
	# sub-problems behind fixture lines 5-7, 9
	assert cigar_decode(o.align_global("ACG", "TTT", q41("TTT"))[0]) == "3D3I"
	assert cigar_decode(o.align_global("ATG", "TTT", q41("TTT"))[0]) == "1D1=1D2I"
	assert cigar_decode(o.align_global("ACG", "ACG", q41("ACG"))[0]) == "3="
	# QUAL-less path: seqan3 nucleotide_scoring_scheme default 0 / -1
	assert cigar_decode(o.align_global("ATG", "TTT", None)[0]) == "1X1=1X"
	# dna5 semantics: case-insensitive, non-ACGT -> N, N == N (quirk Q3)
	assert cigar_decode(o.align_global("acgn", "ACGN", q41("ACGN"))[0]) == "4="
	assert cigar_decode(o.align_global("ACRT", "ACNT", q41("ACNT"))[0]) == "4="
	# empty sides
	assert cigar_decode(o.align_global("", "ACG", q41("ACG"))[0]) == "3I"
	assert cigar_decode(o.align_global("ACG", "", [])[0]) == "3D"
	# Malde scores (align.hh:28-37), orientation values from SURVEY.md §8 a5
	assert abs(o.malde_score(True, 41, 41) - 1.99977) < 1e-4
	assert abs(o.malde_score(False, 41, 41) - (-12.2049)) < 1e-3
	assert abs(o.malde_score(True, 41, 2) - 0.56181) < 1e-4
	assert abs(o.malde_score(False, 41, 2) - (-0.24932)) < 1e-4


def test_align_global_optimal_score_bruteforce(oracle_mod):
	"""The restated DP returns an optimal global alignment score (checked against exhaustive enumeration)."""
	import itertools
	rng = np.random.default_rng(5)
	o = oracle_mod.Oracle()

	def best(ref, qry, qual, go, ge):
		# enumerate all alignments as op strings; affine gap: open + len * ext per maximal gap run of one kind
		memo = {}

		def rec(i, j, last):
			if i == len(ref) and j == len(qry):
				return 0.0
			key = (i, j, last)
			if key in memo:
				return memo[key]
			r = -1e300
			if i < len(ref) and j < len(qry):
				m = ref[i] == qry[j]
				s = o.malde_score(m, 41, qual[j]) if qual is not None else (0.0 if m else -1.0)
				r = max(r, s + rec(i + 1, j + 1, "M"))
			if i < len(ref):
				r = max(r, (0 if last == "D" else go) + ge + rec(i + 1, j, "D"))
			if j < len(qry):
				r = max(r, (0 if last == "I" else go) + ge + rec(i, j + 1, "I"))
			memo[key] = r
			return r
		return rec(0, 0, "M")
	for it in range(150):
		ref = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(0, 7))))
		qry = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(0, 7))))
		qual = [int(x) for x in rng.integers(0, 42, size=len(qry))]
		for go, ge in ((0, -1), (-2, -1), (-5, -2)):
			for ql in (qual, None):
				cig, score = o.align_global(ref, qry, ql, go, ge)
				assert abs(score - best(ref, qry, ql, go, ge)) < 1e-9, (ref, qry, ql, go, ge)
				items = cigar_items(cig)
				assert sum(n for n, op in items if op in "=XD") == len(ref)
				assert sum(n for n, op in items if op in "=XI") == len(qry)


def test_project_alignment_on_shipped_msa(oracle_mod):
	"""Reads sampled from the reference's shipped 5-row test MSA project to the place they came from."""
	rows = load_test_msa()
	assert set(rows) == {"REF", "1", "2", "3", "4"} and len({len(v) for v in rows.values()}) == 1
	o = oracle_mod.Oracle()
	ids = {k: o.add_aligned(v) for k, v in rows.items()}
	ref_text = rows["REF"].replace(b"-", b"")
	o.set_reference(ids["REF"], ref_text)
	assert o.cols(ids["REF"]) == 12636 and o.zeros(ids["REF"]) == 10000
	rng = np.random.default_rng(3)
	for it in range(200):
		k = str(int(rng.integers(1, 5)))
		text = rows[k].replace(b"-", b"")
		L = 100
		p = int(rng.integers(0, len(text) - L))
		read = text[p:p + L].decode().upper()
		pos, cig, runs = o.project_alignment(ids[k], ids["REF"], p, "%dM" % L, read, qual=[30] * L)
		items = cigar_items(cig)
		assert sum(n for n, op in items if op in "=XIS") == L
		assert pos == o.project_position(ids[k], ids["REF"], p)
		# projecting the source sequence onto itself is the identity
		o.set_reference(ids[k], text)
		pos2, cig2, runs2 = o.project_alignment(ids[k], ids[k], p, "%dM" % L, read, qual=[30] * L)
		assert pos2 == p and len(runs2) == 0
		assert all(op in "=X" for n, op in cigar_items(cig2))
