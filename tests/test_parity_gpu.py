"""Parity of the HIP path (through the C ABI) against the CPU oracle: bit-exact on every output.

These tests need a real MI355X (pytest -m gpu).  They read like the reference's tests/alignment_projector.cc and
tests/rewrite_cigar.cc: build a sequence entry pair from two gapped strings, project, compare position + CIGAR.
"""
import numpy as np
import pytest

from helpers import (BatchBuilder, assert_results_equal, cigar_decode, cigar_eq, cigar_items, gen_indel_run_case, gen_rewrite_case,
					 load_test_msa, rewrite_cases)

pytestmark = pytest.mark.gpu


def _setup_pair(pvc3, oracle_mod, src, dst, ctx=None):
	idx = pvc3.Index()
	s, d = idx.add_aligned(src), idx.add_aligned(dst)
	idx.set_reference(d, dst.replace("-", ""))
	ctx = ctx or pvc3.Context(0)
	ctx.upload_index(idx)
	o = oracle_mod.Oracle()
	os_, od = o.add_aligned(src), o.add_aligned(dst)
	o.set_reference(od, dst.replace("-", ""))
	assert (os_, od) == (s, d)
	return ctx, idx, o, s, d


@pytest.mark.parametrize("case", list(rewrite_cases("alignment_projector_inputs.tsv")), ids=lambda c: "line%d" % c[0])
def test_alignment_projector_fixture(pvc3, oracle_mod, case):
	"""tests/alignment_projector.cc:157-247 against the HIP path."""
	ln, src, dst, query, src_pos, exp_pos, cin, cexp = case
	ctx, idx, o, s, d = _setup_pair(pvc3, oracle_mod, src, dst)
	qual = [41] * len(query)
	projector = pvc3.AlignmentProjector(ctx)
	for mode in (pvc3.SCORE_FP64, pvc3.SCORE_INT32_TRUNC):
		pos = projector.project_alignment(src_pos, s, d, query, cin, qual, 0, -1, score_mode=mode)
		opos, ocig, oruns = o.project_alignment(s, d, src_pos, cin, query, qual=qual, score_mode=mode)
		assert pos == exp_pos == opos
		assert cigar_decode(projector.alignment()) == cigar_decode(ocig)
		assert projector.realigned_reference_ranges() == [(int(r["ref_position"]), int(r["ref_length"])) for r in oruns]
		assert projector.realigned_query_ranges() == [(int(r["query_location"]), int(r["query_length"])) for r in oruns]
		if ln != 10:	# line 10 contradicts line 9 in the reference itself
			assert cigar_eq(cexp, projector.alignment(), count_padding=True)
	# QUAL-less records take the 0 / -1 nucleotide scoring path
	pos = projector.project_alignment(src_pos, s, d, query, cin, None, 0, -1)
	opos, ocig, _ = o.project_alignment(s, d, src_pos, cin, query, qual=None)
	assert (pos, cigar_decode(projector.alignment())) == (opos, cigar_decode(ocig))


@pytest.mark.parametrize("case", list(rewrite_cases("rewrite_cigar_inputs.tsv")), ids=lambda c: "line%d" % c[0])
def test_rewrite_cigar_fixture(pvc3, oracle_mod, case):
	"""tests/rewrite_cigar.cc:111-191; where the rewritten CIGAR has no I+D run the projector returns it unchanged."""
	ln, src, dst, query, src_pos, exp_pos, cin, cexp = case
	ctx, idx, o, s, d = _setup_pair(pvc3, oracle_mod, src, dst)
	projector = pvc3.AlignmentProjector(ctx)
	pos = projector.project_alignment(src_pos, s, d, query, cin, [30] * len(query))
	opos, ocig, oruns = o.project_alignment(s, d, src_pos, cin, query, qual=[30] * len(query))
	assert pos == exp_pos == opos
	assert cigar_decode(projector.alignment()) == cigar_decode(ocig)
	if len(oruns) == 0:
		assert cigar_eq(cexp, projector.alignment(), count_padding=True)


def test_restated_rapidcheck_property_batch(pvc3, oracle_mod):
	"""The seeded restatement of tests/rewrite_cigar_arbitrary.cc, all cases in one batch over one multi-entry index."""
	rng = np.random.default_rng(20241016)
	idx = pvc3.Index()
	o = oracle_mod.Oracle()
	bb = BatchBuilder()
	for it in range(1500):
		c = gen_rewrite_case(rng, int(rng.integers(0, 60)))
		if c["src_pos"] >= sum(1 for ch in c["src"] if ch != "-"):
			continue
		s, d = idx.add_aligned(c["src"]), idx.add_aligned(c["dst"])
		assert (o.add_aligned(c["src"]), o.add_aligned(c["dst"])) == (s, d)
		text = c["dst"].replace("-", "")
		idx.set_reference(d, text); o.set_reference(d, text)
		qual = [int(x) for x in rng.integers(0, 45, size=len(c["query"]))] if rng.integers(4) else None
		bb.add(c["src_pos"], s, d, c["cigar_in"], c["query"], qual=qual, mate=(s, d, int(rng.integers(0, c["src_pos"] + 1))))
	batch = bb.build()
	ctx = pvc3.Context(0)
	ctx.upload_index(idx)
	for mode in (0, 1):
		got, stats = pvc3.project_batch(ctx, batch, 0, -1, mode)
		want = o.batch_project(batch, 0, -1, mode)
		assert_results_equal(batch, got, want)
		assert stats["n_failed"] == 0 and stats["n_realigned_records"] > 100
	for go, ge in ((-3, -1), (-5, -2), (0, -4)):
		got, _ = pvc3.project_batch(ctx, batch, go, ge, 0)
		assert_results_equal(batch, got, o.batch_project(batch, go, ge, 0))


@pytest.mark.parametrize("fixture", ["alignment_projector_inputs.tsv", "rewrite_cigar_inputs.tsv"])
def test_cpp_mirror_header_on_reference_fixtures(pvc3, oracle_mod, fixture):
	"""include/pvc3.hpp (the C++ mirror of panvc3::alignment_projector over the C ABI) compiled into a program and run: the
	reference's tests/alignment_projector.cc:62-84 loop on the rows of its own fixtures, compared with the fixture's
	expectations through the reference's comparator and with the oracle string for string."""
	import os
	import subprocess
	from helpers import GOLDEN, ROOT
	if "default" != pvc3.paths_mode:
		pytest.skip("the program uses the library's default configuration")
	exe = os.path.join(ROOT, "bin", "test_alignment_projector_fixture")
	assert os.path.exists(exe), "bin/test_alignment_projector_fixture is built by __graft_entry__.build() / python -m panvc3_amd.build"
	out = subprocess.run([exe, os.path.join(GOLDEN, fixture)], capture_output=True, text=True, timeout=300)
	assert out.returncode == 0, out.stderr
	got = {int(f[0]): f for f in (line.split("\t") for line in out.stdout.strip().split("\n"))}
	cases = list(rewrite_cases(fixture))
	assert sorted(got) == [c[0] for c in cases]
	for ln, src, dst, query, src_pos, exp_pos, cin, cexp in cases:
		f = got[ln]
		o = oracle_mod.Oracle()
		s, d = o.add_aligned(src), o.add_aligned(dst)
		o.set_reference(d, dst.replace("-", ""))
		opos, ocig, oruns = o.project_alignment(s, d, src_pos, cin, query, qual=[41] * len(query))
		assert int(f[1]) == exp_pos == opos
		assert f[2] == cigar_decode(ocig)
		assert f[3] == "".join("%d,%d;" % (r["ref_position"], r["ref_length"]) for r in oruns)
		assert f[4] == "".join("%d,%d;" % (r["query_location"], r["query_length"]) for r in oruns)
		if not (fixture.startswith("alignment_projector") and ln == 10) and (fixture.startswith("alignment_projector") or len(oruns) == 0):
			assert cigar_eq(cexp, f[2], count_padding=True)		# line 10 contradicts line 9 in the reference itself
