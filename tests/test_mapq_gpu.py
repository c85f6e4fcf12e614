"""recalculate_mapq on the GPU (pvc3_rescore_alignments, pvc3_recalculate_mapq through the C ABI) against the restatement of
recalculate-mapq/main.cc in oracle/recalculate_mapq.py: identical MAPQ for every record of randomly generated QNAME groups
(paired, unpaired, mixed; equal and different projected positions; missing tags and scores), identical rescored alignment
scores.  Parity with the reference binary is unpinned (no fixture exists); the restatement is pinned to the source text by
tests/test_mapq_oracle.py."""
import numpy as np
import pytest

from helpers import BatchBuilder, cigar_items
from oracle import recalculate_mapq as rm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
	import panvc3_amd
	assert panvc3_amd.device_count() > 0, "no GPU visible: the HIP path has no CPU fallback"
	c = panvc3_amd.Context(0)
	yield c
	c.close()


def _random_groups(rng, n_groups):
	"""-> (list of groups of oracle records, dict of ABI arrays, group offsets)"""
	groups = []
	for _ in range(n_groups):
		kind = rng.integers(0, 10)		# 0: unpaired group, 1: mixed (skipped by the reference), else paired
		k = int(rng.integers(1, 9))
		recs = []
		seqs = ["ACGT" * 25, "ACGA" * 25, "A" * int(rng.integers(30, 151))]
		n_places = int(rng.integers(1, 4))
		places = [(int(rng.integers(0, 3)), int(rng.integers(0, 50)) * 100) for _ in range(n_places)]
		if kind >= 2:
			for a in range(k):			# k alignments of the pair: founder a, both mates
				place = places[rng.integers(n_places)]
				l1, l2 = int(rng.integers(50, 151)), int(rng.integers(50, 151))
				s1, s2 = -float(rng.integers(0, 40)), -float(rng.integers(0, 40))
				o1, o2 = (10 + a, place[1] + int(rng.integers(0, 3))), (10 + a, place[1] + 300)
				p1, p2 = (place[0], place[1]), (place[0], place[1] + 300 + int(rng.integers(0, 2)))
				broken = rng.random() < 0.1		# YR / YP point nowhere
				no_as = rng.random() < 0.05
				recs.append(dict(ref_id=p1[0], pos=p1[1], mate_ref_id=p2[0], mate_pos=p2[1], orig=o1, orig_mate=(99, 1) if broken else o2, score=None if no_as else s1, seq="A" * l1))
				if rng.random() < 0.9:		# sometimes the mate's record is missing
					recs.append(dict(ref_id=p2[0], pos=p2[1], mate_ref_id=p1[0], mate_pos=p1[1], orig=o2, orig_mate=o1, score=s2, seq="C" * l2))
			if rng.random() < 0.1:
				recs.append(dict(ref_id=0, pos=7, mate_ref_id=0, mate_pos=400, orig=None, orig_mate=None, score=-3.0, seq="G" * 80))		# tags missing
		else:
			for a in range(k):
				place = places[rng.integers(n_places)]
				recs.append(dict(ref_id=place[0], pos=place[1], mate_ref_id=None, mate_pos=None, orig=(10 + a, place[1]), orig_mate=None,
								 score=-float(rng.integers(0, 40)), seq=seqs[rng.integers(0, 3 if rng.random() < 0.2 else 1)]))
			if kind == 1:
				recs.append(dict(ref_id=0, pos=5, mate_ref_id=0, mate_pos=300, orig=(1, 5), orig_mate=(1, 300), score=0.0, seq=seqs[0]))
		order = rng.permutation(len(recs))
		groups.append([recs[i] for i in order])
	NO = 0xFFFFFFFF
	u = lambda v: NO if v is None else v
	fields = {k: [] for k in ("ref_id", "pos", "mate_ref_id", "mate_pos", "orig_ref_id", "orig_pos", "orig_mate_ref_id", "orig_mate_pos", "l_seq", "seq_class", "score")}
	off = [0]
	for g in groups:
		classes = {}
		for r in g:
			fields["ref_id"].append(u(r["ref_id"])); fields["pos"].append(u(r["pos"]))
			fields["mate_ref_id"].append(u(r["mate_ref_id"])); fields["mate_pos"].append(u(r["mate_pos"]))
			fields["orig_ref_id"].append(NO if r["orig"] is None else r["orig"][0]); fields["orig_pos"].append(NO if r["orig"] is None else r["orig"][1])
			fields["orig_mate_ref_id"].append(NO if r["orig_mate"] is None else r["orig_mate"][0]); fields["orig_mate_pos"].append(NO if r["orig_mate"] is None else r["orig_mate"][1])
			fields["l_seq"].append(len(r["seq"]))
			fields["seq_class"].append(classes.setdefault(r["seq"], len(classes)))
			fields["score"].append(rm.ALIGNMENT_SCORE_MIN if r["score"] is None else r["score"])
		off.append(off[-1] + len(g))
	return groups, fields, np.array(off, dtype=np.uint64)


def test_mapq_groups_match_the_restatement(ctx):
	import panvc3_amd
	rng = np.random.default_rng(2024)
	groups, fields, off = _random_groups(rng, 4000)
	mapq, emitted, stats = panvc3_amd.recalculate_mapq(ctx, off, fields)
	want_stats = dict(total_alignments=0, unpaired_alignments=0, reads_with_and_without_mate=0, mate_not_found=0, reads_without_valid_position=0)
	i = 0
	n_255 = 0
	for gi, g in enumerate(groups):
		ok, st = rm.process_alignment_group(g)
		for k in want_stats:
			want_stats[k] += st[k]
		assert bool(emitted[gi]) == ok, gi
		for r in g:
			if ok:
				assert int(mapq[i]) == r["mapq"], (gi, i, r, [x.get("mapq") for x in g])
				n_255 += r["mapq"] == 255
			i += 1
	assert stats == want_stats
	assert n_255 > 100 and stats["reads_with_and_without_mate"] > 100 and stats["mate_not_found"] > 100
	# an empty batch and empty groups
	m, e, st = panvc3_amd.recalculate_mapq(ctx, np.zeros(1, np.uint64), {k: np.zeros(0) for k in fields})
	assert m.size == 0 and e.size == 0 and st["total_alignments"] == 0


def test_rescored_alignments_match_the_restatement(ctx):
	import panvc3_amd
	rng = np.random.default_rng(7)
	bb = BatchBuilder()
	xn_off, xn, want, want_status = [0], [], [], []
	ops = "=XIDSHNP"
	for it in range(3000):
		n_ops = int(rng.integers(1, 12))
		cig, qlen, rlen = [], 0, 0
		for _ in range(n_ops):
			op = ops[rng.integers(len(ops))] if rng.random() < 0.9 or it % 50 else "M"
			cnt = int(rng.integers(1, 30))
			cig.append((cnt, op))
			qlen += cnt if op in "=XIS" else 0
			rlen += cnt if op in "=XDNM" else 0
		qlen_total = qlen + (sum(c for c, o in cig if o == "M"))
		seq = "".join("ACGTN"[x] for x in rng.choice(5, size=qlen_total, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
		qual = [int(x) for x in rng.integers(0, 60, size=qlen_total)]
		short = rng.random() < 0.03		# an XN bit map that does not cover the alignment
		bits = bytes(rng.integers(0, 256, size=(0 if short else (rlen + 7) // 8 + 1), dtype=np.uint8) & rng.integers(0, 256, size=(0 if short else (rlen + 7) // 8 + 1), dtype=np.uint8) & 0x11)
		text = "".join("%d%s" % c for c in cig)
		bb.add(0, 0, 0, text, seq, qual=qual)
		xn.extend(bits); xn_off.append(len(xn))
		try:
			want.append(rm.cigar_score(cig, seq, [min(q, 41) for q in qual], bits)); want_status.append(0)
		except ValueError:
			want.append(0.0); want_status.append(1)
		except IndexError:
			want.append(0.0); want_status.append(2)
	batch = bb.build()
	score, status = panvc3_amd.rescore_alignments(ctx, batch, xn_off, np.array(xn, dtype=np.uint8))
	assert status.tolist() == want_status
	assert score.tolist() == want
	assert sum(want_status) > 20
	# other penalties
	score2, _ = panvc3_amd.rescore_alignments(ctx, batch, xn_off, np.array(xn, dtype=np.uint8), 1, 9, 2, 7, 1)
	want2 = []
	for i in range(len(want)):
		if want_status[i]:
			want2.append(0.0)
			continue
		c0, c1 = int(batch["cigar_off"][i]), int(batch["cigar_off"][i + 1])
		cig = cigar_items(batch["cigar"][c0:c1])
		q0, q1 = int(batch["qual_off"][i]), int(batch["qual_off"][i + 1])
		s0 = int(batch["seq_off"][i])
		codes = batch["seq4"][s0:int(batch["seq_off"][i + 1])]
		seq = "".join("=ACMGRSVTWYHKDBN"[(codes[j >> 1] >> (0 if j & 1 else 4)) & 0xf] for j in range(q1 - q0))
		want2.append(rm.cigar_score(cig, seq, [min(int(q), 41) for q in batch["qual"][q0:q1]], bytes(xn[xn_off[i]:xn_off[i + 1]]), 1, 9, 2, 7, 1))
	assert score2.tolist() == want2
