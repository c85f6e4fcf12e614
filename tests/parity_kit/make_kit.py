#!/usr/bin/env python3
"""Makes the parity kit (this directory's data files): a tiny MSA, its reference FASTA and a SAM file of reads chosen so that the
realignment policies nobody could pin without the real binary give DIFFERENT output records, and the expected output of
project_alignments under every policy.

What is unpinned: the arithmetic of include/panvc3/align.hh:59-112 lives in seqan3, which is not part of the
reference snapshot — (U1) whether the Malde scores are used as doubles or truncated to int32 (score type), and (U2) which of the
co-optimal alignments the traceback returns (tie-break order).  This repository implements both score types and twelve
tie-break orders (include/pvc3.h: pvc3_params); ONE run of the real panvc3_project_alignments on reads.sam tells which is
seqan3's (README.md here).

Every case is one read over one place of the MSA where a block of founder bases (absent from REF) lies next to a block of
REF bases (absent from the founder): the rewritten CIGAR has an insertion next to a deletion there, which the reference
realigns (libpanvc3/alignment_projector.cc:58-137).  The blocks and the read's base qualities are found by search: a case is
kept if it tells apart two policies that no earlier case told apart.

Deterministic (seeded).  Test infrastructure: uses oracle/ (the CPU restatement), nothing of the product.
usage: python tests/parity_kit/make_kit.py [output directory]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

POLICIES = [(s, t) for s in (0, 1) for t in range(12)]		# (score_mode, tiebreak_mode)
FLANK = 25
QUALS = (2, 12, 23, 37, 41)


def policy_name(p):
	return "score%d_tiebreak%02d" % p


def find_cases(o, max_cases=40, tries=400000, seed=20241016):
	"""[(ref block, founder block, qualities of the founder block's bases)]: greedily, cases that separate policy pairs."""
	from oracle.oracle import cigar_decode
	rng = np.random.default_rng(seed)
	unseparated = {(a, b) for i, a in enumerate(POLICIES) for b in POLICIES[i + 1:]}
	cases = []
	for _ in range(tries):
		if not unseparated or len(cases) >= max_cases:
			break
		m, n = int(rng.integers(1, 9)), int(rng.integers(1, 9))
		alphabet = "ACGT"[:int(rng.integers(2, 5))]		# small alphabets make ties
		r = "".join(alphabet[x] for x in rng.integers(0, len(alphabet), size=m))
		q = "".join(alphabet[x] for x in rng.integers(0, len(alphabet), size=n))
		quals = [int(x) for x in rng.choice(QUALS, size=n)] if rng.random() < 0.7 else [int(rng.choice(QUALS))] * n
		out = {}
		for p in sorted({x for pair in unseparated for x in pair}):		# (only the policies that still have a twin)
			cig, _ = o.align_global(r, q, quals, 0, -1, p[0], p[1])
			out[p] = cigar_decode(cig)
		newly = {(a, b) for (a, b) in unseparated if out[a] != out[b]}
		# worth a read: it separates at least three new pairs, or one of the last few
		if len(newly) >= 3 or (newly and len(unseparated) <= 12):
			cases.append((r, q, quals))
			unseparated -= newly
	return cases, unseparated


def build(out_dir):
	from oracle import oracle as orc
	from oracle import project_records as pr
	import bamio
	o = orc.Oracle()
	cases, unseparated = find_cases(o)
	rng = np.random.default_rng(7)
	seg = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
	# ---- the MSA: REF and one founder, the cases one behind the other between flanks of bases both share ---------------------------
	ref_row, f_row, reads = [], [], []
	col = 0
	f_pos = 0		# position in the unaligned founder
	for k, (r, q, quals) in enumerate(cases):
		left, right = seg(FLANK), seg(FLANK)
		ref_row += [left, "-" * len(q), r, right]
		f_row += [left, q, "-" * len(r), right]
		read_seq = left + q + right
		read_qual = [41] * FLANK + quals + [41] * FLANK
		reads.append(bamio.Record("case%02d" % k, 0, "chrK/F1", f_pos + 1, 42, "%dM" % len(read_seq), "*", 0, 0, read_seq, "".join(chr(33 + x) for x in read_qual),
								  tags=[("NM", "i", 0), ("AS", "i", -3 - k)]))
		f_pos += len(read_seq)
	ref_aln, f_aln = "".join(ref_row), "".join(f_row)
	assert len(ref_aln) == len(f_aln)
	ref_text, f_text = ref_aln.replace("-", ""), f_aln.replace("-", "")
	os.makedirs(os.path.join(out_dir, "expected"), exist_ok=True)
	with open(os.path.join(out_dir, "msa.a2m"), "w") as f:
		f.write(">chrK\tREF\n%s\n>chrK\tF1\n%s\n" % (ref_aln, f_aln))
	with open(os.path.join(out_dir, "genome.fa"), "w") as f:
		f.write(">chrK\n")
		for i in range(0, len(ref_text), 60):
			f.write(ref_text[i:i + 60] + "\n")
	header = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chrK/REF\tLN:%d\n@SQ\tSN:chrK/F1\tLN:%d\n@PG\tID:parity-kit\tPN:make_kit.py\n" % (len(ref_text), len(f_text))
	bamio.write_sam(os.path.join(out_dir, "reads.sam"), header, reads)
	# ---- expected records under every policy -------------------------------------------------------------------------------------
	in_refs = [("chrK/REF", len(ref_text), ""), ("chrK/F1", len(f_text), "")]
	e_ref, e_f = o.add_aligned(ref_aln), o.add_aligned(f_aln)
	o.set_reference(e_ref, ref_text)
	out_refs, mapping, _ = pr.process_headers(in_refs, "/", "REF", None)
	opts = dict(preserve=set(), rname_tag="XR", pos_tag="XP", rnext_tag="YR", pnext_tag="YP", query_ranges_tag="XB", ref_ranges_tag="YB", rec_idx_tag="XI")
	table = {}
	for p in POLICIES:
		lines, _ = pr.project_records(reads, in_refs, o, [e_ref, e_f], [e_ref], mapping, out_refs, dict(opts, score_mode=p[0], tiebreak_mode=p[1]))
		with open(os.path.join(out_dir, "expected", policy_name(p) + ".sam"), "w") as f:
			f.write("".join(l + "\n" for l in lines))
		table[p] = [l.split("\t")[5] for l in lines]
	# ---- which policies the kit cannot tell apart, and the CIGAR table ----------------------------------------------------------------
	groups = {}
	for p in POLICIES:
		groups.setdefault(tuple(table[p]), []).append(p)
	with open(os.path.join(out_dir, "policies.tsv"), "w") as f:
		f.write("# groups of policies (score_mode: 0 double, 1 int32-truncated; tiebreak_mode 0..11, include/pvc3.h) with identical output on this kit\n")
		for g in groups.values():
			f.write(" ".join(policy_name(p) for p in g) + "\n")
	with open(os.path.join(out_dir, "cigars.tsv"), "w") as f:
		f.write("read\tref_block\tfounder_block\tqualities\t" + "\t".join(policy_name(p) for p in POLICIES) + "\n")
		for k, (r, q, quals) in enumerate(cases):
			f.write("case%02d\t%s\t%s\t%s\t" % (k, r, q, ",".join(str(x) for x in quals)) + "\t".join(table[p][k] for p in POLICIES) + "\n")
	return len(cases), len(groups), sorted(unseparated)


if __name__ == "__main__":
	out = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))
	n_cases, n_groups, left = build(out)
	print("%d reads; the 24 policies fall into %d groups with identical output; %d pairs not separated by the search" % (n_cases, n_groups, len(left)))
