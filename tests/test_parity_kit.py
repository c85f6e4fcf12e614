"""tests/parity_kit/: the kit for pinning seqan3's score type and tie-break order against the real binary.
CPU: the committed expected files are what the oracle gives for the committed inputs, under all 24 policies, and the kit tells
22 groups of them apart.  GPU: the command line gives the same records under every policy."""
import os
import subprocess

import pytest

from helpers import ROOT

KIT = os.path.join(ROOT, "tests", "parity_kit")
POLICIES = [(s, t) for s in (0, 1) for t in range(12)]


def _name(p):
	return "score%d_tiebreak%02d" % p


def _load_kit():
	import bamio
	rows = {}
	name = None
	for line in open(os.path.join(KIT, "msa.a2m")):
		line = line.rstrip("\n")
		if line.startswith(">"):
			name = tuple(line[1:].split("\t"))
			rows[name] = ""
		else:
			rows[name] += line
	recs = []
	for line in open(os.path.join(KIT, "reads.sam")):
		if line.startswith("@"):
			continue
		f = line.rstrip("\n").split("\t")
		tags = []
		for t in f[11:]:
			tag, typ, val = t.split(":", 2)
			tags.append((tag, typ, int(val) if "i" == typ else val))
		recs.append(bamio.Record(f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[6], int(f[7]), int(f[8]), f[9], f[10], tags))
	return rows, recs


def test_expected_files_are_the_oracles(oracle_mod):
	from oracle import project_records as pr
	rows, recs = _load_kit()
	ref_aln, f_aln = rows[("chrK", "REF")], rows[("chrK", "F1")]
	ref_text = ref_aln.replace("-", "")
	assert open(os.path.join(KIT, "genome.fa")).read().split("\n", 1)[1].replace("\n", "") == ref_text
	o = oracle_mod.Oracle()
	e_ref, e_f = o.add_aligned(ref_aln), o.add_aligned(f_aln)
	o.set_reference(e_ref, ref_text)
	in_refs = [("chrK/REF", len(ref_text), ""), ("chrK/F1", len(f_aln.replace("-", "")), "")]
	out_refs, mapping, _ = pr.process_headers(in_refs, "/", "REF", None)
	opts = dict(preserve=set(), rname_tag="XR", pos_tag="XP", rnext_tag="YR", pnext_tag="YP", query_ranges_tag="XB", ref_ranges_tag="YB", rec_idx_tag="XI")
	outputs = {}
	for p in POLICIES:
		lines, _ = pr.project_records(recs, in_refs, o, [e_ref, e_f], [e_ref], mapping, out_refs, dict(opts, score_mode=p[0], tiebreak_mode=p[1]))
		assert "".join(l + "\n" for l in lines) == open(os.path.join(KIT, "expected", _name(p) + ".sam")).read(), p
		outputs[p] = tuple(lines)
		assert all("\tXB:B:I," in l for l in lines)		# every read of the kit is realigned
	assert len(set(outputs.values())) == 22
	groups = [l.split() for l in open(os.path.join(KIT, "policies.tsv")) if not l.startswith("#")]
	assert sorted(len(g) for g in groups) == [1] * 20 + [2, 2]
	for g in groups:
		assert len({outputs[(int(n[5]), int(n[-2:]))] for n in g}) == 1


@pytest.mark.gpu
def test_command_line_under_every_policy(tmp_path):
	from panvc3_amd import build
	tools = {os.path.basename(t): t for t in build.build_host()}
	subprocess.check_call([tools["panvc3_index_msa"], "--build-index", "--sequences=" + os.path.join(KIT, "msa.a2m"), "--msa-index-output=msa-index.dat"], cwd=tmp_path)
	for s, t in POLICIES:
		p = subprocess.run([tools["panvc3_project_alignments"], "--alignments=" + os.path.join(KIT, "reads.sam"), "--msa-index=msa-index.dat", "--reference=" + os.path.join(KIT, "genome.fa"),
							"--reference-msa-id=REF", "--ref-id-separator=/", "--record-index-tag=XI", "--realign-score-type=" + ("int32" if s else "double"), "--realign-tiebreak=%d" % t],
						   cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
		assert 0 == p.returncode, p.stderr.decode()[-1000:]
		body = "".join(l + "\n" for l in p.stdout.decode().split("\n") if l and not l.startswith("@"))
		assert body == open(os.path.join(KIT, "expected", _name((s, t)) + ".sam")).read(), (s, t)
