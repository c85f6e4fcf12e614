"""CPU restatement of the reference's recalculate_mapq TOOL (recalculate-mapq/main.cc:291-335, 1186-1278) on top of
oracle/recalculate_mapq.py, for the test of bin/panvc3_recalculate_mapq.  Test infrastructure only; parity unpinned (the
reference holds no fixture for this tool).

	process(records, ref_names, rescore=False, ...) -> records to write, in order, with their new MAPQ (and, when rescoring,
	their new AS / XS / YS tags); `records` are tests/bamio.Record objects.
"""
from . import recalculate_mapq as rm


def _dna5(seq):
	return "".join(c if c in "ACGT" else "N" for c in seq.upper())


def _tag(rec, name):
	for t, ty, v in rec.tags:
		if t == name:
			return ty, v
	return None


def _cigar(text):
	out, num = [], ""
	for ch in text:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num), ch)); num = ""
	return out


def process(records, ref_names, rescore=False, scoring=(2, 6, 1, 5, 3), tags=("XN", "XR", "XP", "YR", "YP", "XS", "YS")):
	t_xn, t_xr, t_xp, t_yr, t_yp, t_xs, t_ys = tags
	ref_index = {n: i for i, n in enumerate(ref_names)}
	out = []
	stats = dict(total_alignments=0, unpaired_alignments=0, reads_with_and_without_mate=0, mate_not_found=0, reads_without_valid_position=0)
	groups = []
	for r in records:
		if r.flag & 0x4:				# main.cc:1236-1238
			continue
		if groups and groups[-1][0].qname == r.qname:
			groups[-1].append(r)
		else:
			groups.append([r])
	for g in groups:
		orc = []
		for r in g:
			ref_id = None if r.rname == "*" else ref_index[r.rname]
			pos = None if r.pos <= 0 else r.pos - 1
			mate_ref = None if r.rnext == "*" else (ref_id if r.rnext == "=" else ref_index[r.rnext])
			mate_pos = None if r.pnext <= 0 else r.pnext - 1
			xr, xp, yr, yp = (_tag(r, t) for t in (t_xr, t_xp, t_yr, t_yp))
			a = _tag(r, "AS")
			score = None if a is None else int(a[1])
			if rescore:
				xn = _tag(r, t_xn)
				bits = bytes(xn[1][1]) if xn else b""
				qual = [min(ord(c) - 33, 41) for c in r.qual] if r.qual != "*" else [0] * len(r.seq)
				score = rm.cigar_score(_cigar(r.cigar), _dna5(r.seq), qual, bits, *scoring)
			orc.append(dict(ref_id=ref_id, pos=pos, mate_ref_id=mate_ref, mate_pos=mate_pos,
							orig=None if xr is None or xp is None else (int(xr[1]), int(xp[1])), orig_mate=None if yr is None or yp is None else (int(yr[1]), int(yp[1])),
							score=score, seq=_dna5(r.seq), had_as=a))
		emitted, st = rm.process_alignment_group(orc)
		for k in stats:
			stats[k] += st[k]
		if not emitted:
			continue
		for r, o in zip(g, orc):
			new_tags = list(r.tags)
			if rescore:			# main.cc:316-333
				new_tags = [t for t in new_tags if t[0] not in ("AS",) + ((t_xs, t_ys) if o["had_as"] else ())]
				if o["had_as"]:
					new_tags.append((t_xs, "i", int(o["had_as"][1])))
					new_tags.append((t_ys, "f", "%g" % float(o["score"])))
				new_tags.append(("AS", "i", int(o["score"])))
				new_tags.sort(key=lambda t: (ord(t[0][0]), ord(t[0][1])))
			out.append((r, o["mapq"], new_tags))
	return out, stats
