"""CPU restatement of the reference's per-record rewriting around project_alignment (project-alignments/project_alignments.cc).

TEST INFRASTRUCTURE ONLY (checker for bin/panvc3_project_alignments).  Works on SAM-level Python values (tests/bamio.Record)
and uses the C oracle for the projection itself.  PARITY UNPINNED against the real binary: no reference test covers SAM/BAM
I/O, header or tag rewriting, and no reference output is checked in; the behaviours below are restated
from the source lines cited.
"""
from oracle.oracle import Oracle, cigar_decode  # noqa: F401

PRESERVED = {"AM", "AS", "BC", "BZ", "CB", "CO", "CR", "CS", "CT", "CY", "E2", "FZ", "LB", "MI", "ML", "MM", "OA", "OQ", "OX", "PG", "PU", "QT", "QX", "RX", "TS"}	# :53-87


def process_headers(in_refs, separator, ref_seq_id, order_ids):
	"""in_refs: [(name, length, extra)] -> (out_refs [(chr, length, extra)], mapping input idx -> output idx)   (:1092-1253)"""
	chr_seq = []
	for name, _, _ in in_refs:
		pos = min((name.find(c) for c in separator if c in name), default=-1)		# find_first_of (:1124)
		assert pos >= 0, name
		chr_seq.append((name[:pos], name[pos + 1:]))
	order = {}
	if order_ids is not None:
		present = {c for c, _ in chr_seq}
		for ident in order_ids:
			if ident in present and ident not in order:
				order[ident] = len(order)
	unique, mapping, ref_idxs = {}, [], []
	nxt = len(order)
	for i, (c, s) in enumerate(chr_seq):
		if c not in unique:
			if c in order:
				unique[c] = order[c]
			else:
				unique[c] = nxt; nxt += 1
		mapping.append(unique[c])
		if s == ref_seq_id:
			ref_idxs.append(i)
	out = [[None, 0, ""] for _ in unique]
	for c, idx in unique.items():
		out[idx][0] = c
	for i in ref_idxs:
		out[mapping[i]][1] = in_refs[i][1]; out[mapping[i]][2] = in_refs[i][2]
	return [tuple(x) for x in out], mapping, chr_seq


def keep_record(rec):
	"""The filters of input_processor::process_input (:528-570)."""
	if rec.flag & (0x4 | 0x200 | 0x400):
		return False
	if rec.rname == "*" or rec.pos < 1 or rec.seq == "*":
		return False
	return True


def project_records(records, in_refs, oracle, src_entries, dst_entries, mapping, out_refs, opts):
	"""records: bamio.Record list (all input records, in order).  Returns expected SAM lines of the output records.

	opts: dict(preserve=set, rname_tag, pos_tag, rnext_tag, pnext_tag, query_ranges_tag, ref_ranges_tag, rec_idx_tag, gap_open, gap_ext,
	score_mode, tiebreak_mode)."""
	ref_index = {n: i for i, (n, _, _) in enumerate(in_refs)}
	valid = [(i, r) for i, r in enumerate(records) if keep_record(r)]
	lines, all_ranges = [], []
	for k, (idx, r) in enumerate(valid):
		ref_id = ref_index[r.rname]
		src_pos = r.pos - 1
		# seqan3 reads SEQ as dna5 (non-ACGT -> N) and QUAL as phred42 (saturating); the projection sees those values
		seq = "".join(c if c in "ACGT" else "N" for c in r.seq.upper())
		qual = None if r.qual == "*" else [min(41, ord(c) - 33) for c in r.qual]
		dst_pos, cig, runs = oracle.project_alignment(src_entries[ref_id], dst_entries[mapping[ref_id]], src_pos, r.cigar if r.cigar != "*" else "", seq, qual=qual,
													   gap_open=opts.get("gap_open", 0), gap_ext=opts.get("gap_ext", -1), score_mode=opts.get("score_mode", 0), tiebreak_mode=opts.get("tiebreak_mode", 0))
		tags = {}
		nm = None
		old_oa = ""
		for tag, typ, val in r.tags:
			if tag == "NM" and typ == "i":
				nm = val
			if tag not in PRESERVED and tag not in opts.get("preserve", set()):		# :775-802
				continue
			if tag == "OA" and typ == "Z":
				old_oa = val
				continue
			tags[tag] = (typ, val)
		strand = "-" if r.flag & 0x10 else "+"
		oa = "%s%s,%d,%s,%s,%d,%s;" % (old_oa, r.rname, src_pos, strand, "" if r.cigar == "*" else r.cigar, r.mapq, "" if nm is None else str(nm))	# :805-844
		tags["OA"] = ("Z", oa)
		if len(runs):
			if opts.get("query_ranges_tag"):
				tags[opts["query_ranges_tag"]] = ("B", ("I", [x for rr in runs for x in (int(rr["query_location"]), int(rr["query_location"]) + int(rr["query_length"]))]))
			if opts.get("ref_ranges_tag"):
				tags[opts["ref_ranges_tag"]] = ("B", ("I", [x for rr in runs for x in (int(rr["ref_position"]), int(rr["ref_position"]) + int(rr["ref_length"]))]))
			all_ranges.extend((int(rr["ref_position"]), int(rr["ref_length"])) for rr in runs)
		if opts.get("rec_idx_tag"):			# :864-869 with CHUNK_SIZE 4 (:51)
			chunk_begin = k - k % 4
			chunk = valid[chunk_begin:chunk_begin + 4]
			xi = chunk[-1][0] - len(chunk) + 1
			if xi <= 2**31 - 1:
				tags[opts["rec_idx_tag"]] = ("i", xi)
		if opts.get("rname_tag"): tags[opts["rname_tag"]] = ("i", ref_id)
		if opts.get("pos_tag"): tags[opts["pos_tag"]] = ("i", src_pos)
		rnext, pnext = "*", 0
		if r.rnext != "*":					# :880-911
			mate_ref = ref_id if r.rnext == "=" else ref_index[r.rnext]
			out_mate = mapping[mate_ref]
			rnext = out_refs[out_mate][0]
			if opts.get("rnext_tag"): tags[opts["rnext_tag"]] = ("i", mate_ref)
			if r.pnext >= 1:
				pnext = oracle.project_position(src_entries[mate_ref], dst_entries[out_mate], r.pnext - 1) + 1
				if opts.get("pnext_tag"): tags[opts["pnext_tag"]] = ("i", r.pnext - 1)
		rname = out_refs[mapping[ref_id]][0]
		if rnext == rname:
			rnext = "="
		out_qual = "*" if r.qual == "*" else "".join(chr(33 + q) for q in qual)
		f = [r.qname, str(r.flag), rname, str(dst_pos + 1), str(r.mapq), cigar_decode(cig) or "*", rnext, str(pnext), str(r.tlen), seq, out_qual]
		for tag in sorted(tags):			# seqan3's tag dictionary is an ordered map keyed by the tag (sam_tag.hh:34-44)
			typ, val = tags[tag]
			if typ == "B":
				f.append("%s:B:%s,%s" % (tag, val[0], ",".join(str(x) for x in val[1])))
			elif typ == "f":
				f.append("%s:f:%g" % (tag, float(val)))
			else:
				f.append("%s:%s:%s" % (tag, typ, val))
		lines.append("\t".join(f))
	return lines, sorted(set(all_ranges))
