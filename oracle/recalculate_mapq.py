"""CPU restatement of the reference's recalculate_mapq (recalculate-mapq/main.cc), the step after project_alignments in both
workflows (test-workflow/Snakefile:76-86).  TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker of the HIP kernels
(panvc3_amd/csrc/mapq.hip) and of bin/panvc3_recalculate_mapq.  Nothing under panvc3_amd/ imports this module.

PARITY UNPINNED: the reference has no test, fixture or checked-in output for this tool, and it cannot be built here (seqan3,
libbio, range-v3 absent).  The functions below follow the source line by line; what is pinned are hand-derived values of
the Bowtie 2 (version 2) MAPQ table (tests/test_mapq_oracle.py).

  calculate_mapq            bowtie2_v2_score_calculator::calculate_mapq          main.cc:441-566 (tables :375-431)
  cigar_score               cigar_alignment_scorer::calculate_score              main.cc:189-289 (penalties :182-186)
  process_alignment_group   mapq_scorer::process_alignment_group                 main.cc:962-1180 (add_paired_segment_score :893-917)

Records are dicts with the fields the reference reads: ref_id / pos (projected RNAME id and 0-based POS, None if unset),
mate_ref_id / mate_pos (RNEXT / PNEXT), orig (XR, XP) and orig_mate (YR, YP) tag pairs or None, score (the AS value or the
rescored value; None = no AS tag), seq (string), and they receive "mapq".
"""
import sys

ALIGNMENT_SCORE_MIN = -sys.float_info.max		# main.cc:50
MAPQ_NO_NEXT_RECORD = 255						# main.cc:51
INVALID_POSITION = (0xFFFFFFFF, 0xFFFFFFFF)		# main.cc:635

UNIQUE_ALIGNMENT_SCORES = [(0.0, 0), (0.3, 3), (0.4, 8), (0.5, 23), (0.6, 24), (0.7, 40), (0.8, 42)]		# main.cc:375-383
NON_UNIQUE_ALIGNMENT_SCORES = [																			# main.cc:393-425
	(0.0, 0.0, 2), (0.0, 0.67, 6),
	(0.1, 0.0, 0), (0.1, 0.67, 7), (0.1, 0.88, 12), (0.1, 1.0, 30),
	(0.2, 0.0, 0), (0.2, 0.67, 11), (0.2, 0.88, 17), (0.2, 1.0, 31),
	(0.3, 0.0, 3), (0.3, 0.67, 15), (0.3, 0.88, 18), (0.3, 1.0, 32),
	(0.4, 0.0, 4), (0.4, 0.68, 14), (0.4, 0.84, 21), (0.4, 1.0, 34),
	(0.5, 0.0, 5), (0.5, 0.68, 16), (0.5, 0.84, 25), (0.5, 1.0, 35),
	(0.6, 0.0, 22), (0.6, 1.0, 36),
	(0.7, 0.0, 26), (0.7, 1.0, 37),
	(0.8, 0.0, 27), (0.8, 1.0, 38),
	(0.9, 0.0, 33), (0.9, 1.0, 39),
]


def read_min_score(read_length):
	"""main.cc:455-460: Bowtie 2's --score-min L,-0.6,-0.6."""
	if not read_length:
		return 0.0
	return -0.6 + (-0.6 * read_length)


def _upper_bound(keys, value):
	"""std::upper_bound: index of the first key greater than value."""
	lo, hi = 0, len(keys)
	while lo < hi:
		mid = (lo + hi) // 2
		if value < keys[mid]:
			hi = mid
		else:
			lo = mid + 1
	return lo


def calculate_mapq(read_length, other_read_length, score, next_score_):
	"""main.cc:479-566.  Note :494: a missing next alignment (ALIGNMENT_SCORE_MIN) is first replaced by the minimum valid
	score, so the `ALIGNMENT_SCORE_MIN == next_score` branch (:508) is not reachable and a unique alignment is rated through
	the second table with diff_next = score - min_score, as the source does."""
	min_score = read_min_score(read_length) + read_min_score(other_read_length)
	max_score = 0.0
	score_range = max(1.0, max_score - min_score)
	if score < min_score:
		return 0
	next_score = next_score_ if min_score < next_score_ else min_score
	normalised_score_quotient = (score - min_score) / score_range
	diff_next = score - next_score
	diff_next_quotient = diff_next / score_range
	if ALIGNMENT_SCORE_MIN == next_score:
		i = _upper_bound([e[0] for e in UNIQUE_ALIGNMENT_SCORES], normalised_score_quotient)
		return UNIQUE_ALIGNMENT_SCORES[i - 1][1]
	elif diff_next:
		i = _upper_bound([e[0] for e in NON_UNIQUE_ALIGNMENT_SCORES], diff_next_quotient)
		threshold = NON_UNIQUE_ALIGNMENT_SCORES[i - 1][0]
		j = _upper_bound([(e[0], e[1]) for e in NON_UNIQUE_ALIGNMENT_SCORES[:i]], (threshold, normalised_score_quotient))
		return NON_UNIQUE_ALIGNMENT_SCORES[j - 1][2]
	else:
		return 1 if normalised_score_quotient >= 0.67 else 0


def fill_penalties(lo, hi):
	"""main.cc:182-186: 256 entries, cut-off at quality 40; the double is converted to int32 (truncation)."""
	return [int(min(i, 40) / 40.0 * (hi - lo) + lo) for i in range(256)]


def cigar_score(cigar, seq, qual, ref_n_positions, min_mismatch=2, max_mismatch=6, n_penalty=1, gap_open=5, gap_ext=3):
	"""main.cc:189-289.  cigar: [(count, op char)]; seq: string over ACGTN; qual: phred ranks; ref_n_positions: bytes of the XN
	bit map (bit rp % 8 of byte rp / 8 set iff the reference has N at alignment-relative position rp).  Raises ValueError for
	an M operation (the reference fails) and IndexError where the reference's assertion on the bit map's size fires."""
	mm = fill_penalties(min_mismatch, max_mismatch)
	nn = fill_penalties(n_penalty, n_penalty)
	query_pos = ref_pos = 0
	prev_op = None
	retval = 0.0
	for count, op in cigar:
		if op in "HP":
			pass
		elif op == "N":
			ref_pos += count
		elif op in "ID":
			if prev_op not in ("I", "D"):
				retval -= gap_open
			retval -= count * gap_ext
			if op == "I":
				query_pos += count
			else:
				ref_pos += count
			prev_op = op
		elif op == "S":
			query_pos += count
			prev_op = op
		elif op == "=":
			for i in range(count):
				if seq[query_pos + i] == "N":
					retval -= nn[qual[query_pos + i]]
			query_pos += count
			ref_pos += count
			prev_op = op
		elif op == "X":
			for i in range(count):
				qp, rp = query_pos + i, ref_pos + i
				if rp // 8 >= len(ref_n_positions):
					raise IndexError("XN bit map too short")
				if seq[qp] == "N" or ((1 << (rp % 8)) & ref_n_positions[rp // 8]):
					retval -= nn[qual[qp]]
				else:
					retval -= mm[qual[qp]]
			query_pos += count
			ref_pos += count
			prev_op = op
		else:
			raise ValueError("unexpected CIGAR operation %r" % op)
	return retval


def _position(ref_id, pos):
	return INVALID_POSITION if ref_id is None or pos is None else (ref_id, pos)


def process_alignment_group(records):
	"""main.cc:962-1180.  Sets rec["mapq"] for every record of the group and returns (emitted, statistics delta); a group with
	both paired and unpaired records is skipped (:1039-1044: nothing of it is written).

	One deliberate reading: at :1151 the source compares `it->sequence` (the entry at the upper bound, possibly end()) with
	`other.sequence`; for groups of paired records every entry's sequence is null and the test always passes; for unpaired
	groups the test is read as "the other entry has the sequence of the current record" (the comment's intent)."""
	stats = dict(total_alignments=len(records), unpaired_alignments=0, reads_with_and_without_mate=0, mate_not_found=0, reads_without_valid_position=0)
	if not records:
		return False, stats
	seen = 0
	segs = []		# (original position, score, length)
	scored = []
	for rec in records:
		has_mate = rec.get("mate_ref_id") is not None and rec.get("mate_pos") is not None
		seen |= 1 << int(has_mate)
		stats["unpaired_alignments"] += int(not has_mate)
		score = ALIGNMENT_SCORE_MIN if rec.get("score") is None else float(rec["score"])
		orig = rec.get("orig")
		segs.append((INVALID_POSITION if orig is None else tuple(orig), score, len(rec["seq"])))
		scored.append(dict(rec=rec, score=score, pairwise=ALIGNMENT_SCORE_MIN, mate_length=0))
	if seen == 3:
		stats["reads_with_and_without_mate"] += 1
		return False, stats
	segs.append((INVALID_POSITION, 0.0, 0))
	segs.sort(key=lambda s: (s[0], s[1]))
	if not any(s[0] != INVALID_POSITION for s in segs):
		stats["reads_without_valid_position"] += 1
	seg_pos = [s[0] for s in segs]

	entries = []	# sorted by positions: dict(positions, seq_owner, score, other, has_mate)
	for i, sr in enumerate(scored):
		rec = sr["rec"]
		a, b = _position(rec.get("ref_id"), rec.get("pos")), _position(rec.get("mate_ref_id"), rec.get("mate_pos"))
		positions = (a, b) if a < b else (b, a)			# position_pair::normalise (:735)
		has_mate = positions[1] != INVALID_POSITION
		pss = dict(positions=positions, seq_owner=None if has_mate else i, score=sr["score"], other=0.0, has_mate=False)
		mate_length = 0
		if positions[0] != INVALID_POSITION and has_mate:
			om = rec.get("orig_mate")
			mate_original = INVALID_POSITION if om is None else tuple(om)
			k = _upper_bound(seg_pos, mate_original)
			if k == 0 or seg_pos[k - 1] != mate_original:
				stats["mate_not_found"] += 1
			else:
				pss["other"] = segs[k - 1][1]		# the best score at the mate's position
				pss["has_mate"] = True
				mate_length = segs[k - 1][2]
		sr["pairwise"] = pss["score"] + pss["other"]
		sr["mate_length"] = mate_length
		# add_paired_segment_score (:893-917)
		j = 0
		while j < len(entries) and entries[j]["positions"] < positions:
			j += 1
		done = False
		while j < len(entries) and entries[j]["positions"] == positions:
			if entries[j]["seq_owner"] == pss["seq_owner"]:
				if entries[j]["score"] + entries[j]["other"] < pss["score"] + pss["other"]:
					entries[j] = pss
				done = True
				break
			j += 1
		if not done:
			entries.insert(j, pss)
	entries.sort(key=lambda e: e["score"] + e["other"])
	totals = [e["score"] + e["other"] for e in entries]

	for sr in scored:
		rec = sr["rec"]
		seq = rec["seq"]
		a, b = _position(rec.get("ref_id"), rec.get("pos")), _position(rec.get("mate_ref_id"), rec.get("mate_pos"))
		pos = (a, b) if a < b else (b, a)
		has_mate = pos[1] != INVALID_POSITION
		it = _upper_bound(totals, sr["pairwise"])
		is_best = (it == len(entries))
		mapq = None
		for other in reversed(entries[:it]):
			if other["seq_owner"] is not None and scored[other["seq_owner"]]["rec"]["seq"] != seq:
				continue
			if other["positions"] == pos:
				continue
			o_total = other["score"] + other["other"]
			o_max = max(other["score"], other["other"]) if other["has_mate"] else other["score"]
			mapq = calculate_mapq(len(seq), sr["mate_length"], sr["pairwise"], o_total if has_mate else o_max)
			break
		if mapq is None:
			mapq = calculate_mapq(len(seq), sr["mate_length"], sr["pairwise"], ALIGNMENT_SCORE_MIN) if is_best else MAPQ_NO_NEXT_RECORD
		rec["mapq"] = mapq
	return True, stats
