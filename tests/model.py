"""Pure-Python column-walk model of rewrite_cigar (SURVEY.md §8 a2 "column-walk equivalence").

Independent of both the oracle (per-base select/rank, literal) and the HIP kernel (wave-parallel column walk);
used to cross-check that the two formulations agree.  Small inputs only.
"""

CONSUMES_REF = "M=XDN"


def column_walk(src_aln, dst_aln, dst_text, src_pos, cigar_items, query):
	"""src_aln/dst_aln: gapped strings; cigar_items: [(count, op)].  Returns (dst_pos, [(count, op)])."""
	out = []

	def push(op, n=1):
		if n == 0:
			return
		if out and out[-1][1] == op:
			out[-1][0] += n
		else:
			out.append([n, op])

	# column of the (src_pos)-th (0-based) non-gap character of src
	col = -1
	seen = -1
	while seen < src_pos:
		col += 1
		if src_aln[col] != "-":
			seen += 1
	dst_pos = sum(1 for c in dst_aln[:col] if c != "-")
	r = dst_pos	# unaligned dst index of the next dst non-gap column at or after col
	q = 0
	for n, op in cigar_items:
		if op in "IS":
			push(op, n); q += n
		elif op in "HP":
			push(op, n)
		elif op in CONSUMES_REF:
			m_type = op in "M=X"
			left = n
			while left:
				s_gap = src_aln[col] == "-"
				d_gap = dst_aln[col] == "-"
				if s_gap:
					if not d_gap:
						push("D"); r += 1
				else:
					if m_type:
						if d_gap:
							push("I")
						else:
							push("=" if query[q] == dst_text[r] else "X"); r += 1
						q += 1
					else:
						if not d_gap:
							push("D"); r += 1
					left -= 1
				col += 1
		else:
			raise ValueError(op)
	return dst_pos, [(n, op) for n, op in out]
