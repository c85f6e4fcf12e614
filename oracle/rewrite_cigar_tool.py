"""CPU restatement of the reference's rewrite_cigar TOOL (rewrite-cigar/main.cc) for the tests of bin/panvc3_rewrite_cigar.
Test infrastructure only.  Parity unpinned: the reference holds no fixture or test for this tool; the restatement follows its
source line by line.

	alignment_match(cigar)                      :151-188  = and X become M, cigar_buffer merges equal neighbours
	sequence_match(cigar, seq, ref, pos)        :191-284  M split into = / X (runs inside one M only), N bit map :126-148
"""

OPS = "MIDNSHP=X"


def parse(cigar):
	out, num = [], ""
	for ch in cigar:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num), ch)); num = ""
	return out


def dna5_char(c):
	"""seqan3::dna5: assign_char then to_char."""
	c = c.upper()
	return c if c in "ACGT" else "N"


def alignment_match(cigar):
	out = []
	for n, op in parse(cigar):
		assert op in OPS
		op = "M" if op in "=X" else op
		if n == 0:
			continue							# cigar_buffer drops nothing explicitly, but a zero count adds nothing (cigar.cc:61-87)
		if out and out[-1][1] == op:
			out[-1] = (out[-1][0] + n, op)
		else:
			out.append((n, op))
	return "".join("%d%s" % x for x in out)


def sequence_match(cigar, seq, ref, pos):
	"""-> (cigar, bytes of the N bit map: empty iff no operation consumes the reference)."""
	out, nbits = [], bytearray()
	q = r = 0

	def fill(count):
		need = (r + count + 7) // 8
		while len(nbits) < need:
			nbits.append(0)
		for i in range(count):
			if ref[pos + r + i] == "N":
				nbits[(r + i) // 8] |= 1 << ((r + i) % 8)

	for n, op in parse(cigar):
		if op in "HP":
			out.append((n, op))
		elif op in "IS":
			out.append((n, op)); q += n
		elif op in "=X":
			out.append((n, op)); fill(n); q += n; r += n
		elif op in "DN":
			out.append((n, op)); fill(n); r += n
		elif op == "M":
			fill(n)
			prev_n, prev = 1, "=" if dna5_char(seq[q]) == ref[pos + r] else "X"
			for i in range(1, n):
				cur = "=" if dna5_char(seq[q + i]) == ref[pos + r + i] else "X"
				if cur == prev:
					prev_n += 1
				else:
					out.append((prev_n, prev)); prev_n, prev = 1, cur
			out.append((prev_n, prev))
			q += n; r += n
		else:
			raise ValueError(op)
	return "".join("%d%s" % x for x in out), bytes(nbits)
