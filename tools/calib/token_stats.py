#!/usr/bin/env python3
"""What k_inflate_resolve_lanes meets in zlib level-6 streams of BAM records: the matches of a block 64 at a time, how many of them
read the output of an earlier one of the 64, how many rounds a group takes, how many matches a lane copies by itself (at most 16
bytes, no overlap with its own output or a run of one byte) and how many go across the lanes.  A plain-Python inflater that keeps
the matches; a few blocks take a few seconds.  usage: python tools/calib/token_stats.py [file.bam] [blocks]"""
import gzip, sys, zlib, collections

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
	def __init__(self, data): self.d = data; self.p = 0
	def take(self, n):
		v = 0
		for i in range(n):
			v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i; self.p += 1
		return v


def table(lens):
	count = collections.Counter(l for l in lens if l); code = 0; nxt = {}
	for l in range(1, 16):
		code = (code + count.get(l - 1, 0)) << 1 if l > 1 else 0
		nxt[l] = code
	out = {}
	for s, l in enumerate(lens):
		if l: out[(l, nxt[l])] = s; nxt[l] += 1
	return out


def sym(b, t):
	code = 0
	for l in range(1, 16):
		code = (code << 1) | b.take(1)
		if (l, code) in t: return t[(l, code)], l
	raise ValueError("bad code")


def tokens_of(stream):
	b = Bits(stream); pos = 0; toks = []; n_lit = 0; long_codes = 0; n_sym = 0
	while True:
		last = b.take(1); typ = b.take(2)
		if 0 == typ:
			b.p = (b.p + 7) & ~7; n = b.take(16); b.take(16); b.p += 8 * n; pos += n
		else:
			if 1 == typ:
				ll = table([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8); dd = table([5] * 30)
			else:
				hl = b.take(5) + 257; hd = b.take(5) + 1; hc = b.take(4) + 4
				cl = [0] * 19
				for i in range(hc): cl[ORDER[i]] = b.take(3)
				ct = table(cl); lens = []
				while len(lens) < hl + hd:
					s, _ = sym(b, ct)
					if s < 16: lens.append(s)
					elif 16 == s: lens += [lens[-1]] * (3 + b.take(2))
					elif 17 == s: lens += [0] * (3 + b.take(3))
					else: lens += [0] * (11 + b.take(7))
				ll = table(lens[:hl]); dd = table(lens[hl:])
			while True:
				s, l = sym(b, ll); n_sym += 1
				long_codes += l > 10
				if s < 256: pos += 1; n_lit += 1
				elif 256 == s: break
				else:
					length = LBASE[s - 257] + b.take(LEXT[s - 257])
					ds, l2 = sym(b, dd); long_codes += l2 > 9 and l <= 10
					dist = DBASE[ds] + b.take(DEXT[ds])
					toks.append((pos, length, dist)); pos += length
		if last: break
	return toks, n_lit, pos, long_codes, n_sym


def main():
	path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pvc3_bgzf_bench/in.bam"
	n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
	raw = gzip.open(path, "rb").read()
	tot = collections.Counter(); rounds_hist = collections.Counter(); len_hist = collections.Counter()
	for k in range(n_blocks):
		piece = raw[k * 65280:(k + 1) * 65280]
		c = zlib.compressobj(6, zlib.DEFLATED, -15); z = c.compress(piece) + c.flush()
		toks, n_lit, pos, long_codes, n_sym = tokens_of(z)
		assert pos == len(piece)
		tot["blocks"] += 1; tot["matches"] += len(toks); tot["literals"] += n_lit; tot["bytes"] += pos; tot["compressed"] += len(z); tot["long codes"] += long_codes; tot["symbols"] += n_sym
		for p, l, d in toks:
			len_hist[min(l, 40)] += 1
			tot["by lane (<= 16, no overlap or run of one)"] += l <= 16 and (d >= l or 1 == d)
			tot["<= 32 likewise"] += l <= 32 and (d >= l or 1 == d)
			tot["overlapping, period > 1"] += d < l and d > 1
		for g in range(0, len(toks), 64):
			grp = toks[g:g + 64]
			first = grp[0][0]
			needs = []
			for i, (p, l, d) in enumerate(grp):
				lo, hi = p - d, p - d + min(d, l)
				needs.append([j for j in range(i) if grp[j][0] < hi and grp[j][0] + grp[j][1] > lo])
			tot["groups"] += 1
			tot["groups without a near source"] += all(p - d + min(d, l) <= first for p, l, d in grp)
			tot["matches that read another of their 64"] += sum(1 for n in needs if n)
			done = set(); r = 0; across = 0
			while len(done) < len(grp):
				ready = [i for i in range(len(grp)) if i not in done and all(j in done for j in needs[i])]
				r += 1
				across += sum(1 for i in ready if not (grp[i][1] <= 16 and (grp[i][2] >= grp[i][1] or 1 == grp[i][2])))
				done.update(ready)
			rounds_hist[min(r, 20)] += 1; tot["rounds"] += r; tot["across the lanes"] += across
	for k, v in tot.items(): print("%-48s %d" % (k, v))
	print("rounds per group:", sorted(rounds_hist.items()))
	print("lengths:", sorted(len_hist.items()))


if __name__ == "__main__":
	main()
