"""Minimal SAM / BAM writer and reader for the tests (SAMv1 §4.2; zlib only).  Test infrastructure."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
SEQ4 = "=ACMGRSVTWYHKDBN"


def reg2bin(beg, end):
	end -= 1
	if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
	if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
	if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
	if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
	if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
	return 0


def parse_cigar(text):
	out, num = [], ""
	if text == "*":
		return out
	for ch in text:
		if ch.isdigit():
			num += ch
		else:
			out.append((int(num), ch)); num = ""
	return out


class Record:
	"""A SAM record as Python values; tags = list of (tag, type, value) with type in A i f Z H B (B value = (subtype, [..]))."""

	def __init__(self, qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual, tags=()):
		self.qname, self.flag, self.rname, self.pos, self.mapq, self.cigar = qname, flag, rname, pos, mapq, cigar
		self.rnext, self.pnext, self.tlen, self.seq, self.qual, self.tags = rnext, pnext, tlen, seq, qual, list(tags)

	def sam_line(self):
		f = [self.qname, str(self.flag), self.rname, str(self.pos), str(self.mapq), self.cigar, self.rnext, str(self.pnext), str(self.tlen), self.seq, self.qual]
		for tag, typ, val in self.tags:
			if typ == "B":
				f.append("%s:B:%s,%s" % (tag, val[0], ",".join(str(x) for x in val[1])))
			else:
				f.append("%s:%s:%s" % (tag, typ, val))
		return "\t".join(f)


def _aux(tag, typ, val):
	b = tag.encode()
	if typ == "A":
		return b + b"A" + val.encode()
	if typ == "i":
		v = int(val)
		for code, fmt, lo, hi in (("C", "<B", 0, 255), ("c", "<b", -128, 127), ("S", "<H", 0, 65535), ("s", "<h", -32768, 32767), ("I", "<I", 0, 2**32 - 1), ("i", "<i", -2**31, 2**31 - 1)):
			if lo <= v <= hi and ((v >= 0) == (code.isupper())):
				return b + code.encode() + struct.pack(fmt, v)
		raise ValueError(v)
	if typ == "f":
		return b + b"f" + struct.pack("<f", float(val))
	if typ in "ZH":
		return b + typ.encode() + val.encode() + b"\0"
	if typ == "B":
		st, items = val
		fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
		return b + b"B" + st.encode() + struct.pack("<I", len(items)) + struct.pack("<%d%s" % (len(items), fmt), *items)
	raise ValueError(typ)


def bam_bytes(rec, ref_index):
	ref_id = -1 if rec.rname == "*" else ref_index[rec.rname]
	next_ref = -1 if rec.rnext == "*" else (ref_id if rec.rnext == "=" else ref_index[rec.rnext])
	cig = parse_cigar(rec.cigar)
	seq = "" if rec.seq == "*" else rec.seq
	l = len(seq)
	span = sum(n for n, op in cig if op in "MDN=X")
	pos = rec.pos - 1
	name = rec.qname.encode() + b"\0"
	body = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name), rec.mapq, reg2bin(pos, pos + (span or 1)), len(cig), rec.flag, l, next_ref, rec.pnext - 1, rec.tlen)
	body += name
	body += b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for n, op in cig)
	codes = [SEQ4.index(c.upper()) if c.upper() in SEQ4 else 15 for c in seq] + [0]
	body += bytes((codes[i] << 4) | codes[i + 1] for i in range(0, l, 2))
	body += (b"\xff" * l) if rec.qual == "*" else bytes(ord(c) - 33 for c in rec.qual)
	for tag, typ, val in rec.tags:
		body += _aux(tag, typ, val)
	return struct.pack("<i", len(body)) + body


def _bgzf_block(data):
	co = zlib.compressobj(6, zlib.DEFLATED, -15)
	comp = co.compress(data) + co.flush()
	bsize = 18 + len(comp) + 8
	return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bam(path, header_text, refs, records, aligned=False, block=0xff00):
	"""refs: [(name, length)]; records: Record objects.  aligned: no record straddles two BGZF blocks unless it is larger than
	one (what htslib's bam_write1 does); otherwise the stream is cut every `block` bytes wherever that falls."""
	ref_index = {n: i for i, (n, l) in enumerate(refs)}
	raw = b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs))
	for n, l in refs:
		raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
	with open(path, "wb") as f:
		if aligned:
			cur = raw
			for r in records:
				b = bam_bytes(r, ref_index)
				if cur and len(cur) + len(b) > block:
					f.write(_bgzf_block(cur)); cur = b""
				cur += b
				while len(cur) > block:
					f.write(_bgzf_block(cur[:block])); cur = cur[block:]
			if cur:
				f.write(_bgzf_block(cur))
		else:
			for r in records:
				raw += bam_bytes(r, ref_index)
			for i in range(0, len(raw), block):
				f.write(_bgzf_block(raw[i:i + block]))
		f.write(_bgzf_block(b""))


def write_sam(path, header_text, records):
	with open(path, "w") as f:
		f.write(header_text)
		for r in records:
			f.write(r.sam_line() + "\n")


def read_bam(path):
	"""-> (header_text, refs, [sam-like dict per record])"""
	import gzip
	raw = gzip.open(path, "rb").read()
	assert raw[:4] == b"BAM\1"
	l_text, = struct.unpack_from("<i", raw, 4)
	text = raw[8:8 + l_text].decode().rstrip("\0")
	p = 8 + l_text
	n_ref, = struct.unpack_from("<i", raw, p); p += 4
	refs = []
	for _ in range(n_ref):
		l_name, = struct.unpack_from("<i", raw, p); p += 4
		name = raw[p:p + l_name - 1].decode(); p += l_name
		l_ref, = struct.unpack_from("<i", raw, p); p += 4
		refs.append((name, l_ref))
	recs = []
	while p < len(raw):
		bs, = struct.unpack_from("<i", raw, p); p += 4
		body = raw[p:p + bs]; p += bs
		ref_id, pos, l_name, mapq, bin_, n_cig, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHIiii", body, 0)
		o = 32
		qname = body[o:o + l_name - 1].decode(); o += l_name
		cig = "".join("%d%s" % (c >> 4, CIGAR_OPS[c & 15]) for c in struct.unpack_from("<%dI" % n_cig, body, o)); o += 4 * n_cig
		sb = body[o:o + (l_seq + 1) // 2]; o += (l_seq + 1) // 2
		seq = "".join(SEQ4[(sb[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
		qb = body[o:o + l_seq]; o += l_seq
		qual = "*" if (l_seq == 0 or qb[0] == 0xff) else "".join(chr(33 + x) for x in qb)
		tags = []
		while o < len(body):
			tag = body[o:o + 2].decode(); typ = chr(body[o + 2]); o += 3
			if typ in "cCsSiI":
				fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
				v, = struct.unpack_from(fmt, body, o); o += struct.calcsize(fmt); tags.append((tag, "i", v, typ))
			elif typ == "A":
				tags.append((tag, "A", chr(body[o]), typ)); o += 1
			elif typ == "f":
				v, = struct.unpack_from("<f", body, o); o += 4; tags.append((tag, "f", v, typ))
			elif typ in "ZH":
				e = body.index(b"\0", o); tags.append((tag, typ, body[o:e].decode(), typ)); o = e + 1
			elif typ == "B":
				st = chr(body[o]); n, = struct.unpack_from("<I", body, o + 1); o += 5
				fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
				vals = list(struct.unpack_from("<%d%s" % (n, fmt), body, o)); o += n * struct.calcsize(fmt)
				tags.append((tag, "B", (st, vals), typ))
		recs.append(dict(qname=qname, flag=flag, ref_id=ref_id, pos=pos, mapq=mapq, cigar=cig or "*", next_ref=next_ref, next_pos=next_pos, tlen=tlen,
						 seq=seq or "*", qual=qual, tags=tags, bin=bin_))
	return text, refs, recs
