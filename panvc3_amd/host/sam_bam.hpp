// sam_bam.hpp — SAM / BAM headers and records for the host pipeline.
//
// Records are kept in BAM's binary layout (SAMv1 §4.2) whatever the input format: the 4-bit SEQ, the u32 CIGAR and the
// QUAL bytes are then exactly what the C ABI batch (include/pvc3.h) takes.  SAM text is parsed into / formatted from it.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "bgzf.hpp"

namespace pvc3host {

struct sq_line
{
	std::string name;
	int64_t length{};
	std::string extra;		// remaining TAG:VALUE fields, tab separated
};

struct pg_line
{
	std::string id, name, command_line, previous, description, version;
	std::string other;		// fields this tool does not know, tab separated
};

struct sam_header
{
	std::string hd_version{"1.6"};
	std::vector <sq_line> refs;
	std::vector <std::pair <std::string, std::string>> read_groups;	// (ID, rest of the line)
	std::vector <pg_line> programs;
	std::vector <std::string> comments;
	std::unordered_map <std::string, int32_t> ref_index;

	void index_refs()
	{
		ref_index.clear();
		for (std::size_t i = 0; i < refs.size(); ++i) ref_index.emplace(refs[i].name, (int32_t) i);
	}

	static std::vector <std::string_view> split_tabs(std::string_view s)
	{
		std::vector <std::string_view> out;
		std::size_t b = 0;
		for (;;)
		{
			std::size_t const e = s.find('\t', b);
			out.push_back(s.substr(b, (std::string_view::npos == e) ? e : e - b));
			if (std::string_view::npos == e) break;
			b = e + 1;
		}
		return out;
	}

	void parse_line(std::string const &line)
	{
		if (line.size() < 3 || '@' != line[0]) return;
		std::string_view const kind(line.data(), 3);
		auto const fields = split_tabs(std::string_view(line).substr(line.size() > 3 ? 4 : 3));
		if ("@HD" == kind)
		{
			for (auto f : fields) if (f.substr(0, 3) == "VN:") hd_version = std::string(f.substr(3));
		}
		else if ("@SQ" == kind)
		{
			sq_line sq;
			for (auto f : fields)
			{
				if (f.substr(0, 3) == "SN:") sq.name = std::string(f.substr(3));
				else if (f.substr(0, 3) == "LN:") sq.length = std::atoll(std::string(f.substr(3)).c_str());
				else { if (!sq.extra.empty()) sq.extra += '\t'; sq.extra += std::string(f); }
			}
			refs.push_back(std::move(sq));
		}
		else if ("@RG" == kind)
		{
			std::string id, rest;
			for (auto f : fields)
			{
				if (f.substr(0, 3) == "ID:") id = std::string(f.substr(3));
				else { if (!rest.empty()) rest += '\t'; rest += std::string(f); }
			}
			read_groups.emplace_back(std::move(id), std::move(rest));
		}
		else if ("@PG" == kind)
		{
			pg_line pg;
			for (auto f : fields)
			{
				auto const key = f.substr(0, 3);
				std::string const val(f.size() > 3 ? f.substr(3) : std::string_view());
				if ("ID:" == key) pg.id = val;
				else if ("PN:" == key) pg.name = val;
				else if ("CL:" == key) pg.command_line = val;
				else if ("PP:" == key) pg.previous = val;
				else if ("DS:" == key) pg.description = val;
				else if ("VN:" == key) pg.version = val;
				else { if (!pg.other.empty()) pg.other += '\t'; pg.other += std::string(f); }
			}
			programs.push_back(std::move(pg));
		}
		else if ("@CO" == kind)
			comments.emplace_back(line.size() > 4 ? line.substr(4) : std::string());
	}

	void parse_text(std::string_view text)
	{
		std::size_t b = 0;
		while (b < text.size())
		{
			std::size_t e = text.find('\n', b);
			if (std::string_view::npos == e) e = text.size();
			std::string line(text.substr(b, e - b));
			if (!line.empty() && '\r' == line.back()) line.pop_back();
			parse_line(line);
			b = e + 1;
		}
		index_refs();
	}

	// seqan3's field order: @HD, @SQ, @RG, @PG (ID PN CL PP DS VN), @CO.
	std::string to_text() const
	{
		std::string t = "@HD\tVN:" + hd_version + "\n";
		for (auto const &sq : refs)
		{
			t += "@SQ\tSN:" + sq.name + "\tLN:" + std::to_string(sq.length);
			if (!sq.extra.empty()) t += "\t" + sq.extra;
			t += "\n";
		}
		for (auto const &rg : read_groups)
		{
			t += "@RG\tID:" + rg.first;
			if (!rg.second.empty()) t += "\t" + rg.second;
			t += "\n";
		}
		for (auto const &pg : programs)
		{
			t += "@PG\tID:" + pg.id;
			if (!pg.name.empty()) t += "\tPN:" + pg.name;
			if (!pg.command_line.empty()) t += "\tCL:" + pg.command_line;
			if (!pg.previous.empty()) t += "\tPP:" + pg.previous;
			if (!pg.description.empty()) t += "\tDS:" + pg.description;
			if (!pg.version.empty()) t += "\tVN:" + pg.version;
			if (!pg.other.empty()) t += "\t" + pg.other;
			t += "\n";
		}
		for (auto const &co : comments) t += "@CO\t" + co + "\n";
		return t;
	}
};


// One alignment record in BAM layout (without the leading block_size).
struct bam_record
{
	std::vector <uint8_t> data;

	template <typename t_value> t_value get(std::size_t off) const { t_value v; std::memcpy(&v, data.data() + off, sizeof(v)); return v; }
	template <typename t_value> void set(std::size_t off, t_value v) { std::memcpy(data.data() + off, &v, sizeof(v)); }

	int32_t ref_id() const { return get <int32_t>(0); }
	int32_t pos() const { return get <int32_t>(4); }
	uint8_t l_read_name() const { return data[8]; }
	uint8_t mapq() const { return data[9]; }
	uint16_t n_cigar() const { return get <uint16_t>(12); }
	uint16_t flag() const { return get <uint16_t>(14); }
	uint32_t l_seq() const { return get <uint32_t>(16); }
	int32_t next_ref_id() const { return get <int32_t>(20); }
	int32_t next_pos() const { return get <int32_t>(24); }
	int32_t tlen() const { return get <int32_t>(28); }
	char const *read_name() const { return (char const *) data.data() + 32; }
	std::size_t cigar_offset() const { return 32 + l_read_name(); }
	std::size_t seq_offset() const { return cigar_offset() + 4 * (std::size_t) n_cigar(); }
	std::size_t qual_offset() const { return seq_offset() + (l_seq() + 1) / 2; }
	std::size_t aux_offset() const { return qual_offset() + l_seq(); }
	uint8_t const *seq4() const { return data.data() + seq_offset(); }
	uint8_t const *qual() const { return data.data() + qual_offset(); }
	bool has_qual() const { return l_seq() && 0xff != qual()[0]; }
	uint8_t const *bytes() const { return data.data(); }
	std::size_t size() const { return data.size(); }
};

// The same record layout over bytes owned by somebody else (a batch's arena): no allocation per record.
struct bam_view
{
	uint8_t const *p{};
	std::size_t n{};

	template <typename t_value> t_value get(std::size_t off) const { t_value v; std::memcpy(&v, p + off, sizeof(v)); return v; }
	int32_t ref_id() const { return get <int32_t>(0); }
	int32_t pos() const { return get <int32_t>(4); }
	uint8_t l_read_name() const { return p[8]; }
	uint8_t mapq() const { return p[9]; }
	uint16_t n_cigar() const { return get <uint16_t>(12); }
	uint16_t flag() const { return get <uint16_t>(14); }
	uint32_t l_seq() const { return get <uint32_t>(16); }
	int32_t next_ref_id() const { return get <int32_t>(20); }
	int32_t next_pos() const { return get <int32_t>(24); }
	int32_t tlen() const { return get <int32_t>(28); }
	char const *read_name() const { return (char const *) p + 32; }
	std::size_t cigar_offset() const { return 32 + l_read_name(); }
	std::size_t seq_offset() const { return cigar_offset() + 4 * (std::size_t) n_cigar(); }
	std::size_t qual_offset() const { return seq_offset() + (l_seq() + 1) / 2; }
	std::size_t aux_offset() const { return qual_offset() + l_seq(); }
	uint8_t const *seq4() const { return p + seq_offset(); }
	uint8_t const *qual() const { return p + qual_offset(); }
	bool has_qual() const { return l_seq() && 0xff != qual()[0]; }
	uint8_t const *bytes() const { return p; }
	std::size_t size() const { return n; }
};

constexpr char const CIGAR_CHARS[] = "MIDNSHP=X";
constexpr char const SEQ4_CHARS[] = "=ACMGRSVTWYHKDBN";

inline uint8_t seq4_code(char c)
{
	switch (c)
	{
		case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
		case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
		case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
		case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15;
	}
}

// SAMv1 §5.3 reg2bin.
inline uint16_t reg2bin(int64_t beg, int64_t end)
{
	--end;
	if (beg >> 14 == end >> 14) return (uint16_t) (((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (uint16_t) (((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (uint16_t) (((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (uint16_t) (((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (uint16_t) (((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

struct aux_field
{
	char tag[2];
	char type;				// A c C s S i I f Z H B
	uint8_t const *value;	// points at the value (for B: the subtype byte)
	std::size_t size;		// bytes of the value
};

inline std::size_t aux_scalar_size(char t)
{
	switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; }
}

// Splits the aux block into `out` (cleared first; reuse it across records); throws on malformed data.
inline void parse_aux(uint8_t const *p, uint8_t const *end, std::vector <aux_field> &out)
{
	out.clear();
	while (p < end)
	{
		if (end - p < 3) throw io_error("truncated BAM tag");
		aux_field f{{(char) p[0], (char) p[1]}, (char) p[2], p + 3, 0};
		p += 3;
		if (std::size_t const s = aux_scalar_size(f.type)) f.size = s;
		else if ('Z' == f.type || 'H' == f.type)
		{
			uint8_t const *q = (uint8_t const *) std::memchr(p, 0, (std::size_t) (end - p));
			if (!q) throw io_error("unterminated BAM string tag");
			f.size = (std::size_t) (q - p) + 1;
		}
		else if ('B' == f.type)
		{
			if (end - p < 5) throw io_error("truncated BAM array tag");
			uint32_t n; std::memcpy(&n, p + 1, 4);
			std::size_t const es = aux_scalar_size((char) p[0]);
			if (!es) throw io_error("bad BAM array tag subtype");
			f.size = 5 + es * n;
		}
		else throw io_error(std::string("unknown BAM tag type ") + f.type);
		if ((std::size_t) (end - p) < f.size) throw io_error("truncated BAM tag value");
		p += f.size;
		out.push_back(f);
	}
}

inline std::vector <aux_field> parse_aux(uint8_t const *p, uint8_t const *end)
{
	std::vector <aux_field> out;
	parse_aux(p, end, out);
	return out;
}

// Decimal digits of v appended to s (std::to_string allocates a string per call).
inline void append_decimal(std::string &s, int64_t v)
{
	char buf[24];
	char *e = buf + sizeof(buf), *b = e;
	uint64_t u = (v < 0) ? (uint64_t) 0 - (uint64_t) v : (uint64_t) v;
	do { *--b = (char) ('0' + u % 10); u /= 10; } while (u);
	if (v < 0) *--b = '-';
	s.append(b, (std::size_t) (e - b));
}

inline void append_int_aux(std::vector <uint8_t> &aux, char const tag[2], int64_t v)
{
	aux.push_back((uint8_t) tag[0]); aux.push_back((uint8_t) tag[1]);
	// Smallest type that holds the value (as htslib does; seqan3's choice is not pinned by any reference test).
	auto put = [&](char t, auto x) { aux.push_back((uint8_t) t); uint8_t b[sizeof(x)]; std::memcpy(b, &x, sizeof(x)); aux.insert(aux.end(), b, b + sizeof(x)); };
	if (v >= 0)
	{
		if (v <= 0xff) put('C', (uint8_t) v);
		else if (v <= 0xffff) put('S', (uint16_t) v);
		else put('I', (uint32_t) v);
	}
	else
	{
		if (v >= -128) put('c', (int8_t) v);
		else if (v >= -32768) put('s', (int16_t) v);
		else put('i', (int32_t) v);
	}
}

inline void format_aux_sam(std::string &out, aux_field const &f)
{
	out += '\t'; out += f.tag[0]; out += f.tag[1]; out += ':';
	auto rd = [&](auto proto, uint8_t const *p) { decltype(proto) v; std::memcpy(&v, p, sizeof(v)); return v; };
	switch (f.type)
	{
		case 'A': out += "A:"; out += (char) f.value[0]; break;
		case 'c': out += "i:" + std::to_string((int) rd(int8_t{}, f.value)); break;
		case 'C': out += "i:" + std::to_string((unsigned) f.value[0]); break;
		case 's': out += "i:" + std::to_string(rd(int16_t{}, f.value)); break;
		case 'S': out += "i:" + std::to_string(rd(uint16_t{}, f.value)); break;
		case 'i': out += "i:" + std::to_string(rd(int32_t{}, f.value)); break;
		case 'I': out += "i:" + std::to_string(rd(uint32_t{}, f.value)); break;
		case 'f': { char buf[64]; std::snprintf(buf, sizeof(buf), "%g", (double) rd(float{}, f.value)); out += "f:"; out += buf; break; }
		case 'Z': out += "Z:"; out.append((char const *) f.value, f.size - 1); break;
		case 'H': out += "H:"; out.append((char const *) f.value, f.size - 1); break;
		case 'B':
		{
			char const st = (char) f.value[0];
			uint32_t n; std::memcpy(&n, f.value + 1, 4);
			out += "B:"; out += st;
			uint8_t const *p = f.value + 5;
			for (uint32_t i = 0; i < n; ++i)
			{
				out += ',';
				switch (st)
				{
					case 'c': out += std::to_string((int) rd(int8_t{}, p)); p += 1; break;
					case 'C': out += std::to_string((unsigned) p[0]); p += 1; break;
					case 's': out += std::to_string(rd(int16_t{}, p)); p += 2; break;
					case 'S': out += std::to_string(rd(uint16_t{}, p)); p += 2; break;
					case 'i': out += std::to_string(rd(int32_t{}, p)); p += 4; break;
					case 'I': out += std::to_string(rd(uint32_t{}, p)); p += 4; break;
					case 'f': { char buf[64]; std::snprintf(buf, sizeof(buf), "%g", (double) rd(float{}, p)); out += buf; p += 4; break; }
					default: break;
				}
			}
			break;
		}
		default: break;
	}
}

// SAM text of one record (no trailing newline).
template <typename t_record>
inline void format_sam_record(std::string &out, t_record const &r, sam_header const &hdr)
{
	auto ref_name = [&](int32_t id) -> std::string { return (id >= 0 && (std::size_t) id < hdr.refs.size()) ? hdr.refs[(std::size_t) id].name : std::string("*"); };
	out.append(r.read_name());
	out += '\t'; out += std::to_string(r.flag());
	out += '\t'; out += ref_name(r.ref_id());
	out += '\t'; out += std::to_string((int64_t) r.pos() + 1);
	out += '\t'; out += std::to_string((unsigned) r.mapq());
	out += '\t';
	if (0 == r.n_cigar()) out += '*';
	for (uint16_t i = 0; i < r.n_cigar(); ++i)
	{
		uint32_t const c = r.template get <uint32_t>(r.cigar_offset() + 4u * i);
		out += std::to_string(c >> 4); out += CIGAR_CHARS[(c & 0xf) < 9 ? (c & 0xf) : 0];
	}
	out += '\t';
	if (r.next_ref_id() < 0) out += '*';
	else if (r.next_ref_id() == r.ref_id()) out += '=';
	else out += ref_name(r.next_ref_id());
	out += '\t'; out += std::to_string((int64_t) r.next_pos() + 1);
	out += '\t'; out += std::to_string(r.tlen());
	out += '\t';
	uint32_t const l = r.l_seq();
	if (0 == l) out += '*';
	for (uint32_t i = 0; i < l; ++i)
		out += SEQ4_CHARS[(r.seq4()[i >> 1] >> ((~i & 1) << 2)) & 0xf];
	out += '\t';
	if (!r.has_qual()) out += '*';
	else for (uint32_t i = 0; i < l; ++i) out += (char) (33 + r.qual()[i]);
	for (auto const &f : parse_aux(r.bytes() + r.aux_offset(), r.bytes() + r.size()))
		format_aux_sam(out, f);
}

// SAM text line -> BAM layout.
inline void parse_sam_record(std::string const &line, sam_header const &hdr, bam_record &r)
{
	auto const f = sam_header::split_tabs(line);
	if (f.size() < 11) throw io_error("SAM record with fewer than 11 fields: " + line.substr(0, 80));
	auto to_ll = [](std::string_view s) { return std::atoll(std::string(s).c_str()); };
	auto ref_of = [&](std::string_view s) -> int32_t {
		if ("*" == s) return -1;
		auto const it = hdr.ref_index.find(std::string(s));
		if (hdr.ref_index.end() == it) throw io_error("RNAME not in the header: " + std::string(s));
		return it->second;
	};
	std::string_view const qname = f[0];
	int32_t const ref_id = ref_of(f[2]);
	int32_t const next_ref = ("=" == f[6]) ? ref_id : ref_of(f[6]);
	std::vector <uint32_t> cigar;
	if ("*" != f[5])
	{
		uint64_t n = 0;
		for (char c : f[5])
		{
			if (c >= '0' && c <= '9') n = 10 * n + (uint64_t) (c - '0');
			else
			{
				char const *p = std::strchr(CIGAR_CHARS, c);
				if (!p || !c) throw io_error("bad CIGAR operation in " + std::string(f[5]));
				cigar.push_back((uint32_t) (n << 4) | (uint32_t) (p - CIGAR_CHARS));
				n = 0;
			}
		}
	}
	std::string_view const seq = ("*" == f[9]) ? std::string_view() : f[9];
	uint32_t const l_seq = (uint32_t) seq.size();
	std::size_t const l_name = qname.size() + 1;
	if (l_name > 255) throw io_error("QNAME too long");
	r.data.assign(32 + l_name + 4 * cigar.size() + (l_seq + 1) / 2 + l_seq, 0);
	r.set <int32_t>(0, ref_id);
	r.set <int32_t>(4, (int32_t) (to_ll(f[3]) - 1));
	r.data[8] = (uint8_t) l_name;
	r.data[9] = (uint8_t) to_ll(f[4]);
	r.set <uint16_t>(12, (uint16_t) cigar.size());
	r.set <uint16_t>(14, (uint16_t) to_ll(f[1]));
	r.set <uint32_t>(16, l_seq);
	r.set <int32_t>(20, next_ref);
	r.set <int32_t>(24, (int32_t) (to_ll(f[7]) - 1));
	r.set <int32_t>(28, (int32_t) to_ll(f[8]));
	std::memcpy(r.data.data() + 32, qname.data(), qname.size());
	std::memcpy(r.data.data() + r.cigar_offset(), cigar.data(), 4 * cigar.size());
	uint8_t *s4 = r.data.data() + r.seq_offset();
	for (uint32_t i = 0; i < l_seq; ++i)
		s4[i >> 1] |= (uint8_t) (seq4_code(seq[i]) << ((~i & 1) << 2));
	uint8_t *q = r.data.data() + r.qual_offset();
	if ("*" == f[10]) std::memset(q, 0xff, l_seq);
	else
	{
		if (f[10].size() != l_seq) throw io_error("SEQ and QUAL differ in length");
		for (uint32_t i = 0; i < l_seq; ++i) q[i] = (uint8_t) (f[10][i] - 33);
	}
	// bin from the reference span
	int64_t span = 0;
	for (uint32_t c : cigar) { uint32_t const op = c & 0xf; if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) span += c >> 4; }
	int64_t const pos = r.pos();
	r.set <uint16_t>(10, reg2bin(pos, pos + (span ? span : 1)));

	// tags
	for (std::size_t i = 11; i < f.size(); ++i)
	{
		std::string_view const t = f[i];
		if (t.size() < 5 || ':' != t[2] || ':' != t[4]) throw io_error("malformed SAM tag: " + std::string(t));
		char const tag[2] = {t[0], t[1]};
		std::string_view const val = t.substr(5);
		auto &d = r.data;
		switch (t[3])
		{
			case 'A': d.push_back((uint8_t) tag[0]); d.push_back((uint8_t) tag[1]); d.push_back('A'); d.push_back((uint8_t) (val.empty() ? ' ' : val[0])); break;
			case 'i': append_int_aux(d, tag, to_ll(val)); break;
			case 'f': { float v = (float) std::atof(std::string(val).c_str()); d.push_back((uint8_t) tag[0]); d.push_back((uint8_t) tag[1]); d.push_back('f'); uint8_t b[4]; std::memcpy(b, &v, 4); d.insert(d.end(), b, b + 4); break; }
			case 'Z': case 'H': d.push_back((uint8_t) tag[0]); d.push_back((uint8_t) tag[1]); d.push_back((uint8_t) t[3]); d.insert(d.end(), val.begin(), val.end()); d.push_back(0); break;
			case 'B':
			{
				if (val.empty()) throw io_error("empty B tag");
				char const st = val[0];
				std::size_t const es = aux_scalar_size(st);
				if (!es) throw io_error("bad B tag subtype");
				std::vector <std::string_view> items;
				if (val.size() > 1) { auto rest = val.substr(2); std::size_t b = 0; for (;;) { auto e = rest.find(',', b); items.push_back(rest.substr(b, std::string_view::npos == e ? e : e - b)); if (std::string_view::npos == e) break; b = e + 1; } }
				d.push_back((uint8_t) tag[0]); d.push_back((uint8_t) tag[1]); d.push_back('B'); d.push_back((uint8_t) st);
				uint32_t const n = (uint32_t) items.size(); uint8_t nb[4]; std::memcpy(nb, &n, 4); d.insert(d.end(), nb, nb + 4);
				for (auto it : items)
				{
					uint8_t b[4];
					if ('f' == st) { float v = (float) std::atof(std::string(it).c_str()); std::memcpy(b, &v, 4); }
					else { int64_t v = to_ll(it); std::memcpy(b, &v, es); }
					d.insert(d.end(), b, b + es);
				}
				break;
			}
			default: throw io_error("unknown SAM tag type in " + std::string(t));
		}
	}
}


// append_sam_program_info (include/panvc3/utility.hh:92-143): a new @PG line "<prefix><k + 1>" chained to the last program.
inline void append_sam_program_info(sam_header const &in, sam_header &out, std::string const &prefix, std::string const &name, int argc, char **argv)
{
	out.programs = in.programs;
	uint32_t idx = 0;
	for (auto const &pg : out.programs)
		if (0 == pg.id.compare(0, prefix.size(), prefix))
		{
			char *end{};
			unsigned long const v = std::strtoul(pg.id.c_str() + prefix.size(), &end, 10);
			if (end != pg.id.c_str() + prefix.size()) idx = std::max <uint32_t>(idx, (uint32_t) v);
		}
	pg_line pg;
	pg.id = prefix + std::to_string(idx + 1);
	pg.name = name;
	for (int i = 0; i < argc; ++i) { if (i) pg.command_line += ' '; pg.command_line += argv[i]; }
	pg.version = "0.1-mi355x";
	if (!out.programs.empty()) pg.previous = out.programs.back().id;
	out.programs.push_back(std::move(pg));
}


// ---- input: SAM text or BAM -----------------------------------------------------------------------------------------------
class alignment_reader
{
	byte_reader m_in;
	bool m_is_bam{};
	std::string m_line, m_pending;
	bool m_has_pending{};
	uint64_t m_bam_header_bytes{};

public:
	sam_header header;

	// BAM input: the uncompressed bytes in front of the first record (magic, header text, reference list).
	uint64_t bam_header_bytes() const { return m_bam_header_bytes; }
	bool is_bgzf_file() const { return m_in.is_bgzf_file(); }
	void set_piece_bytes(std::size_t n) { m_in.set_piece_bytes(n); }

	void set_threads(unsigned n) { m_in.set_threads(n); }
	void attach(group_inflater *g) { m_in.attach(g); }		// BGZF input inflated elsewhere from the next piece on (byte_reader::attach)
	void release() { m_in.release(); }						// the input is closed and its buffers are given back
	uint64_t group_blocks() const { return m_in.group_blocks(); }
	uint64_t group_refused() const { return m_in.group_refused(); }

	void open(char const *path, bool bam)
	{
		m_in.open(path);
		m_is_bam = bam;
		if (bam)
		{
			char magic[4];
			if (4 != m_in.read(magic, 4) || 0 != std::memcmp(magic, "BAM\1", 4)) throw io_error("not a BAM file");
			int32_t l_text;
			if (4 != m_in.read(&l_text, 4) || l_text < 0) throw io_error("truncated BAM header");
			std::string text((std::size_t) l_text, '\0');
			if ((std::size_t) l_text != m_in.read(text.data(), (std::size_t) l_text)) throw io_error("truncated BAM header");
			while (!text.empty() && '\0' == text.back()) text.pop_back();
			header.parse_text(text);
			int32_t n_ref;
			if (4 != m_in.read(&n_ref, 4) || n_ref < 0) throw io_error("truncated BAM header");
			// The binary reference list is authoritative for the ids.
			std::vector <sq_line> refs((std::size_t) n_ref);
			for (auto &sq : refs)
			{
				int32_t l_name, l_ref;
				if (4 != m_in.read(&l_name, 4) || l_name <= 0) throw io_error("truncated BAM header");
				sq.name.resize((std::size_t) l_name);
				m_in.read(sq.name.data(), (std::size_t) l_name);
				sq.name.pop_back();
				if (4 != m_in.read(&l_ref, 4)) throw io_error("truncated BAM header");
				sq.length = l_ref;
			}
			for (auto &sq : refs)
				for (auto const &t : header.refs)
					if (t.name == sq.name) { sq.extra = t.extra; break; }
			m_bam_header_bytes = 12 + (uint64_t) l_text;
			for (auto const &sq : refs) m_bam_header_bytes += 8 + sq.name.size() + 1;
			header.refs = std::move(refs);
			header.index_refs();
		}
		else
		{
			while ('@' == m_in.peek() && m_in.getline(m_line))
				header.parse_line(m_line);
			header.index_refs();
		}
	}

	bool next(bam_record &r)
	{
		if (m_is_bam)
		{
			int32_t block_size;
			std::size_t const got = m_in.read(&block_size, 4);
			if (0 == got) return false;
			if (4 != got || block_size < 32) throw io_error("truncated BAM record");
			r.data.resize((std::size_t) block_size);
			if ((std::size_t) block_size != m_in.read(r.data.data(), (std::size_t) block_size)) throw io_error("truncated BAM record");
			// l_read_name, n_cigar_op and l_seq must fit the record: everything downstream slices the buffer by them
			if (r.aux_offset() > r.data.size()) throw io_error("malformed BAM record: its name, CIGAR, SEQ and QUAL exceed block_size");
			return true;
		}
		while (m_in.getline(m_line))
		{
			if (m_line.empty() || '@' == m_line[0]) continue;
			parse_sam_record(m_line, header, r);
			return true;
		}
		return false;
	}

	// BAM from a BGZF file: the inflated bytes not yet read, for framing whole records in place (see byte_reader::span).
	bool bam_span(uint8_t const *&p, std::size_t &n) { return m_is_bam && m_in.span(p, n); }
	void consume(std::size_t n) { m_in.consume(n); }

	// The next record's bytes appended to `arena` (no allocation per record); returns false at the end of the input.
	bool next_append(std::vector <uint8_t> &arena, bam_record &scratch)
	{
		if (m_is_bam)
		{
			int32_t block_size;
			std::size_t const got = m_in.read(&block_size, 4);
			if (0 == got) return false;
			if (4 != got || block_size < 32) throw io_error("truncated BAM record");
			std::size_t const at = arena.size();
			if ((std::size_t) block_size != m_in.read_append(arena, (std::size_t) block_size)) throw io_error("truncated BAM record");
			bam_view const v{arena.data() + at, (std::size_t) block_size};
			if (v.aux_offset() > v.size()) throw io_error("malformed BAM record: its name, CIGAR, SEQ and QUAL exceed block_size");
			return true;
		}
		if (!next(scratch)) return false;
		arena.insert(arena.end(), scratch.data.begin(), scratch.data.end());
		return true;
	}
};


// ---- output ---------------------------------------------------------------------------------------------------------------------
class alignment_writer
{
	byte_writer m_out;
	bool m_is_bam{};
	std::string m_buf;
	sam_header const *m_header{};

public:
	void set_threads(unsigned n) { m_out.set_threads(n); }
	void attach(group_deflater *g) { m_out.attach(g); }		// BAM blocks compressed elsewhere from here on (byte_writer::attach)

	void open(char const *path, bool bam, sam_header const &hdr, int compression_level = -1)
	{
		m_is_bam = bam; m_header = &hdr;
		m_out.open(path, bam, compression_level);
		std::string const text = hdr.to_text();
		if (bam)
		{
			m_out.write("BAM\1", 4);
			int32_t const l_text = (int32_t) text.size();
			m_out.write(&l_text, 4); m_out.write(text.data(), text.size());
			int32_t const n_ref = (int32_t) hdr.refs.size();
			m_out.write(&n_ref, 4);
			for (auto const &sq : hdr.refs)
			{
				int32_t const l_name = (int32_t) sq.name.size() + 1, l_ref = (int32_t) sq.length;
				m_out.write(&l_name, 4); m_out.write(sq.name.c_str(), (std::size_t) l_name); m_out.write(&l_ref, 4);
			}
		}
		else m_out.write(text);
	}

	void write(bam_record const &r)
	{
		if (m_is_bam)
		{
			int32_t const bs = (int32_t) r.data.size();
			m_out.write(&bs, 4); m_out.write(r.data.data(), r.data.size());
		}
		else
		{
			m_buf.clear();
			format_sam_record(m_buf, r, *m_header);
			m_buf += '\n';
			m_out.write(m_buf);
		}
	}

	// Appends to `out` exactly the bytes write(r) would emit (callable from several threads at once).
	template <typename t_record>
	void format(t_record const &r, std::string &out) const
	{
		if (m_is_bam)
		{
			int32_t const bs = (int32_t) r.size();
			out.append((char const *) &bs, 4);
			out.append((char const *) r.bytes(), r.size());
		}
		else
		{
			format_sam_record(out, r, *m_header);
			out += '\n';
		}
	}

	void write_formatted(char const *p, std::size_t n) { m_out.write(p, n); }

	// For a caller that makes the BGZF blocks of the records itself (the record path on the GPU): what has been written so far (the
	// header) goes out as blocks of its own, then the caller's blocks as they are; close() ends the file as usual.
	void flush_blocks() { m_out.flush_all(); }
	void write_blocks(void const *p, std::size_t n) { m_out.write_framed(p, n); }

	void close() { m_out.close(); }
};

}
