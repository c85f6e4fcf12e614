// fast_deflate.hpp — a small, fast DEFLATE (RFC 1951) encoder for BGZF blocks.
//
// The BAM writer of panvc3_project_alignments spent 10 core-seconds per million records in zlib's deflate at its default
// level, six times everything else in the pipeline.  BGZF blocks are independent and at most 65 280 bytes long, which
// makes a much simpler encoder sufficient: one greedy pass with a single-probe hash table over 4-byte sequences (positions
// fit 16 bits) that records the matches, one dynamic Huffman block per input (its exact size is known from the symbol counts
// before a bit is written, so the bit stream is written without bounds checks), canonical codes limited to 15 bits.  The output is an ordinary
// DEFLATE stream (any inflate reads it); it is about 10 % larger than zlib's level 6 on BAM records.
// No reference code corresponds to this file: the reference writes BAM through seqan3 / zlib.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pvc3host {

class fast_deflater
{
	static constexpr unsigned HASH_BITS = 15;
	static constexpr unsigned MIN_MATCH = 4, MAX_MATCH = 258, MAX_DIST = 32768;
	static constexpr unsigned N_LITLEN = 286, N_DIST = 30, N_CLEN = 19;

	std::vector <uint16_t> m_head;		// hash -> position + 1 of the last occurrence (0: none)

	struct tables
	{
		uint8_t len_sym[256];			// length - 3 -> symbol - 257
		uint8_t len_extra[29];
		uint16_t len_base[29];
		uint8_t dist_sym_lo[256];		// distance - 1 < 256
		uint8_t dist_sym_hi[256];		// (distance - 1) >> 7
		uint8_t dist_extra[30];
		uint16_t dist_base[30];
		tables()
		{
			static uint8_t const le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
			unsigned b = 3;
			for (unsigned s = 0; s < 29; ++s)
			{
				len_extra[s] = le[s]; len_base[s] = (uint16_t) b;
				if (28 == s) { len_base[s] = 258; len_sym[255] = 28; break; }
				for (unsigned k = 0; k < (1u << le[s]) && b - 3 < 255; ++k, ++b) len_sym[b - 3] = (uint8_t) s;
			}
			unsigned d = 1;
			for (unsigned s = 0; s < 30; ++s)
			{
				unsigned const e = (s < 4) ? 0 : (s / 2 - 1);
				dist_extra[s] = (uint8_t) e; dist_base[s] = (uint16_t) d;		// base 1 .. 24577
				for (unsigned k = 0; k < (1u << e); ++k, ++d)
				{
					if (d - 1 < 256) dist_sym_lo[d - 1] = (uint8_t) s;
					if (0 == ((d - 1) & 127)) dist_sym_hi[(d - 1) >> 7] = (uint8_t) s;
				}
			}
			// distances whose (d - 1) >> 7 bucket starts inside a code's range: all buckets >= 2 are covered above because codes
			// with more than 6 extra bits span whole multiples of 128; buckets 0 and 1 are only reached through dist_sym_lo.
		}
	};
	static tables const &tab() { static tables const t; return t; }

	static uint32_t load32(uint8_t const *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
	static uint64_t load64(uint8_t const *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

	// Code lengths of a complete prefix code for freq[0 .. n), none longer than `limit`; unused symbols get 0.  Huffman's
	// construction; should a code come out longer than the limit (it takes a Fibonacci-like distribution), the frequencies
	// are halved (rounding up) and the construction repeated, which flattens the tree and always ends.
	static void code_lengths(uint32_t const *freq_in, unsigned n, unsigned limit, uint8_t *len)
	{
		struct node { uint32_t f; int16_t left, right; };
		uint32_t freq[N_LITLEN];
		uint16_t order[N_LITLEN];
		unsigned used = 0;
		for (unsigned i = 0; i < n; ++i) { len[i] = 0; freq[i] = freq_in[i]; if (freq[i]) order[used++] = (uint16_t) i; }
		if (0 == used) return;
		if (1 == used) { len[order[0]] = 1; return; }
		for (;;)
		{
			std::sort(order, order + used, [&](uint16_t a, uint16_t b) { return freq[a] < freq[b] || (freq[a] == freq[b] && a < b); });
			// Two-queue construction: leaves in `order`, internal nodes come into being in non-decreasing weight.
			node nodes[2 * N_LITLEN];
			for (unsigned i = 0; i < used; ++i) nodes[i] = node{freq[order[i]], -1, -1};
			unsigned leaf = 0, inner = used, next = used;
			auto take = [&]() -> unsigned {
				if (leaf < used && (inner >= next || nodes[leaf].f <= nodes[inner].f)) return leaf++;
				return inner++;
			};
			while ((used - leaf) + (next - inner) > 1)
			{
				unsigned const a = take(), b = take();
				nodes[next] = node{nodes[a].f + nodes[b].f, (int16_t) a, (int16_t) b};
				++next;
			}
			// Depths, root first (children have smaller indices than their parents).
			uint16_t depth[2 * N_LITLEN];
			depth[next - 1] = 0;
			unsigned deepest = 0;
			for (unsigned i = next; i-- > used; )
			{
				depth[nodes[i].left] = (uint16_t) (depth[i] + 1);
				depth[nodes[i].right] = (uint16_t) (depth[i] + 1);
			}
			for (unsigned i = 0; i < used; ++i) deepest = std::max <unsigned>(deepest, depth[i]);
			if (deepest <= limit)
			{
				for (unsigned i = 0; i < used; ++i) len[order[i]] = (uint8_t) depth[i];
				return;
			}
			for (unsigned i = 0; i < used; ++i) freq[order[i]] = (freq[order[i]] + 1) / 2;
		}
	}

	// Canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first bit stream.
	static void assign_codes(uint8_t const *len, unsigned n, uint16_t *code)
	{
		unsigned count[16] = {0}, next[16];
		for (unsigned i = 0; i < n; ++i) ++count[len[i]];
		count[0] = 0;
		unsigned c = 0;
		for (unsigned l = 1; l < 16; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
		for (unsigned i = 0; i < n; ++i)
		{
			unsigned const l = len[i];
			if (0 == l) { code[i] = 0; continue; }
			unsigned v = next[l]++, r = 0;
			for (unsigned k = 0; k < l; ++k) { r = (r << 1) | (v & 1); v >>= 1; }
			code[i] = (uint16_t) r;
		}
	}

	static std::size_t stored(uint8_t const *src, std::size_t n, uint8_t *dst, std::size_t cap)
	{
		if (cap < n + 5 || n > 65535) return 0;
		dst[0] = 1;		// BFINAL = 1, BTYPE = 00
		dst[1] = (uint8_t) n; dst[2] = (uint8_t) (n >> 8); dst[3] = (uint8_t) ~dst[1]; dst[4] = (uint8_t) ~dst[2];
		std::memcpy(dst + 5, src, n);
		return n + 5;
	}

	struct match { uint16_t pos, len, dist; };
	std::vector <match> m_matches;

public:
	static std::size_t bound(std::size_t n) { return n + 5 + 64; }		// a stored block is the fallback

	// One final DEFLATE block (dynamic Huffman, or stored when that is smaller) for src[0 .. n), n <= 65535.
	// Returns the number of bytes written to dst, or 0 if cap < bound(n).
	std::size_t compress(uint8_t const *src, std::size_t n, uint8_t *dst, std::size_t cap)
	{
		if (cap < bound(n) || n > 65535) return 0;
		if (0 == n) { dst[0] = 3; dst[1] = 0; return 2; }		// fixed-code block with the end-of-block symbol only
		tables const &T = tab();
		m_head.assign(std::size_t(1) << HASH_BITS, 0);
		m_matches.clear();
		// Literal counts in two tables (even / odd positions): base qualities repeat one value for long stretches, and a single
		// counter would make every increment wait for the one before.
		uint32_t f_lit[2][256] = {{0}}, f_ll[N_LITLEN] = {0}, f_d[N_DIST] = {0};
		uint64_t extra_bits = 0;

		// ---- greedy parse: the matches; everything between them is literals ------------------------------------------------------
		std::size_t i = 0, misses = 0;
		std::size_t const last_hashable = (n >= 8) ? (n - 8) : 0;		// matches are extended with 8-byte loads
		while (i < last_hashable)
		{
			uint32_t const v = load32(src + i);
			uint32_t const h = (v * 2654435761u) >> (32 - HASH_BITS);
			unsigned const cand = m_head[h];
			m_head[h] = (uint16_t) (i + 1);
			if (cand && i + 1 - cand <= MAX_DIST && load32(src + cand - 1) == v)
			{
				std::size_t const c = cand - 1;
				std::size_t const max_len = std::min <std::size_t>(MAX_MATCH, n - i);
				std::size_t len = 4;
				while (len + 8 <= max_len)
				{
					uint64_t const x = load64(src + i + len) ^ load64(src + c + len);
					if (x) { len += (std::size_t) (__builtin_ctzll(x) >> 3); goto matched; }
					len += 8;
				}
				while (len < max_len && src[i + len] == src[c + len]) ++len;
			matched:
				unsigned const dist = (unsigned) (i - c);
				m_matches.push_back(match{(uint16_t) i, (uint16_t) len, (uint16_t) (dist - 1)});
				unsigned const ls = T.len_sym[len - 3], ds = (dist - 1 < 256) ? T.dist_sym_lo[dist - 1] : T.dist_sym_hi[(dist - 1) >> 7];
				++f_ll[257 + ls]; ++f_d[ds];
				extra_bits += T.len_extra[ls] + T.dist_extra[ds];
				// a position inside the match keeps the table fresh for what follows it
				if (i + len < last_hashable)
				{
					std::size_t const j = i + len - 1;
					m_head[(load32(src + j) * 2654435761u) >> (32 - HASH_BITS)] = (uint16_t) (j + 1);
				}
				i += len;
				misses = 0;
				continue;
			}
			// No match here: literals, in larger steps the longer nothing has matched (SEQ and QUAL hardly ever do).
			std::size_t const step = 1 + (misses >> 3);
			++misses;
			std::size_t const stop = std::min(i + step, n);
			for (; i < stop; ++i) ++f_lit[i & 1][src[i]];
		}
		for (; i < n; ++i) ++f_lit[i & 1][src[i]];
		for (unsigned b = 0; b < 256; ++b) f_ll[b] = f_lit[0][b] + f_lit[1][b];
		f_ll[256] = 1;

		// ---- codes -----------------------------------------------------------------------------------------------------------
		uint8_t l_ll[N_LITLEN], l_d[N_DIST];
		uint16_t c_ll[N_LITLEN], c_d[N_DIST];
		code_lengths(f_ll, N_LITLEN, 15, l_ll);
		code_lengths(f_d, N_DIST, 15, l_d);
		unsigned n_ll = N_LITLEN, n_d = N_DIST;
		while (n_ll > 257 && 0 == l_ll[n_ll - 1]) --n_ll;
		while (n_d > 1 && 0 == l_d[n_d - 1]) --n_d;
		if (1 == n_d && 0 == l_d[0]) l_d[0] = 1;		// no match at all: one unused distance code (RFC 1951 3.2.7)
		assign_codes(l_ll, n_ll, c_ll);
		assign_codes(l_d, n_d, c_d);
		uint64_t body_bits = extra_bits;
		for (unsigned k = 0; k < n_ll; ++k) body_bits += (uint64_t) f_ll[k] * l_ll[k];
		for (unsigned k = 0; k < n_d; ++k) body_bits += (uint64_t) f_d[k] * l_d[k];

		// The two length sequences, run-length coded with symbols 16 (repeat 3-6), 17 (zeros 3-10) and 18 (zeros 11-138).
		uint8_t seq[N_LITLEN + N_DIST];
		std::memcpy(seq, l_ll, n_ll);
		std::memcpy(seq + n_ll, l_d, n_d);
		unsigned const n_seq = n_ll + n_d;
		struct rle { uint8_t sym, extra; };
		rle out[N_LITLEN + N_DIST];
		unsigned n_out = 0;
		uint32_t f_cl[N_CLEN] = {0};
		for (unsigned k = 0; k < n_seq; )
		{
			unsigned run = 1;
			while (k + run < n_seq && seq[k + run] == seq[k]) ++run;
			unsigned const v = seq[k];
			k += run;
			if (0 == v)
				while (run)
				{
					if (run >= 11) { unsigned const r = std::min(run, 138u); out[n_out++] = rle{18, (uint8_t) (r - 11)}; ++f_cl[18]; run -= r; }
					else if (run >= 3) { out[n_out++] = rle{17, (uint8_t) (run - 3)}; ++f_cl[17]; run = 0; }
					else { out[n_out++] = rle{0, 0}; ++f_cl[0]; --run; }
				}
			else
			{
				out[n_out++] = rle{(uint8_t) v, 0}; ++f_cl[v]; --run;
				while (run)
				{
					if (run >= 3) { unsigned const r = std::min(run, 6u); out[n_out++] = rle{16, (uint8_t) (r - 3)}; ++f_cl[16]; run -= r; }
					else { out[n_out++] = rle{(uint8_t) v, 0}; ++f_cl[v]; --run; }
				}
			}
		}
		uint8_t l_cl[N_CLEN];
		uint16_t c_cl[N_CLEN];
		code_lengths(f_cl, N_CLEN, 7, l_cl);
		{
			unsigned used = 0;
			for (unsigned k = 0; k < N_CLEN; ++k) used += (0 != l_cl[k]);
			if (1 == used)		// a complete code needs two symbols: add an unused one
				for (unsigned k = 0; k < N_CLEN; ++k)
					if (0 == l_cl[k]) { l_cl[k] = 1; break; }
		}
		assign_codes(l_cl, N_CLEN, c_cl);
		static uint8_t const cl_order[N_CLEN] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
		unsigned n_cl = N_CLEN;
		while (n_cl > 4 && 0 == l_cl[cl_order[n_cl - 1]]) --n_cl;
		uint64_t head_bits = 3 + 5 + 5 + 4 + 3 * n_cl;
		for (unsigned k = 0; k < n_out; ++k)
			head_bits += l_cl[out[k].sym] + ((16 == out[k].sym) ? 2 : (17 == out[k].sym) ? 3 : (18 == out[k].sym) ? 7 : 0);
		// The size is known before a bit is written: stored instead when that is not larger, and no bounds checks below
		// (cap >= n + 5 + 64; the bit writer stores 8 bytes at a time).
		std::size_t const size = (std::size_t) ((head_bits + body_bits + 7) / 8);
		if (size >= n + 5)
			return stored(src, n, dst, cap);

		// ---- bit stream --------------------------------------------------------------------------------------------------
		uint8_t *p = dst;
		uint64_t acc = 0;
		unsigned nb = 0;		// bits in acc, < 8 after a flush
		auto put = [&](uint64_t bits, unsigned count) { acc |= bits << nb; nb += count; };
		auto flush = [&]() { std::memcpy(p, &acc, 8); p += nb >> 3; acc >>= (nb & ~7u); nb &= 7; };
		put(1, 1); put(2, 2);		// BFINAL, BTYPE = 10
		put(n_ll - 257, 5); put(n_d - 1, 5); put(n_cl - 4, 4);
		flush();
		for (unsigned k = 0; k < n_cl; ++k) { put(l_cl[cl_order[k]], 3); if (0 == ((k + 1) & 7)) flush(); }
		flush();
		for (unsigned k = 0; k < n_out; ++k)
		{
			put(c_cl[out[k].sym], l_cl[out[k].sym]);
			if (16 == out[k].sym) put(out[k].extra, 2);
			else if (17 == out[k].sym) put(out[k].extra, 3);
			else if (18 == out[k].sym) put(out[k].extra, 7);
			flush();
		}
		uint32_t lit[256];		// code | length << 16
		for (unsigned b = 0; b < 256; ++b) lit[b] = c_ll[b] | ((uint32_t) l_ll[b] << 16);
		auto literals = [&](std::size_t from, std::size_t to) {
			std::size_t k = from;
			for (; k + 3 <= to; k += 3)		// three codes of at most 15 bits + 7 left over < 64
			{
				uint32_t const a = lit[src[k]], b = lit[src[k + 1]], c = lit[src[k + 2]];
				put(a & 0xffff, a >> 16); put(b & 0xffff, b >> 16); put(c & 0xffff, c >> 16);
				flush();
			}
			for (; k < to; ++k) { uint32_t const a = lit[src[k]]; put(a & 0xffff, a >> 16); }
			flush();
		};
		std::size_t at = 0;
		for (match const &m : m_matches)
		{
			literals(at, m.pos);
			unsigned const len3 = m.len - 3u, d1 = m.dist;
			unsigned const ls = T.len_sym[len3];
			put(c_ll[257 + ls], l_ll[257 + ls]);
			put(len3 + 3 - T.len_base[ls], T.len_extra[ls]);
			unsigned const ds = (d1 < 256) ? T.dist_sym_lo[d1] : T.dist_sym_hi[d1 >> 7];
			put(c_d[ds], l_d[ds]);
			put(d1 + 1 - T.dist_base[ds], T.dist_extra[ds]);		// 15 + 5 + 15 + 13 + 7 left over < 64
			flush();
			at = (std::size_t) m.pos + m.len;
		}
		literals(at, n);
		put(c_ll[256], l_ll[256]);
		flush();
		if (nb) *p++ = (uint8_t) acc;
		return (std::size_t) (p - dst);
	}
};

}
