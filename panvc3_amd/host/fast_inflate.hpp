// fast_inflate.hpp — a DEFLATE (RFC 1951) decoder for BGZF blocks: whole input and whole output in memory, at most 64 KB each.
//
// The BGZF reader inflates every block of the input BAM; with zlib that was 3 of the 14 core-seconds that four million records
// cost panvc3_project_alignments on the 16 CPUs of the GPU box.  This decoder does what zlib's inflate_fast does — a 64-bit bit
// buffer refilled eight bytes at a time, table look-ups of 11 bits for literals / lengths and 8 bits for distances with
// second-level tables behind them, matches copied in 8-byte pieces — without the streaming state machine around it.
// Every block a caller inflates with it is verified by the caller against the CRC-32 and the length the BGZF trailer gives;
// on any error (return false) the caller falls back to zlib, so a flaw here can cost time, not data.
// Checked against zlib-made streams of every block type in tests/test_host_io_cpu.py.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace pvc3host {

class fast_inflater
{
	// Table entry: bits 0-3 code length consumed at this level (0: invalid), bit 4 "pointer to a second-level table",
	// bits 8-15 extra bits (lengths / distances) or, for a pointer, the width of the second-level table, bits 16-31 the value:
	// literal, length base (with bit 5 set) / distance base, end of block (bit 6), or the second-level table's offset.
	static constexpr uint32_t LITLEN_BITS = 11, DIST_BITS = 8;
	static constexpr uint32_t F_SUB = 1u << 4, F_LENGTH = 1u << 5, F_END = 1u << 6;
	uint32_t m_litlen[(1u << LITLEN_BITS) + 1024];
	uint32_t m_dist[(1u << DIST_BITS) + 512];

	// Canonical Huffman code -> look-up table.  lens[i] = code length of symbol i (0: unused); entry_of(symbol) gives the
	// table entry without its length field.  Returns false for an over-subscribed or (except for a single code) incomplete code.
	template <typename t_entry_of>
	static bool build(uint8_t const *lens, uint32_t n_syms, uint32_t table_bits, uint32_t *table, uint32_t table_cap, t_entry_of &&entry_of)
	{
		uint16_t count[16] = {0}, offs[16];
		for (uint32_t i = 0; i < n_syms; ++i) ++count[lens[i]];
		count[0] = 0;
		uint32_t left = 1, max_len = 0, used = 0;
		for (uint32_t l = 1; l <= 15; ++l)
		{
			left <<= 1;
			if (count[l] > left) return false;		// over-subscribed
			left -= count[l];
			if (count[l]) max_len = l;
			used += count[l];
		}
		if (0 == used)
		{
			for (uint32_t i = 0; i < (1u << table_bits); ++i) table[i] = 0;		// no codes: every look-up is invalid
			return true;
		}
		if (left && !(1 == used && 1 == max_len)) return false;		// incomplete (RFC 1951 allows one distance code of one bit)
		// symbols sorted by code length
		uint16_t sorted[320];
		offs[1] = 0;
		for (uint32_t l = 1; l < 15; ++l) offs[l + 1] = (uint16_t) (offs[l] + count[l]);
		for (uint32_t i = 0; i < n_syms; ++i) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t) i;
		// Codes are assigned in order of length, then symbol; the bit stream holds them most significant bit first, the bit
		// buffer reads least significant bit first: the tables are indexed by the bit-reversed code.  Codes longer than
		// table_bits share a second-level table with the other codes that begin with the same table_bits bits; it is as wide
		// as the longest of them needs (first pass), then every code fills its entries (second pass).
		uint32_t const root = 1u << table_bits;
		for (uint32_t i = 0; i < root; ++i) table[i] = 0;
		auto reversed = [](uint32_t code, uint32_t l) { uint32_t rev = 0; for (uint32_t b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b); return rev; };
		uint8_t sub_width[1u << 11];
		if (max_len > table_bits)
		{
			for (uint32_t i = 0; i < root; ++i) sub_width[i] = 0;
			uint32_t code = 0;
			for (uint32_t l = 1; l <= max_len; ++l)
			{
				if (l > table_bits)
					for (uint32_t k = 0; k < count[l]; ++k)
					{
						uint32_t const prefix = reversed(code + k, l) & (root - 1);
						if (l - table_bits > sub_width[prefix]) sub_width[prefix] = (uint8_t) (l - table_bits);
					}
				code = (code + count[l]) << 1;
			}
			uint32_t next_sub = root;
			for (uint32_t prefix = 0; prefix < root; ++prefix)
				if (sub_width[prefix])
				{
					uint32_t const size = 1u << sub_width[prefix];
					if (next_sub + size > table_cap) return false;
					for (uint32_t i = 0; i < size; ++i) table[next_sub + i] = 0;
					table[prefix] = F_SUB | table_bits | ((uint32_t) sub_width[prefix] << 8) | (next_sub << 16);
					next_sub += size;
				}
		}
		uint32_t code = 0, si = 0;
		for (uint32_t l = 1; l <= max_len; ++l)
		{
			for (uint32_t k = 0; k < count[l]; ++k, ++si, ++code)
			{
				uint32_t const rev = reversed(code, l);
				uint32_t const payload = entry_of(sorted[si]);
				if (l <= table_bits)
				{
					uint32_t const e = payload | l;
					for (uint32_t i = rev; i < root; i += 1u << l) table[i] = e;
				}
				else
				{
					uint32_t const head = table[rev & (root - 1)];
					uint32_t const base = head >> 16, width = (head >> 8) & 0xff;
					uint32_t const e = payload | (l - table_bits);
					for (uint32_t i = rev >> table_bits; i < (1u << width); i += 1u << (l - table_bits)) table[base + i] = e;
				}
			}
			code <<= 1;
		}
		return true;
	}

	static uint32_t litlen_entry(uint32_t sym)
	{
		static uint16_t const base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
		static uint8_t const extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
		if (sym < 256) return sym << 16;
		if (256 == sym) return F_END;
		if (sym > 285) return F_END | F_LENGTH;		// 286, 287: never valid in a stream (marked: end + length)
		return F_LENGTH | ((uint32_t) extra[sym - 257] << 8) | ((uint32_t) base[sym - 257] << 16);
	}

	static uint32_t dist_entry(uint32_t sym)
	{
		static uint16_t const base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
		static uint8_t const extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
		if (sym > 29) return F_END;					// 30, 31: invalid
		return ((uint32_t) extra[sym] << 8) | ((uint32_t) base[sym] << 16);
	}

public:
	// Inflates the raw DEFLATE stream in[0, in_len) into out[0, out_len): true iff the stream is well formed and produces
	// exactly out_len bytes.  Reads only in[0, in_len), writes only out[0, out_len).
	bool inflate(uint8_t const *in, std::size_t in_len, uint8_t *out, std::size_t out_len)
	{
		uint8_t const *ip = in, *const in_end = in + in_len;
		uint8_t *op = out, *const out_end = out + out_len;
		uint64_t bits = 0;
		uint32_t n_bits = 0;
		// Bits past the end of the input read as zeros; a stream that really needs them fails the checks below (overrun).
		std::size_t overrun = 0;
		auto refill = [&]() {
			if (in_end - ip >= 8)
			{
				uint64_t w;
				std::memcpy(&w, ip, 8);
				bits |= w << n_bits;
				uint32_t const take = (63 - n_bits) >> 3;
				ip += take; n_bits += 8 * take;
			}
			else
				while (n_bits <= 56)
				{
					if (ip < in_end) bits |= (uint64_t) *ip++ << n_bits;
					else ++overrun;
					n_bits += 8;
				}
		};
		auto take = [&](uint32_t n) -> uint32_t { uint32_t const v = (uint32_t) (bits & ((UINT64_C(1) << n) - 1)); bits >>= n; n_bits -= n; return v; };

		for (bool last = false; !last; )
		{
			refill();
			last = 0 != take(1);
			uint32_t const type = take(2);
			if (0 == type)
			{
				// stored: skip to the byte boundary, LEN, NLEN, bytes
				take(n_bits & 7);
				refill();
				uint32_t const len = take(16), nlen = take(16);
				if ((len ^ 0xffffu) != nlen) return false;
				// the bit buffer holds whole bytes here: give them back
				{
					// (only bytes that exist are given back: the pointer never leaves [in, in_end])
					std::size_t const held = n_bits >> 3;
					if (overrun > held) return false;
					ip -= held - overrun;
					overrun = 0;
				}
				bits = 0; n_bits = 0;
				if ((std::size_t) (in_end - ip) < len || (std::size_t) (out_end - op) < len) return false;
				std::memcpy(op, ip, len);
				ip += len; op += len;
				continue;
			}
			if (3 == type) return false;
			if (1 == type)
			{
				uint8_t lens[288 + 32];
				for (uint32_t i = 0; i < 144; ++i) lens[i] = 8;
				for (uint32_t i = 144; i < 256; ++i) lens[i] = 9;
				for (uint32_t i = 256; i < 280; ++i) lens[i] = 7;
				for (uint32_t i = 280; i < 288; ++i) lens[i] = 8;
				for (uint32_t i = 0; i < 32; ++i) lens[288 + i] = 5;
				if (!build(lens, 288, LITLEN_BITS, m_litlen, sizeof(m_litlen) / 4, litlen_entry)) return false;
				if (!build(lens + 288, 32, DIST_BITS, m_dist, sizeof(m_dist) / 4, dist_entry)) return false;
			}
			else
			{
				uint32_t const hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
				if (hlit > 286 || hdist > 30) return false;
				static uint8_t const order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
				uint8_t cl[19] = {0};
				refill();
				for (uint32_t i = 0; i < hclen; ++i) { if (n_bits < 3) refill(); cl[order[i]] = (uint8_t) take(3); }
				uint32_t cl_table[128];
				if (!build(cl, 19, 7, cl_table, 128, [](uint32_t sym) { return sym << 16; })) return false;
				uint8_t lens[286 + 30 + 138] = {0};
				for (uint32_t i = 0; i < hlit + hdist; )
				{
					refill();
					uint32_t const e = cl_table[bits & 127];
					if (0 == (e & 15)) return false;
					take(e & 15);
					uint32_t const sym = e >> 16;
					if (sym < 16) { lens[i++] = (uint8_t) sym; continue; }
					uint32_t rep, val = 0;
					if (16 == sym) { if (0 == i) return false; val = lens[i - 1]; rep = 3 + take(2); }
					else if (17 == sym) rep = 3 + take(3);
					else rep = 11 + take(7);
					if (i + rep > hlit + hdist) return false;
					for (uint32_t k = 0; k < rep; ++k) lens[i++] = (uint8_t) val;
				}
				if (0 == lens[256]) return false;		// no end-of-block code
				uint8_t dl[32] = {0};
				std::memcpy(dl, lens + hlit, hdist);
				if (!build(lens, hlit, LITLEN_BITS, m_litlen, sizeof(m_litlen) / 4, litlen_entry)) return false;
				if (!build(dl, hdist, DIST_BITS, m_dist, sizeof(m_dist) / 4, dist_entry)) return false;
			}

			// ---- the symbols of the block ---------------------------------------------------------------------------------------------
			uint32_t const litlen_mask = (1u << LITLEN_BITS) - 1, dist_mask = (1u << DIST_BITS) - 1;
			bool end_of_block = false;
			// Away from the ends of both buffers: eight-byte refills without a test, literals stored without a test (a refill
			// gives at least 56 bits: a literal / length code (15) + its extra bits (5) + a distance code (15) + its extra bits (13)
			// = 48, or up to three literals behind each other).
			while (!end_of_block && in_end - ip >= 16 && out_end - op >= 274)
			{
				{
					uint64_t w;
					std::memcpy(&w, ip, 8);
					bits |= w << n_bits;
					uint32_t const got = (63 - n_bits) >> 3;
					ip += got; n_bits += 8 * got;
				}
				uint32_t e = m_litlen[bits & litlen_mask];
				if (0 == (e & (F_SUB | F_LENGTH | F_END)) && (e & 15))
				{
					// up to three literals from one refill (most symbols of a BAM block are literals): 3 x 11 bits at most
					bits >>= e & 15; n_bits -= e & 15;
					*op++ = (uint8_t) (e >> 16);
					e = m_litlen[bits & litlen_mask];
					if (0 == (e & (F_SUB | F_LENGTH | F_END)) && (e & 15))
					{
						bits >>= e & 15; n_bits -= e & 15;
						*op++ = (uint8_t) (e >> 16);
						e = m_litlen[bits & litlen_mask];
						if (0 == (e & (F_SUB | F_LENGTH | F_END)) && (e & 15))
						{
							bits >>= e & 15; n_bits -= e & 15;
							*op++ = (uint8_t) (e >> 16);
							continue;
						}
					}
					if (n_bits < 48) continue;		// (what follows the literals may need 48 bits)
				}
				if (e & F_SUB)
				{
					bits >>= e & 15; n_bits -= e & 15;
					e = m_litlen[(e >> 16) + (uint32_t) (bits & ((1u << ((e >> 8) & 0xff)) - 1))];
				}
				if (0 == (e & 15)) return false;
				bits >>= e & 15; n_bits -= e & 15;
				if (0 == (e & (F_LENGTH | F_END))) { *op++ = (uint8_t) (e >> 16); continue; }
				if (e & F_END)
				{
					if (e & F_LENGTH) return false;		// symbols 286 / 287
					end_of_block = true;
					break;
				}
				uint32_t const length = (e >> 16) + take((e >> 8) & 0xff);
				uint32_t d = m_dist[bits & dist_mask];
				if (d & F_SUB)
				{
					bits >>= d & 15; n_bits -= d & 15;
					d = m_dist[(d >> 16) + (uint32_t) (bits & ((1u << ((d >> 8) & 0xff)) - 1))];
				}
				if (0 == (d & 15) || (d & F_END)) return false;
				bits >>= d & 15; n_bits -= d & 15;
				uint32_t const dist = (d >> 16) + take((d >> 8) & 0xff);
				if (dist > (std::size_t) (op - out)) return false;
				uint8_t const *from = op - dist;
				if (dist >= 8)
					for (uint32_t k = 0; k < length; k += 8) { uint64_t w; std::memcpy(&w, from + k, 8); std::memcpy(op + k, &w, 8); }		// (room for 258 + 7 checked above)
				else
					for (uint32_t k = 0; k < length; ++k) op[k] = from[k];		// (overlapping: byte by byte)
				op += length;
			}
			// Near the end of either buffer: the same, one symbol at a time, with every test.
			while (!end_of_block)
			{
				refill();
				uint32_t e = m_litlen[bits & litlen_mask];
				if (e & F_SUB)
				{
					take(e & 15);
					e = m_litlen[(e >> 16) + (uint32_t) (bits & ((1u << ((e >> 8) & 0xff)) - 1))];
				}
				if (0 == (e & 15)) return false;
				take(e & 15);
				if (0 == (e & (F_LENGTH | F_END)))
				{
					if (op == out_end) return false;
					*op++ = (uint8_t) (e >> 16);
					continue;
				}
				if (e & F_END)
				{
					if (e & F_LENGTH) return false;		// symbols 286 / 287
					break;
				}
				uint32_t const length = (e >> 16) + take((e >> 8) & 0xff);
				uint32_t d = m_dist[bits & dist_mask];
				if (d & F_SUB)
				{
					take(d & 15);
					d = m_dist[(d >> 16) + (uint32_t) (bits & ((1u << ((d >> 8) & 0xff)) - 1))];
				}
				if (0 == (d & 15) || (d & F_END)) return false;
				take(d & 15);
				uint32_t const dist = (d >> 16) + take((d >> 8) & 0xff);
				if (dist > (std::size_t) (op - out) || length > (std::size_t) (out_end - op)) return false;
				uint8_t const *from = op - dist;
				for (uint32_t k = 0; k < length; ++k) op[k] = from[k];
				op += length;
			}
			if (overrun > 8) return false;		// the block's symbols came out of bits that are not there
		}
		// Everything consumed that exists (up to the padding bits of the last byte), nothing that does not; the output is full.
		if (overrun > (n_bits >> 3)) return false;
		return op == out_end;
	}
};

}
