// inflate.hip — BGZF blocks inflated on the GPU, one wavefront per block (libpvc3_bgzf.so; C ABI in include/pvc3_bgzf.h).
//
// No reference code corresponds to this file: the reference reads BAM through seqan3 / zlib.  The BGZF reader of
// panvc3_project_alignments inflates every block of its input; with the output's blocks compressed on the GPU (deflate.hip)
// that was the largest share of the host's CPU time that was left.
//
// DEFLATE is decoded one symbol after the other, so a block is one wave's work and the wave runs it like a scalar program:
// the decoder's state (a bit window of 97 .. 128 bits, positions) is the same in every lane and lives in scalar registers,
// every branch is uniform.  What the 64 lanes add:
//   look-ups   lane k looks up the literal / length code and the distance code that would begin at bit k of the window; a scalar
//              walk then follows the codes that do begin somewhere (o, o + length of the code at o, ...) and takes the entries
//              out of the lanes with v_readlane: a run of literals and the matches between them for one trip to LDS;
//   copying    the last 32 KB of the output live in a ring in LDS (a match reaches no further back): a match is `length` bytes
//              copied by `length` lanes, overlapping copies included (src = j mod distance); what lies behind the ring goes to
//              global memory 16 KB at a time, in whole dwords;
//   staging    the compressed input comes through a 2 KB ring in LDS, 64 dwords of it at a time through the lanes of a register;
//   tables     canonical codes from the code lengths by all lanes: a 10-bit table for literals / lengths, 9 bits for distances;
//              the few longer codes are found by their length, first code and rank — no second-level tables.
// 39 KB of LDS per wave: four waves per CU, 1 024 blocks in flight.  The kernel is bound by the issue of its scalar
// instructions (about 35 per symbol), which is why every wave more per SIMD counts.
//
// Whatever the kernel does not accept (a code that does not exist, a distance before the block, an output longer than the
// ISIZE says, input that ends early) gets a status, and the caller inflates that block itself; the caller also checks every
// block's CRC-32.  So a flaw here costs time, not data.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/pvc3_bgzf.h"
#include "crc32_device.hpp"

extern thread_local char g_bgzf_error[256];		// deflate.hip

namespace {

constexpr unsigned WAVE = 64;
constexpr unsigned LL_BITS = 10, D_BITS = 9, CL_BITS = 7;
constexpr unsigned RING_WORDS = 512;		// two halves of 256 dwords
constexpr unsigned OUT_CAP = 65536;			// ISIZE at most
// The output's last 32 KB live in a ring in LDS: as far back as a match can reach.  A smaller ring (-DPVC3_INF_RING=8192: more
// blocks in flight per CU; a match that begins behind it reads the output in global memory, copy_far below) was measured and is
// slower by three on BAM records with binned qualities, whose matches lie anywhere in the 32 KB (tools/calib/inf_variants/ring.py,
// profiles/r04/inflate_ring.txt): 3.0 GB/s with 8 KB and 2 560 blocks in flight against 9.1 with 32 KB and 1 024.
#ifndef PVC3_INF_RING
#define PVC3_INF_RING 32768
#endif
constexpr unsigned OUT_RING = PVC3_INF_RING;
constexpr unsigned OUT_HALF = OUT_RING / 2;
static_assert(OUT_RING >= 2048 && OUT_RING <= 32768 && 0 == (OUT_RING & (OUT_RING - 1)), "the ring is a power of two that holds a match and a run of literals twice over");

enum : uint32_t
{
	ST_OK = 0, ST_BAD_BLOCK_TYPE = 1, ST_BAD_STORED = 2, ST_BAD_CODE_LENGTHS = 3, ST_BAD_SYMBOL = 4, ST_BAD_DISTANCE = 5, ST_OUTPUT_OVERFLOW = 6,
	ST_INPUT_OVERRUN = 7, ST_LENGTH_MISMATCH = 8, ST_TOO_LARGE = 9
};

struct code_info		// canonical decoding of the codes longer than the table's index
{
	uint32_t count[16], first[16], offset[16];
};

// t_ring: the single-kernel form keeps the output's last 32 KB here; the decoding kernel of the two-phase form (below) has no
// output window at all, and 7 KB of LDS instead of 39.
template <bool t_ring>
struct lds_t
{
	uint8_t out[t_ring ? OUT_RING : 16];
	uint32_t ring[RING_WORDS];
	uint16_t t_ll[1u << LL_BITS];		// symbol | length << 12; length 0: not in the table
	uint16_t t_d[1u << D_BITS];
	uint16_t t_cl[1u << CL_BITS];
	uint16_t sorted_ll[288], sorted_d[32];
	uint8_t lens[320];					// code lengths of the literal / length codes, then of the distance codes
	code_info ll, d;
	uint32_t flag;
};

__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

#define OUTB(i) L.out[(i) & (OUT_RING - 1)]

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// The compressed bytes as a stream of bits, least significant first: a window of 97 .. 128 bits in two 64-bit halves, the same in
// every lane (the compiler keeps it in scalar registers).  The ring's dwords are fetched 64 at a time, one per lane, and handed
// to the window with v_readlane, so that a refill is not a trip to LDS.
struct bit_reader
{
	uint32_t *ring;
	uint32_t const *src;		// the dword that holds the block's first byte
	uint32_t n_words;			// dwords that hold bytes of the block
	uint32_t next_word{};		// the next dword for the window
	uint32_t batch_first{};		// lane j holds dword batch_first + j
	uint32_t batch{};
	uint64_t lo{}, hi{};		// bits 0 .. 63 and 64 .. 127 of the window
	uint32_t valid{};
	uint32_t lead_bits{};

	__device__ __forceinline__ void stage(uint32_t first_word)		// half a ring of dwords from first_word on
	{
		for (uint32_t i = threadIdx.x; i < RING_WORDS / 2; i += WAVE)
		{
			uint32_t const w = first_word + i;
			ring[w & (RING_WORDS - 1)] = (w < n_words) ? src[w] : 0;
		}
		__syncthreads();
	}

	__device__ __forceinline__ void load_batch()
	{
		batch_first = next_word;
		batch = ring[(next_word + threadIdx.x) & (RING_WORDS - 1)];		// staged: the ring holds 256 dwords from next_word on at least
	}

	__device__ void start(uint8_t const *data, uint32_t n_bytes, uint32_t *ring_)
	{
		ring = ring_;
		uintptr_t const a = reinterpret_cast <uintptr_t>(data);
		src = reinterpret_cast <uint32_t const *>(a & ~uintptr_t(3));
		uint32_t const lead = (uint32_t) (a & 3);
		n_words = (lead + n_bytes + 3) / 4;
		next_word = 0; lo = 0; hi = 0; valid = 0; lead_bits = 8 * lead;
		__syncthreads();
		stage(0);
		stage(RING_WORDS / 2);
		load_batch();
		refill();
		drop(lead_bits);
		refill();
	}

	// At least 97 bits in the window afterwards (zeros behind the end of the input: consumed_bits() tells).
	__device__ __forceinline__ void refill()
	{
		while (valid <= 96)
		{
			if (next_word - batch_first >= WAVE) load_batch();
			uint64_t const w = (uint32_t) __builtin_amdgcn_readlane((int) batch, (int) (next_word - batch_first));
			if (valid >= 64) hi |= w << (valid - 64);
			else { lo |= w << valid; if (valid > 32) hi |= w >> (64 - valid); }
			valid += 32;
			++next_word;
			if (0 == (next_word & (RING_WORDS / 2 - 1)))		// entered a half: the one behind it is free for what follows this one
				stage(next_word + RING_WORDS / 2);
		}
	}

	__device__ __forceinline__ uint32_t low32() const { return (uint32_t) lo; }
	__device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t) lo & ((1u << n) - 1); }
	// n (at most 13) bits from bit `at` (below 96) on
	__device__ __forceinline__ uint32_t peek_at(uint32_t at, uint32_t n) const
	{
		uint64_t const v = (at >= 64) ? (hi >> (at - 64)) : (0 == at) ? lo : ((lo >> at) | (hi << (64 - at)));
		return (uint32_t) v & ((1u << n) - 1);
	}
	__device__ __forceinline__ void drop(uint32_t n)		// n below 128
	{
		valid -= n;
		if (n >= 64) { lo = hi; hi = 0; n -= 64; }
		if (n) { lo = (lo >> n) | (hi << (64 - n)); hi >>= n; }
	}
	__device__ __forceinline__ uint32_t take(uint32_t n) { uint32_t const v = peek(n); drop(n); return v; }
	// the 64 bits that begin at bit `k` (below 64) of the window
	__device__ __forceinline__ uint64_t from_bit(uint32_t k) const { return (0 == k) ? lo : ((lo >> k) | (hi << (64 - k))); }
	__device__ __forceinline__ uint64_t consumed_bits() const { return uint64_t(32) * next_word - valid - lead_bits; }
};

// Entries of the literal / length table: a literal is length << 8 | byte (0x0100 .. 0x0aff: one comparison tells), anything else
// 0x8000 | length << 9 | (symbol - 256); 0: the code is not in the table.
__device__ __forceinline__ uint16_t litlen_entry(uint32_t sym, uint32_t len) { return (uint16_t) ((sym < 256) ? ((len << 8) | sym) : (0x8000u | (len << 9) | (sym - 256))); }
__device__ __forceinline__ bool is_literal(uint32_t e) { return e - 0x100u < 0x0f00u; }
__device__ __forceinline__ void litlen_decode(uint32_t e, uint32_t &sym, uint32_t &len)		// of a non-zero entry
{
	if (is_literal(e)) { sym = e & 0xff; len = e >> 8; }
	else { sym = 256 + (e & 31); len = (e >> 9) & 15; }
}

// Canonical codes of lens[0 .. n) (RFC 1951 3.2.2) into a look-up table indexed by the next `bits` bits of the stream
// (entry = symbol | length << 12, or litlen_entry) and, for longer codes, the symbols sorted by (length, symbol) with the first code and the
// count of every length.  Returns false for an over-subscribed code.  Called by the whole wave.
template <typename t_entry, typename t_make>
__device__ bool build_table(uint8_t const *lens, uint32_t n, uint32_t bits, t_entry *table, uint16_t *sorted, code_info &ci, uint32_t *flag, t_make const &make_entry)
{
	unsigned const lane = threadIdx.x;
	for (uint32_t i = lane; i < (1u << bits); i += WAVE) table[i] = 0;
	if (lane < 16) ci.count[lane] = 0;
	__syncthreads();
	for (uint32_t s = lane; s < n; s += WAVE) if (lens[s]) atomicAdd(&ci.count[lens[s]], 1u);
	__syncthreads();
	if (0 == lane)
	{
		uint32_t code = 0, off = 0, left = 1;
		bool ok = true;
		for (uint32_t l = 1; l < 16; ++l)
		{
			left <<= 1;
			uint32_t const c = ci.count[l];
			if (c > left) ok = false;
			left -= (c > left) ? left : c;
			ci.first[l] = code; ci.offset[l] = off;
			code = (code + c) << 1;
			off += c;
		}
		ci.first[0] = 0; ci.offset[0] = 0;
		*flag = ok ? 1u : 0u;
	}
	__syncthreads();
	if (0 == *flag) return false;
	for (uint32_t s = lane; s < n; s += WAVE)
	{
		uint32_t const l = lens[s];
		if (0 == l) continue;
		uint32_t k = 0;
		for (uint32_t t = 0; t < s; ++t) k += (lens[t] == l);
		uint32_t const code = ci.first[l] + k;
		sorted[ci.offset[l] + k] = (uint16_t) s;
		if (l <= bits)
		{
			uint32_t const r = __brev(code) >> (32 - l);
			t_entry const e = (t_entry) make_entry(s, l);
			for (uint32_t j = r; j < (1u << bits); j += (1u << l)) table[j] = e;
		}
	}
	__syncthreads();
	return true;
}

// The symbol whose code (longer than the table's index) begins the bit buffer, by length; false if there is none.
__device__ __forceinline__ bool long_code(uint32_t low_bits, uint32_t from_len, uint16_t const *sorted, code_info const &ci, uint32_t &sym, uint32_t &len)
{
	for (uint32_t l = from_len; l < 16; ++l)
	{
		uint32_t const code = __brev(low_bits) >> (32 - l);		// the first l bits of the stream, most significant first
		uint32_t const k = code - uni(ci.first[l]);
		if (code >= uni(ci.first[l]) && k < uni(ci.count[l]))
		{
			sym = uni(sorted[uni(ci.offset[l]) + k]);
			len = l;
			return true;
		}
	}
	return false;
}

// ---- two phases (round 4) -----------------------------------------------------------------------------------------------------------
// What makes a block slow is the symbol-by-symbol decoding, a chain of dependent scalar instructions that one wave per SIMD cannot
// hide; what makes it expensive is the 32 KB window every block in flight holds in LDS while it is decoded — four blocks per CU,
// and nothing else of a caller's fits next to them.  The two do not need each other: decoding needs the bit stream and the code
// tables, the window is only needed to COPY.  So
//   k_inflate_decode  (this kernel with t_tokens)   decodes; a literal goes straight to its place in the output in global memory,
//                     a match becomes a token (position, length, distance) in global memory.  7 KB of LDS: twenty blocks per CU, whose
//                     waves hide each other's latency;
//   k_inflate_resolve takes a block's output — the literals are in place — into LDS whole, copies the matches in order, and stores
//                     it back: a few instructions per match, the window held for a fraction of the time.
// The single-kernel form stays (t_tokens = false; PVC3_BGZF_INFLATE_SINGLE=1 selects it) as the comparator.
typedef unsigned long long token_t;		// position | length << 17 | distance << 32
__device__ __forceinline__ token_t make_token(uint32_t pos, uint32_t length, uint32_t dist) { return (token_t) pos | ((token_t) length << 17) | ((token_t) dist << 32); }
// Tokens of the block whose bytes are out[off, off + len): a match makes at least three bytes, so floor(len / 3) tokens at most, and
// ranges of bytes that do not overlap get ranges of tokens that do not overlap.
__device__ __forceinline__ size_t token_base(uint64_t out_off) { return (size_t) ((out_off + 2) / 3); }

template <bool t_tokens>
__global__ __launch_bounds__(WAVE) void k_inflate_blocks(
	uint8_t const *__restrict__ in, pvc3_bgzf_block const *__restrict__ blocks, uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t n_blocks,
	token_t *__restrict__ tokens, uint32_t *__restrict__ n_tokens)
{
	__shared__ lds_t <!t_tokens> L;
	unsigned const lane = threadIdx.x;

	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		pvc3_bgzf_block const blk = blocks[b];
		uint32_t const want = blk.out_len;
		__syncthreads();		// the previous block's output has left LDS
		if (want > OUT_CAP) { if (0 == lane) status[b] = ST_TOO_LARGE; continue; }
		bit_reader br;
		br.start(in + blk.in_off, blk.in_len, L.ring);
		uint32_t pos = 0, st = ST_OK;
#ifdef PVC3_INF_COUNT
		uint32_t c_pass = 0, c_slow = 0, c_lit = 0, c_match = 0, c_tables = 0;
#endif
		// The output's last 32 KB live in LDS (a match reaches no further back); what lies behind goes to global memory half a
		// ring at a time, before the bytes that will take its place are written.
		uint32_t flushed = 0;
		uint8_t *const dst = out + blk.out_off;
		token_t *const tok = t_tokens ? (tokens + token_base(blk.out_off)) : nullptr;
		uint32_t n_tok = 0;
		auto store_region = [&](uint32_t start, uint32_t n) {		// output bytes start .. start + n from the ring (they do not wrap)
			uint8_t *g = dst + start;
			uint32_t const ro = start & (OUT_RING - 1);
			uint32_t const head = (uint32_t) ((4 - (reinterpret_cast <uintptr_t>(g) & 3)) & 3);
			uint32_t const h = (head < n) ? head : n;
			if (lane < h) g[lane] = L.out[ro + lane];
			uint32_t const n_dw = (n - h) / 4;
			uint32_t *g32 = reinterpret_cast <uint32_t *>(g + h);
			uint32_t const *lds32 = reinterpret_cast <uint32_t const *>(L.out);
			for (uint32_t i = lane; i < n_dw; i += WAVE)
			{
				uint32_t const x = ro + h + 4 * i;
				uint32_t const w0 = lds32[x >> 2], w1 = lds32[((x >> 2) + 1) & (OUT_RING / 4 - 1)];
				g32[i] = __builtin_amdgcn_alignbyte(w1, w0, x & 3);
			}
			uint32_t const tail = h + 4 * n_dw;
			if (tail + lane < n) g[tail + lane] = L.out[ro + tail + lane];
		};
		auto make_room = [&]() {		// before up to 322 bytes are written at pos
			if (t_tokens) return;
			if (pos + 322 > flushed + OUT_RING) { store_region(flushed, OUT_HALF); flushed += OUT_HALF; }
		};
		// (Rings smaller than a match's reach only.)  A match that begins behind the ring: a byte whose place in the ring the next 322
		// bytes may take is read from the output in global memory — make_room() has stored it; the fence orders the stores of all
		// lanes before the loads, which go past the first-level cache — the others from the ring.  Call after make_room().
		auto copy_far = [&](uint32_t from, uint32_t dist, uint32_t length) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
			for (uint32_t j = lane; j < length; j += WAVE)
			{
				uint32_t const s = from + ((dist >= length) ? j : (j % dist));
				uint8_t v;
				if (s + OUT_RING >= pos + 322) v = OUTB(s);
				else
				{
					// the aligned dword that holds the byte, past the first-level cache
					uintptr_t const a = reinterpret_cast <uintptr_t>(dst + s);
					uint32_t const w = __hip_atomic_load(reinterpret_cast <uint32_t const *>(a & ~uintptr_t(3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					v = (uint8_t) (w >> (8 * (a & 3)));
				}
				OUTB(pos + j) = v;
			}
		};

		for (bool last = false; !last && ST_OK == st; )
		{
			br.refill();
			last = 0 != br.take(1);
			uint32_t const type = br.take(2);
			if (0 == type)
			{
				// stored (RFC 1951 3.2.4): to the next byte boundary, LEN, NLEN, the bytes
				br.drop((uint32_t) ((0 - br.consumed_bits()) & 7));
				br.refill();
				uint32_t const len = br.take(16), nlen = br.take(16);
				if ((len ^ nlen) != 0xffff) { st = ST_BAD_STORED; break; }
				if (pos + len > want) { st = ST_OUTPUT_OVERFLOW; break; }
				for (uint32_t i = 0; i < len; ++i)
				{
					br.refill();
					uint32_t const v = br.take(8);
					make_room();
					if (0 == lane) { if (t_tokens) dst[pos] = (uint8_t) v; else OUTB(pos) = (uint8_t) v; }
					++pos;
				}
				continue;
			}
			if (3 == type) { st = ST_BAD_BLOCK_TYPE; break; }
			uint32_t n_ll, n_d;
			if (1 == type)
			{
				// fixed codes (3.2.6)
				n_ll = 288; n_d = 30;
				for (uint32_t s = lane; s < 288; s += WAVE) L.lens[s] = (uint8_t) ((s < 144) ? 8 : (s < 256) ? 9 : (s < 280) ? 7 : 8);
				if (lane < 30) L.lens[288 + lane] = 5;
				__syncthreads();
			}
			else
			{
				// dynamic codes (3.2.7)
				br.refill();
				n_ll = br.take(5) + 257; n_d = br.take(5) + 1;
				uint32_t const n_cl = br.take(4) + 4;
				uint64_t cl_lens = 0;		// three bits per symbol of the code length alphabet
				for (uint32_t i = 0; i < n_cl; ++i)
				{
					br.refill();
					cl_lens |= uint64_t(br.take(3)) << (3 * c_cl_order[i]);
				}
				if (n_ll > 286 || n_d > 30) { st = ST_BAD_CODE_LENGTHS; break; }
				{
					// the code length code: 19 symbols, one per lane
					uint32_t const l = (lane < 19) ? (uint32_t) ((cl_lens >> (3 * lane)) & 7) : 0;
					for (uint32_t i = lane; i < (1u << CL_BITS); i += WAVE) L.t_cl[i] = 0;
					__syncthreads();
					uint32_t code = 0, left = 1, mine = 0;
					bool ok = true;
					for (uint32_t k = 1; k < 8; ++k)
					{
						uint64_t const same = __ballot(l == k);
						uint32_t const c = (uint32_t) __popcll(same);
						left <<= 1;
						if (c > left) ok = false;
						left -= (c > left) ? left : c;
						if (l == k) mine = code + (uint32_t) __popcll(same & ((uint64_t(1) << lane) - 1));
						code = (code + c) << 1;
					}
					if (!ok) { st = ST_BAD_CODE_LENGTHS; break; }
					if (l)
					{
						uint32_t const r = __brev(mine) >> (32 - l);
						uint16_t const e = (uint16_t) (lane | (l << 12));
						for (uint32_t j = r; j < (1u << CL_BITS); j += (1u << l)) L.t_cl[j] = e;
					}
					__syncthreads();
				}
				uint32_t const n = n_ll + n_d;
				uint32_t i = 0, prev = 0;
				while (i < n && ST_OK == st)
				{
					br.refill();
					uint32_t const e = uni(L.t_cl[br.peek(CL_BITS)]);
					uint32_t const l = e >> 12, sym = e & 0xfff;
					if (0 == l) { st = ST_BAD_CODE_LENGTHS; break; }
					br.drop(l);
					uint32_t rep = 1, v = sym;
					if (16 == sym) { if (0 == i) { st = ST_BAD_CODE_LENGTHS; break; } rep = 3 + br.take(2); v = prev; }
					else if (17 == sym) { rep = 3 + br.take(3); v = 0; }
					else if (18 == sym) { rep = 11 + br.take(7); v = 0; }
					if (i + rep > n) { st = ST_BAD_CODE_LENGTHS; break; }
					for (uint32_t r = lane; r < rep; r += WAVE) L.lens[i + r] = (uint8_t) v;
					i += rep; prev = v;
				}
				if (ST_OK != st) break;
				__syncthreads();
				if (0 == L.lens[256]) { st = ST_BAD_CODE_LENGTHS; break; }		// no end-of-block code
			}
			uint8_t const *lens_d = L.lens + ((1 == type) ? 288 : n_ll);
#ifdef PVC3_INF_COUNT
			++c_tables;
#endif
			if (!build_table(L.lens, n_ll, LL_BITS, L.t_ll, L.sorted_ll, L.ll, &L.flag, [](uint32_t sy, uint32_t le) { return litlen_entry(sy, le); }) || !build_table(lens_d, n_d, D_BITS, L.t_d, L.sorted_d, L.d, &L.flag, [](uint32_t sy, uint32_t le) { return (uint16_t) (sy | (le << 12)); }))
			{ st = ST_BAD_CODE_LENGTHS; break; }

			// ---- the symbols -----------------------------------------------------------------------------------------------------------
			// One pass = one trip to LDS: lane k looks up the literal / length code and the distance code that would begin at bit
			// k of the window; then a scalar walk follows the codes that do begin somewhere — o, o + length of the code at o, ... —
			// taking the entries out of the lanes with v_readlane: a run of literals (literal i kept by lane i, one LDS store) and
			// the matches between them, until the offsets leave the lanes.  Codes longer than the tables' index (rare) and whatever
			// is wrong go through the one-token path behind the walk.
			for (bool end_of_block = false; !end_of_block && ST_OK == st; )
			{
#ifdef PVC3_INF_COUNT
				++c_pass;
#endif
				br.refill();
				uint64_t const mine = br.from_bit(lane);
				uint32_t const e_ll = L.t_ll[(uint32_t) mine & ((1u << LL_BITS) - 1)], e_d = L.t_d[(uint32_t) mine & ((1u << D_BITS) - 1)];
				uint32_t o = 0, n_lit = 0;
				int lits = 0;
				bool slow = false;
				auto flush_literals = [&]() {
					if (pos + n_lit > want) { st = ST_OUTPUT_OVERFLOW; return; }
					make_room();
#ifndef PVC3_INF_SKIP_LIT
					if (lane < n_lit) { if (t_tokens) dst[pos + lane] = (uint8_t) lits; else OUTB(pos + lane) = (uint8_t) lits; }
#endif
#ifdef PVC3_INF_COUNT
					c_lit += n_lit;
#endif
					pos += n_lit; n_lit = 0;
				};
				for (;;)
				{
					// the literals: one comparison each
					uint32_t e = 0;
					while (o < WAVE)
					{
						e = (uint32_t) __builtin_amdgcn_readlane((int) e_ll, (int) o);
						if (!is_literal(e)) break;
						lits = (lane == n_lit) ? (int) (e & 0xff) : lits;
						++n_lit;
						o += e >> 8;
					}
					if (o >= WAVE) break;		// out of the lanes
					if (0 == e) { slow = true; break; }
					uint32_t const l = (e >> 9) & 15, sym = 256 + (e & 31);
					if (256 == sym) { o += l; end_of_block = true; break; }
					if (sym > 285) { st = ST_BAD_SYMBOL; break; }
					uint32_t const ls = sym - 257;
					uint32_t const x = (ls < 8 || 28 == ls) ? 0 : ((ls - 4) >> 2);
					uint32_t const od = o + l + x;		// where the distance code begins
					if (od >= WAVE) break;					// not in the lanes: the next pass begins with this match
					uint32_t const ed = (uint32_t) __builtin_amdgcn_readlane((int) e_d, (int) od);
					uint32_t const ld = ed >> 12, ds = ed & 0xfff;
					if (0 == ld) { slow = true; break; }
					if (ds > 29) { st = ST_BAD_SYMBOL; break; }
					uint32_t const length = (ls < 8) ? (3 + ls) : (28 == ls) ? 258 : (3 + ((4 + ((ls - 4) & 3)) << x) + br.peek_at(o + l, x));
					uint32_t const xd = (ds < 4) ? 0 : ((ds >> 1) - 1);
					uint32_t const dist = (ds < 4) ? (1 + ds) : (1 + ((2 + (ds & 1)) << xd) + br.peek_at(od + ld, xd));
					flush_literals();
					if (ST_OK != st) break;
					if (dist > pos) { st = ST_BAD_DISTANCE; break; }
					if (pos + length > want) { st = ST_OUTPUT_OVERFLOW; break; }
					// `length` bytes from `dist` back; a copy that overlaps itself repeats its first `dist` bytes
					uint32_t const from = pos - dist;
					make_room();
#ifndef PVC3_INF_SKIP_COPY
					if (t_tokens) { if (0 == lane) tok[n_tok] = make_token(pos, length, dist); ++n_tok; }
					else if (OUT_RING < 32768 && from + OUT_RING < pos + 322) copy_far(from, dist, length);
					else if (dist >= length)
						for (uint32_t j = lane; j < length; j += WAVE) OUTB(pos + j) = OUTB(from + j);
					else if (1 == dist)
					{
						uint8_t const v = OUTB(from);
						for (uint32_t j = lane; j < length; j += WAVE) OUTB(pos + j) = v;
					}
					else
						for (uint32_t j = lane; j < length; j += WAVE) OUTB(pos + j) = OUTB(from + j % dist);
#else
					(void) from;
#endif
#ifdef PVC3_INF_COUNT
					++c_match;
#endif
					pos += length;
					o = od + ld + xd;
				}
				if (ST_OK == st) flush_literals();
				br.drop(o);
				if (!slow || ST_OK != st) continue;
#ifdef PVC3_INF_COUNT
				++c_slow;
#endif

				// one token the long way
				br.refill();
				uint32_t e = uni(L.t_ll[br.peek(LL_BITS)]);
				uint32_t l = 0, sym = 0;
				if (e) litlen_decode(e, sym, l);
				else if (!long_code(br.low32(), LL_BITS + 1, L.sorted_ll, L.ll, sym, l)) { st = ST_BAD_SYMBOL; break; }
				br.drop(l);
				if (sym < 256)
				{
					if (pos >= want) { st = ST_OUTPUT_OVERFLOW; break; }
					make_room();
					if (0 == lane) { if (t_tokens) dst[pos] = (uint8_t) sym; else OUTB(pos) = (uint8_t) sym; }
					++pos;
					continue;
				}
				if (256 == sym) { end_of_block = true; break; }
				if (sym > 285) { st = ST_BAD_SYMBOL; break; }
				uint32_t const ls = sym - 257;
				uint32_t length;
				if (ls < 8) length = 3 + ls;
				else if (28 == ls) length = 258;
				else { uint32_t const x = (ls - 4) >> 2; length = 3 + ((4 + ((ls - 4) & 3)) << x) + br.take(x); }
				br.refill();
				e = uni(L.t_d[br.peek(D_BITS)]);
				l = e >> 12; uint32_t ds = e & 0xfff;
				if (0 == l && !long_code(br.low32(), D_BITS + 1, L.sorted_d, L.d, ds, l)) { st = ST_BAD_SYMBOL; break; }
				br.drop(l);
				if (ds > 29) { st = ST_BAD_SYMBOL; break; }
				uint32_t dist;
				if (ds < 4) dist = 1 + ds;
				else { uint32_t const x = (ds >> 1) - 1; dist = 1 + ((2 + (ds & 1)) << x) + br.take(x); }
				if (dist > pos) { st = ST_BAD_DISTANCE; break; }
				if (pos + length > want) { st = ST_OUTPUT_OVERFLOW; break; }
				uint32_t const from = pos - dist;
				make_room();
				if (t_tokens) { if (0 == lane) tok[n_tok] = make_token(pos, length, dist); ++n_tok; }
				else if (OUT_RING < 32768 && from + OUT_RING < pos + 322) copy_far(from, dist, length);
				else if (dist >= length)
					for (uint32_t j = lane; j < length; j += WAVE) OUTB(pos + j) = OUTB(from + j);
				else
					for (uint32_t j = lane; j < length; j += WAVE) OUTB(pos + j) = OUTB(from + j % dist);
				pos += length;
			}
		}
		if (ST_OK == st && br.consumed_bits() > uint64_t(8) * blk.in_len) st = ST_INPUT_OVERRUN;
		if (ST_OK == st && pos != want) st = ST_LENGTH_MISMATCH;
		__syncthreads();
		if (ST_OK == st && !t_tokens)
		{
			while (want - flushed > OUT_HALF) { store_region(flushed, OUT_HALF); flushed += OUT_HALF; }
			store_region(flushed, want - flushed);
		}
#ifdef PVC3_INF_COUNT
		__threadfence();
		if (0 == lane && want >= 20) { uint32_t const c[5] = {c_pass, c_slow, c_lit, c_match, c_tables}; for (int k = 0; k < 20; ++k) dst[k] = (uint8_t) (c[k / 4] >> (8 * (k & 3))); }
#endif
		if (0 == lane) { status[b] = st; if (t_tokens) n_tokens[b] = (ST_OK == st) ? n_tok : 0; }
	}
}

// ---- the decoding kernel, second form ------------------------------------------------------------------------------------------------
// Table entries that hold everything a lane needs to decode its symbol without a branch.
//   literal / length: value (the literal's byte | the length's base) | code length << 9 | extra bits << 13 | kind << 16
//   distance:         base | code length << 15 | extra bits << 19 | 1 << 23
// 0: no such code in the table (longer than its index, or not a symbol of the format).
enum : uint32_t { K_NONE = 0, K_LITERAL = 1, K_MATCH = 2, K_END = 3 };
struct { __device__ __forceinline__ uint32_t operator()(uint32_t sym, uint32_t len) const
{
	if (sym < 256) return sym | (len << 9) | (K_LITERAL << 16);
	if (256 == sym) return (len << 9) | (K_END << 16);
	if (sym > 285) return 0;
	uint32_t const ls = sym - 257;
	uint32_t const x = (ls < 8 || 28 == ls) ? 0 : ((ls - 4) >> 2);
	uint32_t const base = (ls < 8) ? (3 + ls) : (28 == ls) ? 258 : (3 + ((4 + ((ls - 4) & 3)) << x));		// RFC 1951 3.2.5
	return base | (len << 9) | (x << 13) | (K_MATCH << 16);
} } const litlen_entry32;
struct { __device__ __forceinline__ uint32_t operator()(uint32_t ds, uint32_t len) const
{
	if (ds > 29) return 0;
	uint32_t const x = (ds < 4) ? 0 : ((ds >> 1) - 1);
	uint32_t const base = (ds < 4) ? (1 + ds) : (1 + ((2 + (ds & 1)) << x));
	return base | (len << 15) | (x << 19) | (1u << 23);
} } const dist_entry32;

struct lds_decode
{
	uint32_t ring[RING_WORDS];
	uint32_t t_ll[1u << LL_BITS];
	uint32_t t_d[1u << D_BITS];
	uint16_t t_cl[1u << CL_BITS];
	uint16_t sorted_ll[288], sorted_d[32];
	uint8_t lens[320];
	code_info ll, d;
	uint32_t flag;
};

// A symbol takes 10 + 5 + 9 + 13 = 37 bits at most through the tables and the window holds 97: the walk goes on while a symbol
// begins at bit 60 or before.
constexpr unsigned WALK_LAST = 60;

__global__ __launch_bounds__(WAVE) void k_inflate_decode(
	uint8_t const *__restrict__ in, pvc3_bgzf_block const *__restrict__ blocks, uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t n_blocks,
	token_t *__restrict__ tokens, uint32_t *__restrict__ n_tokens)
{
	__shared__ lds_decode L;
	unsigned const lane = threadIdx.x;

	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		pvc3_bgzf_block const blk = blocks[b];
		uint32_t const want = blk.out_len;
		if (want > OUT_CAP) { if (0 == lane) status[b] = ST_TOO_LARGE; continue; }
		bit_reader br;
		br.start(in + blk.in_off, blk.in_len, L.ring);
		uint32_t pos = 0, st = ST_OK;
		uint8_t *const dst = out + blk.out_off;
		token_t *const tok = tokens + token_base(blk.out_off);
		uint32_t n_tok = 0;

		for (bool last = false; !last && ST_OK == st; )
		{
			br.refill();
			last = 0 != br.take(1);
			uint32_t const type = br.take(2);
			if (0 == type)
			{
				// stored (RFC 1951 3.2.4): to the next byte boundary, LEN, NLEN, the bytes
				br.drop((uint32_t) ((0 - br.consumed_bits()) & 7));
				br.refill();
				uint32_t const len = br.take(16), nlen = br.take(16);
				if ((len ^ nlen) != 0xffff) { st = ST_BAD_STORED; break; }
				if (pos + len > want) { st = ST_OUTPUT_OVERFLOW; break; }
				for (uint32_t i = 0; i < len; ++i)
				{
					br.refill();
					uint32_t const v = br.take(8);
					if (0 == lane) dst[pos] = (uint8_t) v;
					++pos;
				}
				continue;
			}
			if (3 == type) { st = ST_BAD_BLOCK_TYPE; break; }
			uint32_t n_ll, n_d;
			if (1 == type)
			{
				// fixed codes (3.2.6)
				n_ll = 288; n_d = 30;
				for (uint32_t s = lane; s < 288; s += WAVE) L.lens[s] = (uint8_t) ((s < 144) ? 8 : (s < 256) ? 9 : (s < 280) ? 7 : 8);
				if (lane < 30) L.lens[288 + lane] = 5;
				__syncthreads();
			}
			else
			{
				// dynamic codes (3.2.7)
				br.refill();
				n_ll = br.take(5) + 257; n_d = br.take(5) + 1;
				uint32_t const n_cl = br.take(4) + 4;
				uint64_t cl_lens = 0;		// three bits per symbol of the code length alphabet
				for (uint32_t i = 0; i < n_cl; ++i)
				{
					br.refill();
					cl_lens |= uint64_t(br.take(3)) << (3 * c_cl_order[i]);
				}
				if (n_ll > 286 || n_d > 30) { st = ST_BAD_CODE_LENGTHS; break; }
				{
					// the code length code: 19 symbols, one per lane
					uint32_t const l = (lane < 19) ? (uint32_t) ((cl_lens >> (3 * lane)) & 7) : 0;
					for (uint32_t i = lane; i < (1u << CL_BITS); i += WAVE) L.t_cl[i] = 0;
					__syncthreads();
					uint32_t code = 0, left = 1, mine = 0;
					bool ok = true;
					for (uint32_t k = 1; k < 8; ++k)
					{
						uint64_t const same = __ballot(l == k);
						uint32_t const c = (uint32_t) __popcll(same);
						left <<= 1;
						if (c > left) ok = false;
						left -= (c > left) ? left : c;
						if (l == k) mine = code + (uint32_t) __popcll(same & ((uint64_t(1) << lane) - 1));
						code = (code + c) << 1;
					}
					if (!ok) { st = ST_BAD_CODE_LENGTHS; break; }
					if (l)
					{
						uint32_t const r = __brev(mine) >> (32 - l);
						uint16_t const e = (uint16_t) (lane | (l << 12));
						for (uint32_t j = r; j < (1u << CL_BITS); j += (1u << l)) L.t_cl[j] = e;
					}
					__syncthreads();
				}
				uint32_t const n = n_ll + n_d;
				uint32_t i = 0, prev = 0;
				while (i < n && ST_OK == st)
				{
					br.refill();
					uint32_t const e = uni(L.t_cl[br.peek(CL_BITS)]);
					uint32_t const l = e >> 12, sym = e & 0xfff;
					if (0 == l) { st = ST_BAD_CODE_LENGTHS; break; }
					br.drop(l);
					uint32_t rep = 1, v = sym;
					if (16 == sym) { if (0 == i) { st = ST_BAD_CODE_LENGTHS; break; } rep = 3 + br.take(2); v = prev; }
					else if (17 == sym) { rep = 3 + br.take(3); v = 0; }
					else if (18 == sym) { rep = 11 + br.take(7); v = 0; }
					if (i + rep > n) { st = ST_BAD_CODE_LENGTHS; break; }
					for (uint32_t r = lane; r < rep; r += WAVE) L.lens[i + r] = (uint8_t) v;
					i += rep; prev = v;
				}
				if (ST_OK != st) break;
				__syncthreads();
				if (0 == L.lens[256]) { st = ST_BAD_CODE_LENGTHS; break; }		// no end-of-block code
			}
			uint8_t const *lens_d = L.lens + ((1 == type) ? 288 : n_ll);
			if (!build_table(L.lens, n_ll, LL_BITS, L.t_ll, L.sorted_ll, L.ll, &L.flag, litlen_entry32) || !build_table(lens_d, n_d, D_BITS, L.t_d, L.sorted_d, L.d, &L.flag, dist_entry32))
			{ st = ST_BAD_CODE_LENGTHS; break; }

			// ---- the symbols -----------------------------------------------------------------------------------------------------------
			// One pass: lane k decodes the symbol that would begin at bit k of the window WHOLE — the literal, or the length with
			// its extra bits, the distance code behind them (a second look-up) and its extra bits: 37 bits at most, and the window
			// holds 97 — and says how many bits it takes and how many bytes it makes.  A scalar walk then only follows the chain
			// (o, o + bits of the symbol at o, ...: one v_readlane and a few scalar instructions per symbol) and hands every symbol
			// on it the place of its output; after the walk the lanes on the chain store — a literal its byte, a match its token.
			// Codes longer than the tables' index (rare) go through the one-symbol path behind the walk.
			for (bool end_of_block = false; !end_of_block && ST_OK == st; )
			{
				br.refill();
				uint64_t const mine = br.from_bit(lane);
				uint32_t const e = L.t_ll[(uint32_t) mine & ((1u << LL_BITS) - 1)];
				uint32_t const l = (e >> 9) & 15, x = (e >> 13) & 7, kind = e >> 16;
				uint32_t const length = (e & 0x1ff) + __builtin_amdgcn_ubfe((uint32_t) (mine >> l), 0, x);		// (a literal's byte, for a literal)
				uint32_t const behind = (uint32_t) (mine >> (l + x));
				uint32_t const ed = L.t_d[behind & ((1u << D_BITS) - 1)];
				uint32_t const ld = (ed >> 15) & 15, xd = (ed >> 19) & 15;
				uint32_t const dist = (ed & 0x7fff) + __builtin_amdgcn_ubfe(behind >> ld, 0, xd);
				bool const is_match = K_MATCH == kind && 0 != (ed >> 23);
				// bits | end of block << 7 | bytes << 8; no bits: the walk stops here
				uint32_t const packed = (K_LITERAL == kind) ? (l | (1u << 8)) : is_match ? ((l + x + ld + xd) | (length << 8)) : (K_END == kind) ? (l | 0x80u) : 0u;
				uint32_t o = 0, made = 0;
				uint64_t chain = 0;
				uint32_t place = 0;
				bool slow = false;
				while (o <= WALK_LAST)
				{
					uint32_t const p = (uint32_t) __builtin_amdgcn_readlane((int) packed, (int) o);
					if (0 == (p & 0x7f)) { slow = true; break; }
					place = (lane == o) ? made : place;
					chain |= uint64_t(1) << o;
					made += p >> 8;
					o += p & 0x7f;
					if (p & 0x80) { end_of_block = true; break; }
				}
				if (pos + made > want) { st = ST_OUTPUT_OVERFLOW; break; }
				bool const on_chain = 0 != ((chain >> lane) & 1);
				uint32_t const at = pos + place;
				if (on_chain && K_LITERAL == kind) dst[at] = (uint8_t) e;
				uint64_t const matches = __ballot(on_chain && is_match);
				if (matches)
				{
					if (__ballot(on_chain && is_match && dist > at)) { st = ST_BAD_DISTANCE; break; }
					if (on_chain && is_match)
						tok[n_tok + __builtin_amdgcn_mbcnt_hi((uint32_t) (matches >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) matches, 0))] = make_token(at, length, dist);
					n_tok += (uint32_t) __popcll(matches);
				}
				pos += made;
				br.drop(o);
				if (!slow) continue;

				// one symbol the long way: a code longer than the table's index, or one that does not exist
				br.refill();
				uint32_t e1 = uni(L.t_ll[br.peek(LL_BITS)]);
				if (0 == e1)
				{
					uint32_t sym = 0, l1 = 0;
					if (!long_code(br.low32(), LL_BITS + 1, L.sorted_ll, L.ll, sym, l1)) { st = ST_BAD_SYMBOL; break; }
					e1 = litlen_entry32(sym, l1);
					if (0 == e1) { st = ST_BAD_SYMBOL; break; }
				}
				br.drop((e1 >> 9) & 15);
				uint32_t const kind1 = e1 >> 16;
				if (K_LITERAL == kind1)
				{
					if (pos >= want) { st = ST_OUTPUT_OVERFLOW; break; }
					if (0 == lane) dst[pos] = (uint8_t) e1;
					++pos;
					continue;
				}
				if (K_END == kind1) { end_of_block = true; break; }
				uint32_t const length1 = (e1 & 0x1ff) + br.take((e1 >> 13) & 7);
				br.refill();
				uint32_t ed1 = uni(L.t_d[br.peek(D_BITS)]);
				if (0 == ed1)
				{
					uint32_t ds = 0, l1 = 0;
					if (!long_code(br.low32(), D_BITS + 1, L.sorted_d, L.d, ds, l1)) { st = ST_BAD_SYMBOL; break; }
					ed1 = dist_entry32(ds, l1);
					if (0 == ed1) { st = ST_BAD_SYMBOL; break; }
				}
				br.drop((ed1 >> 15) & 15);
				uint32_t const dist1 = (ed1 & 0x7fff) + br.take((ed1 >> 19) & 15);
				if (dist1 > pos) { st = ST_BAD_DISTANCE; break; }
				if (pos + length1 > want) { st = ST_OUTPUT_OVERFLOW; break; }
				if (0 == lane) tok[n_tok] = make_token(pos, length1, dist1);
				++n_tok;
				pos += length1;
			}
		}
		if (ST_OK == st && br.consumed_bits() > uint64_t(8) * blk.in_len) st = ST_INPUT_OVERRUN;
		if (ST_OK == st && pos != want) st = ST_LENGTH_MISMATCH;
		if (0 == lane) { status[b] = st; n_tokens[b] = (ST_OK == st) ? n_tok : 0; }
		__syncthreads();		// the tables are the next block's from here
	}
}

// The second phase: a block's output, literals in place, into LDS whole; its matches copied in the order of the stream (a match may
// copy what an earlier one made, and may overlap its own output); then back.  Tokens come 64 at a time through the lanes.
constexpr unsigned RESOLVE_BYTES = 65536;
__global__ __launch_bounds__(WAVE) void k_inflate_resolve(pvc3_bgzf_block const *__restrict__ blocks, uint8_t *__restrict__ out, uint32_t const *__restrict__ status, uint32_t n_blocks,
	token_t const *__restrict__ tokens, uint32_t const *__restrict__ n_tokens)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[RESOLVE_BYTES + 16];
	unsigned const lane = threadIdx.x;
	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		uint32_t const n = uni(n_tokens[b]);
		if (0 == n || ST_OK != uni(status[b])) continue;		// no match: the literals are the output
		pvc3_bgzf_block const blk = blocks[b];
		uint32_t const want = blk.out_len;
		uint8_t *const dst = out + blk.out_off;
		token_t const *tok = tokens + token_base(blk.out_off);
		__syncthreads();
		// in: bytes up to a dword boundary of the address, whole dwords, the rest.  Output byte i lies at B[i] = buf[a + i] with a
		// chosen so that the dwords are dwords in LDS too.
		uint32_t const head = (uint32_t) ((4 - (reinterpret_cast <uintptr_t>(dst) & 3)) & 3);
		uint32_t const h = (head < want) ? head : want;
		uint8_t *const B = buf + ((4 - h) & 3);
		if (lane < h) B[lane] = dst[lane];
		uint32_t const n_dw = (want - h) / 4;
		{
			uint32_t const *g32 = reinterpret_cast <uint32_t const *>(dst + h);
			uint32_t *l32 = reinterpret_cast <uint32_t *>(B + h);
			for (uint32_t i = lane; i < n_dw; i += WAVE) l32[i] = g32[i];
		}
		uint32_t const tail = h + 4 * n_dw;
		if (tail + lane < want) B[tail + lane] = dst[tail + lane];
		__syncthreads();
		for (uint32_t t0 = 0; t0 < n; t0 += WAVE)
		{
			token_t const mine = (t0 + lane < n) ? tok[t0 + lane] : 0;
			uint32_t const m_lo = (uint32_t) mine, m_hi = (uint32_t) (mine >> 32);
			uint32_t const cnt = (n - t0 < WAVE) ? (n - t0) : WAVE;
			for (uint32_t k = 0; k < cnt; ++k)
			{
				uint32_t const lo = (uint32_t) __builtin_amdgcn_readlane((int) m_lo, (int) k), dist = (uint32_t) __builtin_amdgcn_readlane((int) m_hi, (int) k);
				uint32_t const pos = lo & 0x1ffffu, length = lo >> 17, from = pos - dist;
				if (dist >= length)
					for (uint32_t j = lane; j < length; j += WAVE) B[pos + j] = B[from + j];
				else if (1 == dist)
				{
					uint8_t const v = B[from];
					for (uint32_t j = lane; j < length; j += WAVE) B[pos + j] = v;
				}
				else
					for (uint32_t j = lane; j < length; j += WAVE) B[pos + j] = B[from + j % dist];
				__builtin_amdgcn_wave_barrier();		// (a wave's LDS operations are executed in order; the compiler must not move the next match's reads up)
			}
		}
		__syncthreads();
		// out
		if (lane < h) dst[lane] = B[lane];
		{
			uint32_t *g32 = reinterpret_cast <uint32_t *>(dst + h);
			uint32_t const *l32 = reinterpret_cast <uint32_t const *>(B + h);
			for (uint32_t i = lane; i < n_dw; i += WAVE) g32[i] = l32[i];
		}
		if (tail + lane < want) dst[tail + lane] = B[tail + lane];
	}
}

// The second phase, second form.  What the first form spends per match is the latency of LDS — read, wait, write — 64 lanes wide for
// a dozen bytes.  Most matches of a stretch do not depend on each other (they copy from a record or two back), so: the matches come
// 64 at a time, one per lane; a lane finds the earlier matches of the 64 whose output it reads (their places ascend: two binary
// searches through ds_bpermute, and none at all when every source lies before the first place of the 64 — the usual case); then, in
// rounds, every match whose sources are all written copies at once — each lane its own, five aligned dwords in and up to 16
// bytes out — and the few that are long, or repeat a pattern of their own output, go one at a time across the lanes as before.
constexpr unsigned RESOLVE_SLACK = 48;		// the block may begin 15 bytes in, and a lane reads 19 bytes past its match's first
constexpr unsigned LANE_COPY_MAX = 16;
__global__ __launch_bounds__(WAVE) void k_inflate_resolve_lanes(pvc3_bgzf_block const *__restrict__ blocks, uint8_t *__restrict__ out, uint32_t const *__restrict__ status, uint32_t n_blocks,
	token_t const *__restrict__ tokens, uint32_t const *__restrict__ n_tokens)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[RESOLVE_BYTES + RESOLVE_SLACK];
	unsigned const lane = threadIdx.x;
	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		uint32_t const n = uni(n_tokens[b]);
		if (0 == n || ST_OK != uni(status[b])) continue;		// no match: the literals are the output
		pvc3_bgzf_block const blk = blocks[b];
		uint32_t const want = blk.out_len;
		uint8_t *const dst = out + blk.out_off;
		token_t const *tok = tokens + token_base(blk.out_off);
		token_t mine = (lane < n) ? tok[lane] : 0;
		__syncthreads();
		// in: bytes up to a 16-byte boundary of the address, whole 16 bytes, the rest.  Output byte i lies at B[i] = buf[a + i] with a
		// chosen so that the boundary is one in LDS too.
		uint32_t const head = (uint32_t) ((16 - (reinterpret_cast <uintptr_t>(dst) & 15)) & 15);
		uint32_t const h = (head < want) ? head : want;
		uint32_t const b_off = (16 - head) & 15;
		uint8_t *const B = buf + b_off;
		if (lane < h) B[lane] = dst[lane];
		uint32_t const n_q = (want - h) / 16;
		{
			uint4 const *g = reinterpret_cast <uint4 const *>(dst + h);
			uint4 *l = reinterpret_cast <uint4 *>(B + h);
			for (uint32_t i = lane; i < n_q; i += WAVE) l[i] = g[i];
		}
		uint32_t const tail = h + 16 * n_q;
		if (tail + lane < want) B[tail + lane] = dst[tail + lane];
		__syncthreads();
		uint32_t const *const words = reinterpret_cast <uint32_t const *>(buf);
		for (uint32_t t0 = 0; t0 < n; t0 += WAVE)
		{
			bool const valid = t0 + lane < n;
			uint32_t const m_lo = (uint32_t) mine, dist = (uint32_t) (mine >> 32);
			uint32_t const pos = valid ? (m_lo & 0x1ffffu) : 0xffffffffu, length = valid ? (m_lo >> 17) : 0, from = pos - dist;
			uint32_t const end = valid ? (pos + length) : 0xffffffffu;
			uint32_t const src_end = from + ((dist < length) ? dist : length);
			if (t0 + WAVE < n) mine = (t0 + WAVE + lane < n) ? tok[t0 + WAVE + lane] : 0;		// the next 64 on their way
			// the earlier matches of these 64 whose output this one reads: q .. p - 1, as a mask
			uint64_t needs = 0;
			uint32_t const first_pos = (uint32_t) __builtin_amdgcn_readfirstlane((int) pos);
			if (__ballot(valid && src_end > first_pos))
			{
				uint32_t q = 0, p_ = 0;		// how many ends lie at or before my source's first byte; how many places lie before my source's end
				for (uint32_t step = 32; step; step >>= 1)
				{
					uint32_t const e_q = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (4 * (q + step - 1)), (int) end);
					uint32_t const p_p = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (4 * (p_ + step - 1)), (int) pos);
					if (e_q <= from) q += step;
					if (p_p < src_end) p_ += step;
				}
				uint32_t const hi = (p_ < lane) ? p_ : lane;
				if (valid && q < hi) needs = ((uint64_t(1) << hi) - 1) & ~((uint64_t(1) << q) - 1);
			}
			bool const by_lane = valid && length <= LANE_COPY_MAX && (dist >= length || 1 == dist);
			uint64_t undone = __ballot(valid);
			while (undone)
			{
				bool const ready = 0 != ((undone >> lane) & 1) && 0 == (undone & needs);
				uint64_t const round = __ballot(ready);
				if (__ballot(ready && by_lane))
				{
					if (ready && by_lane)
					{
						uint32_t const a = b_off + from;
						uint32_t const *w = words + (a >> 2);
						uint32_t const w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
						uint32_t d[4] = {__builtin_amdgcn_alignbyte(w1, w0, a & 3), __builtin_amdgcn_alignbyte(w2, w1, a & 3), __builtin_amdgcn_alignbyte(w3, w2, a & 3), __builtin_amdgcn_alignbyte(w4, w3, a & 3)};
						if (1 == dist) d[0] = d[1] = d[2] = d[3] = (d[0] & 0xff) * 0x01010101u;
#pragma unroll
						for (uint32_t j = 0; j < LANE_COPY_MAX; ++j)
							if (j < length) B[pos + j] = (uint8_t) (d[j / 4] >> (8 * (j % 4)));
					}
				}
				for (uint64_t rest = round & ~__ballot(ready && by_lane); rest; rest &= rest - 1)
				{
					uint32_t const k = (uint32_t) __builtin_ctzll(rest);
					uint32_t const k_lo = (uint32_t) __builtin_amdgcn_readlane((int) m_lo, (int) k), k_dist = (uint32_t) __builtin_amdgcn_readlane((int) dist, (int) k);
					uint32_t const k_pos = k_lo & 0x1ffffu, k_len = k_lo >> 17, k_from = k_pos - k_dist;
					if (k_dist >= k_len)
						for (uint32_t j = lane; j < k_len; j += WAVE) B[k_pos + j] = B[k_from + j];
					else if (1 == k_dist)
					{
						uint8_t const v = B[k_from];
						for (uint32_t j = lane; j < k_len; j += WAVE) B[k_pos + j] = v;
					}
					else
						for (uint32_t j = lane; j < k_len; j += WAVE) B[k_pos + j] = B[k_from + j % k_dist];
				}
				undone &= ~round;
				__builtin_amdgcn_wave_barrier();		// (a wave's LDS operations are executed in order; the compiler must not move the next round's reads up)
			}
		}
		__syncthreads();
		// out
		if (lane < h) dst[lane] = B[lane];
		{
			uint4 *g = reinterpret_cast <uint4 *>(dst + h);
			uint4 const *l = reinterpret_cast <uint4 const *>(B + h);
			for (uint32_t i = lane; i < n_q; i += WAVE) g[i] = l[i];
		}
		if (tail + lane < want) dst[tail + lane] = B[tail + lane];
	}
}

// ---- host ------------------------------------------------------------------------------------------------------------------------

int fail(int code, char const *what, hipError_t e = hipSuccess)
{
	if (hipSuccess != e) std::snprintf(g_bgzf_error, sizeof(g_bgzf_error), "%s: %s", what, hipGetErrorString(e));
	else std::snprintf(g_bgzf_error, sizeof(g_bgzf_error), "%s", what);
	return code;
}

enum { E_INVALID = -1, E_NOMEM = -2, E_HIP = -3, E_NO_DEVICE = -4, E_STATE = -6 };

#define HIP_TRY(expr, what) do { hipError_t const e_ = (expr); if (hipSuccess != e_) return fail(E_HIP, what, e_); } while (0)

constexpr size_t IN_SLACK = 4096;		// behind a slot's input: the kernel reads whole dwords

}

struct pvc3_bgzf_inflater
{
	int device{};
	size_t max_in{}, max_out{};
	uint32_t max_blocks{}, n_slots{}, grid_max{}, grid_full{};
	hipStream_t stream{};
	uint8_t *d_in{}, *d_out{};
	pvc3_bgzf_block *d_blocks{};
	uint32_t *d_status{};
	// the two-phase form: the matches of every block between the two kernels
	token_t *d_tokens{};
	size_t tokens_cap{};				// in tokens
	uint32_t *d_ntok{};
	uint32_t grid_resolve{};
	bool single{};						// PVC3_BGZF_INFLATE_SINGLE=1: the one-kernel form
	bool decode_v1{}, resolve_v1{};		// PVC3_BGZF_INFLATE_DECODE=1, PVC3_BGZF_INFLATE_RESOLVE=1: the first forms of the two phases
	// the device-resident form (pvc3_bgzf_inflater_submit_device): where the blocks' bytes went and their checksums
	uint64_t *d_off{};
	uint32_t *d_lens{}, *d_crc{};
	pvc3_crc::tables *d_crc_tables{};
	struct slot
	{
		uint8_t *h_in{}, *h_out{};
		pvc3_bgzf_block *h_blocks{};
		uint32_t *h_status{};
		uint64_t *h_off{};
		uint32_t *h_lens{}, *h_crc{};
		hipEvent_t start{}, stop{}, done{};
		uint32_t n{};
		bool busy{};
	} slots[4];
};

namespace {

// The kernels of one group of blocks on the handle's stream: the two phases (tokens for out_bytes of output), or the single kernel.
int launch_inflate(pvc3_bgzf_inflater *h, uint8_t *d_out, size_t out_bytes, uint32_t n_blocks, hipEvent_t start, hipEvent_t stop)
{
	uint32_t const grid = (n_blocks < h->grid_max) ? n_blocks : h->grid_max;
	if (!h->single)
	{
		size_t const need = out_bytes / 3 + n_blocks + 64;
		if (need > h->tokens_cap)
		{
			HIP_TRY(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
			if (h->d_tokens) { (void) hipFree(h->d_tokens); h->d_tokens = nullptr; h->tokens_cap = 0; }
			HIP_TRY(hipMalloc((void **) &h->d_tokens, (need + need / 8) * sizeof(token_t)), "hipMalloc");
			h->tokens_cap = need + need / 8;
		}
	}
	HIP_TRY(hipEventRecord(start, h->stream), "hipEventRecord");
	if (h->single)
		hipLaunchKernelGGL(k_inflate_blocks <false>, dim3(grid), dim3(WAVE), 0, h->stream, h->d_in, h->d_blocks, d_out, h->d_status, n_blocks, (token_t *) nullptr, (uint32_t *) nullptr);
	else
	{
		if (h->decode_v1) hipLaunchKernelGGL(k_inflate_blocks <true>, dim3(grid), dim3(WAVE), 0, h->stream, h->d_in, h->d_blocks, d_out, h->d_status, n_blocks, h->d_tokens, h->d_ntok);
		else hipLaunchKernelGGL(k_inflate_decode, dim3(grid), dim3(WAVE), 0, h->stream, h->d_in, h->d_blocks, d_out, h->d_status, n_blocks, h->d_tokens, h->d_ntok);
		uint32_t const grid_b = (n_blocks < h->grid_resolve) ? n_blocks : h->grid_resolve;
		if (h->resolve_v1) hipLaunchKernelGGL(k_inflate_resolve, dim3(grid_b), dim3(WAVE), 0, h->stream, h->d_blocks, d_out, h->d_status, n_blocks, h->d_tokens, h->d_ntok);
		else hipLaunchKernelGGL(k_inflate_resolve_lanes, dim3(grid_b), dim3(WAVE), 0, h->stream, h->d_blocks, d_out, h->d_status, n_blocks, h->d_tokens, h->d_ntok);
	}
	HIP_TRY(hipGetLastError(), "k_inflate_blocks");
	HIP_TRY(hipEventRecord(stop, h->stream), "hipEventRecord");
	return 0;
}

}

extern "C" {

void pvc3_bgzf_inflater_destroy(pvc3_bgzf_inflater *h)
{
	if (!h) return;
	(void) hipSetDevice(h->device);
	if (h->stream) (void) hipStreamSynchronize(h->stream);
	for (auto &s : h->slots)
	{
		if (s.h_in) (void) hipHostFree(s.h_in);
		if (s.h_out) (void) hipHostFree(s.h_out);
		if (s.h_blocks) (void) hipHostFree(s.h_blocks);
		if (s.h_status) (void) hipHostFree(s.h_status);
		if (s.h_off) (void) hipHostFree(s.h_off);
		if (s.h_lens) (void) hipHostFree(s.h_lens);
		if (s.h_crc) (void) hipHostFree(s.h_crc);
		if (s.start) (void) hipEventDestroy(s.start);
		if (s.stop) (void) hipEventDestroy(s.stop);
		if (s.done) (void) hipEventDestroy(s.done);
	}
	if (h->d_in) (void) hipFree(h->d_in);
	if (h->d_out) (void) hipFree(h->d_out);
	if (h->d_blocks) (void) hipFree(h->d_blocks);
	if (h->d_status) (void) hipFree(h->d_status);
	for (void *p : {(void *) h->d_off, (void *) h->d_lens, (void *) h->d_crc, (void *) h->d_crc_tables, (void *) h->d_tokens, (void *) h->d_ntok}) if (p) (void) hipFree(p);
	if (h->stream) (void) hipStreamDestroy(h->stream);
	delete h;
}

int pvc3_bgzf_inflater_create(int device, size_t max_in_bytes, size_t max_out_bytes, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf_inflater **out)
{
	if (!out || 0 == max_in_bytes || 0 == max_out_bytes || 0 == max_blocks || 0 == n_slots || n_slots > 4)
		return fail(E_INVALID, "pvc3_bgzf_inflater_create: bad argument");
	*out = nullptr;
	int count = 0;
	if (hipSuccess != hipGetDeviceCount(&count) || device < 0 || device >= count) return fail(E_NO_DEVICE, "pvc3_bgzf_inflater_create: no such GPU");
	HIP_TRY(hipSetDevice(device), "hipSetDevice");
	pvc3_bgzf_inflater *h = new (std::nothrow) pvc3_bgzf_inflater;
	if (!h) return fail(E_NOMEM, "out of memory");
	h->device = device; h->max_in = max_in_bytes; h->max_out = max_out_bytes; h->max_blocks = max_blocks; h->n_slots = n_slots;
	int rc = 0;
	auto check = [&](hipError_t e, char const *what) { if (0 == rc && hipSuccess != e) rc = fail((hipErrorOutOfMemory == e) ? E_NOMEM : E_HIP, what, e); };
	hipDeviceProp_t prop;
	check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
	if (0 == rc)
	{
		int per_cu = 0;
		char const *e = std::getenv("PVC3_BGZF_INFLATE_SINGLE");
		h->single = e && '1' == e[0];
		char const *e1 = std::getenv("PVC3_BGZF_INFLATE_DECODE"), *e2 = std::getenv("PVC3_BGZF_INFLATE_RESOLVE");
		h->decode_v1 = e1 && '1' == e1[0];
		h->resolve_v1 = e2 && '1' == e2[0];
		if (h->single) check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_blocks <false>, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		else if (h->decode_v1) check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_blocks <true>, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		else check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_decode, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		if (per_cu < 1) per_cu = 1;
		h->grid_max = (uint32_t) prop.multiProcessorCount * (uint32_t) per_cu;
		h->grid_full = h->grid_max;
		int per_cu_b = 0;
		check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_inflate_resolve, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		h->grid_resolve = (uint32_t) prop.multiProcessorCount * (uint32_t) ((per_cu_b < 1) ? 1 : per_cu_b);
	}
	check(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate");
	if (0 == rc) check(hipMalloc((void **) &h->d_in, max_in_bytes + IN_SLACK), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_out, max_out_bytes), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_blocks, max_blocks * sizeof(pvc3_bgzf_block)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_status, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_ntok, max_blocks * sizeof(uint32_t)), "hipMalloc");
	for (uint32_t i = 0; i < n_slots && 0 == rc; ++i)
	{
		auto &s = h->slots[i];
		check(hipHostMalloc((void **) &s.h_in, max_in_bytes + IN_SLACK, hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_out, max_out_bytes, hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_blocks, max_blocks * sizeof(pvc3_bgzf_block), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_status, max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipEventCreate(&s.start), "hipEventCreate");
		if (0 == rc) check(hipEventCreate(&s.stop), "hipEventCreate");
		if (0 == rc) check(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
	}
	if (rc) { char keep[sizeof(g_bgzf_error)]; std::memcpy(keep, g_bgzf_error, sizeof(keep)); pvc3_bgzf_inflater_destroy(h); std::memcpy(g_bgzf_error, keep, sizeof(keep)); return rc; }
	*out = h;
	return 0;
}

uint32_t pvc3_bgzf_inflater_blocks_in_flight(pvc3_bgzf_inflater const *h) { return h ? h->grid_max : 0; }

// Fewer blocks at a time than the chip's LDS holds (0: as many as it holds again): a caller whose other kernels run next to the
// inflater's leaves them room — a block being inflated keeps 39 KB of its CU's LDS for milliseconds.
int pvc3_bgzf_inflater_limit_in_flight(pvc3_bgzf_inflater *h, uint32_t blocks)
{
	if (!h) return fail(E_INVALID, "pvc3_bgzf_inflater_limit_in_flight: null argument");
	if (0 == blocks || blocks > h->grid_full) blocks = h->grid_full;
	h->grid_max = blocks;
	return 0;
}
uint8_t *pvc3_bgzf_inflater_input(pvc3_bgzf_inflater *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].h_in : nullptr; }
uint8_t *pvc3_bgzf_inflater_output(pvc3_bgzf_inflater *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].h_out : nullptr; }
pvc3_bgzf_block *pvc3_bgzf_inflater_blocks(pvc3_bgzf_inflater *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].h_blocks : nullptr; }

int pvc3_bgzf_inflater_submit(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t n_blocks, size_t in_bytes, size_t out_bytes)
{
	if (!h || slot >= h->n_slots || 0 == n_blocks || n_blocks > h->max_blocks || in_bytes > h->max_in || out_bytes > h->max_out)
		return fail(E_INVALID, "pvc3_bgzf_inflater_submit: bad argument");
	auto &s = h->slots[slot];
	if (s.busy) return fail(E_STATE, "pvc3_bgzf_inflater_submit: the slot has not been waited for");
	for (uint32_t i = 0; i < n_blocks; ++i)
	{
		pvc3_bgzf_block const &b = s.h_blocks[i];
		if (b.in_off > in_bytes || b.in_len > in_bytes - b.in_off || b.out_off > out_bytes || b.out_len > out_bytes - b.out_off)
			return fail(E_INVALID, "pvc3_bgzf_inflater_submit: a block outside the slot's input or output");
	}
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	HIP_TRY(hipMemcpyAsync(h->d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(h->d_blocks, s.h_blocks, n_blocks * sizeof(pvc3_bgzf_block), hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	if (int const rc = launch_inflate(h, h->d_out, h->max_out, n_blocks, s.start, s.stop)) return rc;
	HIP_TRY(hipMemcpyAsync(s.h_status, h->d_status, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	if (out_bytes) HIP_TRY(hipMemcpyAsync(s.h_out, h->d_out, out_bytes, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.done, h->stream), "hipEventRecord");
	s.n = n_blocks; s.busy = true;
	return 0;
}

// The device-resident form: the slot's input is uploaded as before, but the blocks' bytes go to DEVICE memory of the caller (block
// i's at d_out + blocks[i].out_off) and stay there; what comes back is every block's status and the CRC-32 of its bytes, computed
// on the device, for the caller to compare with the checksum in the file.  max_out_bytes of pvc3_bgzf_inflater_create may be 1 for
// an inflater that is only used this way (no page-locked output is needed).
int pvc3_bgzf_inflater_submit_device(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t n_blocks, size_t in_bytes, void *d_out, size_t d_out_bytes)
{
	if (!h || slot >= h->n_slots || 0 == n_blocks || n_blocks > h->max_blocks || in_bytes > h->max_in || !d_out)
		return fail(E_INVALID, "pvc3_bgzf_inflater_submit_device: bad argument");
	auto &s = h->slots[slot];
	if (s.busy) return fail(E_STATE, "pvc3_bgzf_inflater_submit_device: the slot has not been waited for");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	if (!h->d_off)
	{
		HIP_TRY(hipMalloc((void **) &h->d_off, h->max_blocks * sizeof(uint64_t)), "hipMalloc");
		HIP_TRY(hipMalloc((void **) &h->d_lens, h->max_blocks * sizeof(uint32_t)), "hipMalloc");
		HIP_TRY(hipMalloc((void **) &h->d_crc, h->max_blocks * sizeof(uint32_t)), "hipMalloc");
		HIP_TRY(hipMalloc((void **) &h->d_crc_tables, sizeof(pvc3_crc::tables)), "hipMalloc");
		pvc3_crc::tables const t = pvc3_crc::make_tables();
		HIP_TRY(hipMemcpy(h->d_crc_tables, &t, sizeof(t), hipMemcpyHostToDevice), "hipMemcpy");
	}
	if (!s.h_off)
	{
		HIP_TRY(hipHostMalloc((void **) &s.h_off, h->max_blocks * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc");
		HIP_TRY(hipHostMalloc((void **) &s.h_lens, h->max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
		HIP_TRY(hipHostMalloc((void **) &s.h_crc, h->max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
	}
	for (uint32_t i = 0; i < n_blocks; ++i)
	{
		pvc3_bgzf_block const &b = s.h_blocks[i];
		if (b.in_off > in_bytes || b.in_len > in_bytes - b.in_off || b.out_off > d_out_bytes || b.out_len > d_out_bytes - b.out_off)
			return fail(E_INVALID, "pvc3_bgzf_inflater_submit_device: a block outside the slot's input or the output");
		s.h_off[i] = b.out_off; s.h_lens[i] = b.out_len;
	}
	HIP_TRY(hipMemcpyAsync(h->d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(h->d_blocks, s.h_blocks, n_blocks * sizeof(pvc3_bgzf_block), hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(h->d_off, s.h_off, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(h->d_lens, s.h_lens, n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	if (int const rc = launch_inflate(h, (uint8_t *) d_out, d_out_bytes, n_blocks, s.start, s.stop)) return rc;
	hipLaunchKernelGGL(pvc3_crc::k_crc32_ranges, dim3((n_blocks < 4096) ? n_blocks : 4096), dim3(64), 0, h->stream, (uint8_t const *) d_out, h->d_off, h->d_lens, 0u, 0u, n_blocks, h->d_crc_tables, h->d_crc);
	HIP_TRY(hipGetLastError(), "k_crc32_ranges");
	HIP_TRY(hipMemcpyAsync(s.h_status, h->d_status, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.h_crc, h->d_crc, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.done, h->stream), "hipEventRecord");
	s.n = n_blocks; s.busy = true;
	return 0;
}

int pvc3_bgzf_inflater_wait_device(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t const **status, uint32_t const **crc, float *kernel_ms)
{
	if (!h || slot >= h->n_slots) return fail(E_INVALID, "pvc3_bgzf_inflater_wait_device: bad argument");
	auto &s = h->slots[slot];
	if (!s.busy) return fail(E_STATE, "pvc3_bgzf_inflater_wait_device: nothing was submitted");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	s.busy = false;
	HIP_TRY(hipEventSynchronize(s.done), "hipEventSynchronize");
	if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s.start, s.stop), "hipEventElapsedTime");
	if (status) *status = s.h_status;
	if (crc) *crc = s.h_crc;
	return 0;
}

int pvc3_bgzf_inflater_wait(pvc3_bgzf_inflater *h, uint32_t slot, uint32_t const **status, float *kernel_ms)
{
	if (!h || slot >= h->n_slots) return fail(E_INVALID, "pvc3_bgzf_inflater_wait: bad argument");
	auto &s = h->slots[slot];
	if (!s.busy) return fail(E_STATE, "pvc3_bgzf_inflater_wait: nothing was submitted");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	s.busy = false;
	HIP_TRY(hipEventSynchronize(s.done), "hipEventSynchronize");
	if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s.start, s.stop), "hipEventElapsedTime");
	if (status) *status = s.h_status;
	return 0;
}

}
