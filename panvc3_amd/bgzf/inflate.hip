// inflate.hip — BGZF blocks inflated on the GPU, one wavefront per block (libpvc3_bgzf.so; C ABI in include/pvc3_bgzf.h).
//
// No reference code corresponds to this file: the reference reads BAM through seqan3 / zlib.  The BGZF reader of
// panvc3_project_alignments inflates every block of its input; with the output's blocks compressed on the GPU (deflate.hip)
// that was the largest share of the host's CPU time that was left.
//
// DEFLATE is decoded one symbol after the other, so a block is one wave's work.  What makes that slow is the chain of dependent
// steps from one symbol to the next; what made the first form of this file (one kernel; rounds 3 and 4, profiles/r04/
// inflate_phases.txt) expensive was the 32 KB of output every block in flight held in LDS for a match to copy from while it
// was being decoded — four blocks per CU, and nothing else of a caller's fitted next to them.  The two do not need each other:
// decoding needs the bit stream and the code tables, the window is only needed to COPY.  So a block is two kernels' work:
//   k_inflate_decode         decodes.  A literal goes straight to its place in the output in global memory, a match becomes a
//                            token (place, length, distance) in global memory.  10 KB of LDS, sixteen blocks per CU.
//     look-ups   lane k decodes the symbol that would begin at bit k of the window WHOLE (table entries hold base values and
//                numbers of extra bits, so there is no branch in it): a literal, or a length, the distance code behind it — a
//                second look-up — and the extra bits of both;
//     the walk   a scalar loop follows the symbols that do begin somewhere (o, o + bits of the symbol at o, ...: one v_readlane
//                and a dozen scalar instructions each) and hands each its place in the output; the lanes on that chain then
//                store, all at once — five or six symbols for two trips to LDS;
//     staging    the compressed input comes through a 2 KB ring in LDS, 64 dwords of it at a time through the lanes of a
//                register; the window itself (97 .. 128 bits) is the same in every lane and lives in scalar registers;
//     tables     canonical codes from the code lengths by all lanes: a 10-bit table for literals / lengths, 9 bits for
//                distances; the few longer codes are found by their length, first code and rank — no second-level tables;
//   k_inflate_resolve_lanes  takes a block's output — the literals are in place — into LDS whole (64 KB: two blocks per CU, for
//                            a fraction of the time), copies the matches, 64 at a time and each lane its own wherever the
//                            matches do not read each other's output, and stores the block back.
// 32.9 GB/s of output at 4 096 blocks a launch (decoding 5.5 ms, resolving 2.6 ms), against 12.4 of the first form.
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
enum : uint32_t
{
	ST_OK = 0, ST_BAD_BLOCK_TYPE = 1, ST_BAD_STORED = 2, ST_BAD_CODE_LENGTHS = 3, ST_BAD_SYMBOL = 4, ST_BAD_DISTANCE = 5, ST_OUTPUT_OVERFLOW = 6,
	ST_INPUT_OVERRUN = 7, ST_LENGTH_MISMATCH = 8, ST_TOO_LARGE = 9
};

struct code_info		// canonical decoding of the codes longer than the table's index
{
	uint32_t count[16], first[16], offset[16];
};

__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

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
	uint32_t staged_until{};	// dwords of the input in the ring so far
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

	// Lane j gets dword first + j; the ring is filled up to there first, half a ring at a time (it then holds the 512 dwords
	// before staged_until, `first` among them: staging stops less than 320 dwords past it, and a batch that begins a few dwords
	// before the last one — the fast path after the one-symbol path has read ahead — is still there).
	__device__ __forceinline__ void load_batch(uint32_t first)
	{
		while (staged_until < first + WAVE) { stage(staged_until); staged_until += RING_WORDS / 2; }
		batch_first = first;
		batch = ring[(first + threadIdx.x) & (RING_WORDS - 1)];
	}

	__device__ void start(uint8_t const *data, uint32_t n_bytes, uint32_t *ring_)
	{
		ring = ring_;
		uintptr_t const a = reinterpret_cast <uintptr_t>(data);
		src = reinterpret_cast <uint32_t const *>(a & ~uintptr_t(3));
		uint32_t const lead = (uint32_t) (a & 3);
		n_words = (lead + n_bytes + 3) / 4;
		next_word = 0; lo = 0; hi = 0; valid = 0; lead_bits = 8 * lead; staged_until = 0;
		__syncthreads();
		load_batch(0);
		refill();
		drop(lead_bits);
		refill();
	}

	// At least 97 bits in the window afterwards (zeros behind the end of the input: consumed_bits() tells).
	__device__ __forceinline__ void refill()
	{
		while (valid <= 96)
		{
			if (next_word - batch_first >= WAVE) load_batch(next_word);
			uint64_t const w = (uint32_t) __builtin_amdgcn_readlane((int) batch, (int) (next_word - batch_first));
			if (valid >= 64) hi |= w << (valid - 64);
			else { lo |= w << valid; if (valid > 32) hi |= w >> (64 - valid); }
			valid += 32;
			++next_word;
		}
	}

	// Where the window begins, in bits from the first staged dword; and the window moved there (forwards).
	__device__ __forceinline__ uint32_t position() const { return 32 * next_word - valid; }
	__device__ __forceinline__ void seek(uint32_t bit)
	{
		next_word = bit >> 5; lo = 0; hi = 0; valid = 0;
		refill();
		drop(bit & 31);
		refill();
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

// Canonical codes of lens[0 .. n) (RFC 1951 3.2.2) into a look-up table indexed by the next `bits` bits of the stream
// (entry = make_entry(symbol, length)) and, for longer codes, the symbols sorted by (length, symbol) with the first code and the
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

// ---- between the two kernels ------------------------------------------------------------------------------------------------------
typedef unsigned long long token_t;		// position | length << 17 | distance << 32
__device__ __forceinline__ token_t make_token(uint32_t pos, uint32_t length, uint32_t dist) { return (token_t) pos | ((token_t) length << 17) | ((token_t) dist << 32); }
// Tokens of the block whose bytes are out[off, off + len): a match makes at least three bytes, so floor(len / 3) tokens at most, and
// ranges of bytes that do not overlap get ranges of tokens that do not overlap.
__device__ __forceinline__ size_t token_base(uint64_t out_off) { return (size_t) ((out_off + 2) / 3); }

// -DPVC3_INF_CYCLES (tools/calib/inf_variants/cycles.py): the shader clock at the seams of a pass, summed per block and left in the
// block's first 72 bytes of output instead of its bytes.
#ifdef PVC3_INF_CYCLES
#define TICK(k) do { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); cyc[k] += now_ - last_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_) :: "memory"); } while (0)
#else
#define TICK(k) do {} while (0)
#endif

// ---- decoding ---------------------------------------------------------------------------------------------------------------------------
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

// Sum of v over the lanes up to and including this one: within rows of 16 lanes by DPP shifts, then lane 15 of a row to the next row,
// lane 31 to the upper two (the sequence LLVM's atomic optimizer emits for gfx9).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);		// row_shr:1
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);		// row_shr:2
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);		// row_shr:4
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);		// row_shr:8
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);		// row_bcast:15 to rows 1 and 3
	v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);		// row_bcast:31 to rows 2 and 3
	return v;
}

// (A symbol takes 10 + 5 + 9 + 13 = 37 bits at most through the tables, and every lane looks at 64 of its own: the walk goes on
// while a symbol begins in the lanes.)

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
#ifdef PVC3_INF_CYCLES
		uint64_t cyc[9] = {}, last_ = 0;
#endif

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
			// One pass: lane k decodes the symbol that would begin at bit k WHOLE — the literal, or the length with its extra bits,
			// the distance code behind them (a second look-up) and its extra bits: 37 bits at most of the 64 it looks at — and says
			// how many bits it takes and how many bytes it makes.  A scalar walk then only follows the chain (o, o + bits of the
			// symbol at o, ...: one v_readlane and seven scalar instructions per symbol); the lanes on the chain get the places of
			// their output from a prefix sum across the wave and store — a literal its byte, a match its token.
			// The bits come straight from the staged dwords here (five of them, out of the lanes, hold every lane's 64 bits); the
			// scalar window is set up again behind the block.  What the lanes do not decode — codes longer than the tables' index
			// (3.7 % of the symbols of zlib's level 6 on BAM records) and the end of the block — goes through the one-symbol path
			// behind the walk.
			uint32_t at_bit = br.position();
#ifdef PVC3_INF_CYCLES
			asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_) :: "memory");
#endif
			for (bool end_of_block = false; !end_of_block && ST_OK == st; )
			{
				TICK(7);
				uint32_t const w0 = at_bit >> 5;
				if (w0 < br.batch_first || w0 + 5 > br.batch_first + WAVE) br.load_batch(w0);		// (before: the one-symbol path has read ahead)
				uint32_t const i0 = w0 - br.batch_first;
				uint32_t const s0 = (uint32_t) __builtin_amdgcn_readlane((int) br.batch, (int) i0), s1 = (uint32_t) __builtin_amdgcn_readlane((int) br.batch, (int) (i0 + 1)),
					s2 = (uint32_t) __builtin_amdgcn_readlane((int) br.batch, (int) (i0 + 2)), s3 = (uint32_t) __builtin_amdgcn_readlane((int) br.batch, (int) (i0 + 3)),
					s4 = (uint32_t) __builtin_amdgcn_readlane((int) br.batch, (int) (i0 + 4));
				uint32_t const q = (at_bit & 31) + lane;		// 0 .. 94
				bool const z0 = q < 32, z1 = q < 64;
				uint32_t const d0 = z0 ? s0 : z1 ? s1 : s2, d1 = z0 ? s1 : z1 ? s2 : s3, d2 = z0 ? s2 : z1 ? s3 : s4;
				uint64_t const mine = (uint64_t) __builtin_amdgcn_alignbit(d1, d0, q & 31) | ((uint64_t) __builtin_amdgcn_alignbit(d2, d1, q & 31) << 32);
				asm volatile("" :: "v"(mine));
				TICK(0);
				uint32_t const e = L.t_ll[(uint32_t) mine & ((1u << LL_BITS) - 1)];
				uint32_t const l = (e >> 9) & 15, x = (e >> 13) & 7, kind = e >> 16;
				uint32_t const length = (e & 0x1ff) + __builtin_amdgcn_ubfe((uint32_t) (mine >> l), 0, x);		// (a literal's byte, for a literal)
				uint32_t const behind = (uint32_t) (mine >> (l + x));
				asm volatile("" :: "v"(behind), "v"(length));
				TICK(1);
				uint32_t const ed = L.t_d[behind & ((1u << D_BITS) - 1)];
				uint32_t const ld = (ed >> 15) & 15, xd = (ed >> 19) & 15;
				uint32_t const dist = (ed & 0x7fff) + __builtin_amdgcn_ubfe(behind >> ld, 0, xd);
				bool const is_match = K_MATCH == kind && 0 != (ed >> 23);
				uint32_t const bits = (K_LITERAL == kind) ? l : is_match ? (l + x + ld + xd) : 0u;		// none: the walk stops here
				asm volatile("" :: "v"(bits), "v"(dist));
				TICK(2);
				uint32_t o = 0;
				uint64_t chain = 0;
				uint32_t t = (uint32_t) __builtin_amdgcn_readlane((int) bits, 0);
				// while (t) { chain |= 1 << o; o += t; if (o > 63) break; t = readlane(bits, o); } — written out, because the compiler makes
				// fifteen instructions and two taken branches per symbol of it, and this loop is where a block's time goes: eight symbols in
				// a straight line (a pass has six on average), so that the only branch taken is the one that leaves.  (s_nop 1: four wait
				// states between the scalar write of o and its use as v_readlane's lane select.)
				if (t)
					asm volatile("1:\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_bitset1_b64 %[chain], %[o]\n\t"
						"s_add_u32 %[o], %[o], %[t]\n\t"
						"s_cmp_gt_u32 %[o], 63\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_nop 1\n\t"
						"v_readlane_b32 %[t], %[bits], %[o]\n\t"
						"s_cmp_eq_u32 %[t], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_branch 1b\n\t"
						"2:"
						: [chain] "+s"(chain), [o] "+s"(o), [t] "+s"(t) : [bits] "v"(bits) : "scc");
				TICK(3);
				bool const stopped = 0 == t;
				bool const on_chain = 0 != ((chain >> lane) & 1);
				uint32_t const makes = !on_chain ? 0u : (K_LITERAL == kind) ? 1u : length;
				uint32_t const upto = wave_inclusive_sum(makes);
				uint32_t const made = (uint32_t) __builtin_amdgcn_readlane((int) upto, 63);
				TICK(4);
				if (pos + made > want) { st = ST_OUTPUT_OVERFLOW; break; }
				uint32_t const at = pos + upto - makes;
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
				at_bit += o;
				TICK(5);
#ifdef PVC3_INF_CYCLES
				++cyc[8];
#endif
				if (!stopped) continue;

				// one symbol the long way — a code longer than the table's index, the end of the block, or a code that does not exist — out
				// of the 64 bits the lane it begins in has looked at (48 at most: 15 + 5 + 15 + 13)
				uint64_t w = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) mine, (int) o) | ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (mine >> 32), (int) o) << 32);
				uint32_t e1 = uni(L.t_ll[(uint32_t) w & ((1u << LL_BITS) - 1)]);
				if (0 == e1)
				{
					uint32_t sym = 0, l1 = 0;
					if (!long_code((uint32_t) w, LL_BITS + 1, L.sorted_ll, L.ll, sym, l1)) { st = ST_BAD_SYMBOL; break; }
					e1 = litlen_entry32(sym, l1);
					if (0 == e1) { st = ST_BAD_SYMBOL; break; }
				}
				uint32_t used = (e1 >> 9) & 15;
				w >>= used;
				uint32_t const kind1 = e1 >> 16;
				if (K_LITERAL == kind1)
				{
					if (pos >= want) { st = ST_OUTPUT_OVERFLOW; break; }
					if (0 == lane) dst[pos] = (uint8_t) e1;
					++pos;
					at_bit += used;
					continue;
				}
				if (K_END == kind1) { at_bit += used; end_of_block = true; break; }
				uint32_t const x1 = (e1 >> 13) & 7;
				uint32_t const length1 = (e1 & 0x1ff) + ((uint32_t) w & ((1u << x1) - 1));
				w >>= x1; used += x1;
				uint32_t ed1 = uni(L.t_d[(uint32_t) w & ((1u << D_BITS) - 1)]);
				if (0 == ed1)
				{
					uint32_t ds = 0, l1 = 0;
					if (!long_code((uint32_t) w, D_BITS + 1, L.sorted_d, L.d, ds, l1)) { st = ST_BAD_SYMBOL; break; }
					ed1 = dist_entry32(ds, l1);
					if (0 == ed1) { st = ST_BAD_SYMBOL; break; }
				}
				uint32_t const ld1 = (ed1 >> 15) & 15, xd1 = (ed1 >> 19) & 15;
				w >>= ld1;
				uint32_t const dist1 = (ed1 & 0x7fff) + ((uint32_t) w & ((1u << xd1) - 1));
				used += ld1 + xd1;
				if (dist1 > pos) { st = ST_BAD_DISTANCE; break; }
				if (pos + length1 > want) { st = ST_OUTPUT_OVERFLOW; break; }
				if (0 == lane) tok[n_tok] = make_token(pos, length1, dist1);
				++n_tok;
				pos += length1;
				at_bit += used;
				TICK(6);
			}
			br.seek(at_bit);		// the scalar window again, for what follows the block
		}
		if (ST_OK == st && br.consumed_bits() > uint64_t(8) * blk.in_len) st = ST_INPUT_OVERRUN;
		if (ST_OK == st && pos != want) st = ST_LENGTH_MISMATCH;
#ifdef PVC3_INF_CYCLES
		__threadfence();
		if (0 == lane && want >= 72) { for (int k = 0; k < 72; ++k) dst[k] = (uint8_t) (cyc[k / 8] >> (8 * (k & 7))); n_tok = 0; }
#endif
		if (0 == lane) { status[b] = st; n_tokens[b] = (ST_OK == st) ? n_tok : 0; }
		__syncthreads();		// the tables are the next block's from here
	}
}

// ---- resolving ------------------------------------------------------------------------------------------------------------------------
// A match at a time across the lanes costs the latency of LDS — read, wait, write — for a dozen bytes (the first form of this
// kernel: 0.70 ms a block).  Most matches of a stretch do not depend on each other (they copy from a record or two back), so: the
// matches come 64 at a time, one per lane; a lane finds the earlier matches of the 64 whose output it reads (their places ascend:
// two binary searches through ds_bpermute, and none at all when every source lies before the first place of the 64); then, in
// rounds, every match whose sources are all written copies at once — each lane its own, up to 32 bytes in four
// unaligned 8-byte accesses each way — and the few that are longer, or repeat a pattern of their own output, go one at a time across
// the lanes.  (zlib level 6 on BAM records, tools/calib/token_stats.py: 5 900 matches a block, 99 % of them at most 32 bytes, 18 % read
// another of their 64, 3.8 rounds per 64.)
constexpr unsigned RESOLVE_BYTES = 65536;		// a block's output: ISIZE at most
constexpr unsigned RESOLVE_SLACK = 48;		// the block may begin 15 bytes in, and a lane reads 24 bytes from its match's first whatever its length
constexpr unsigned LANE_COPY_MAX = 32;
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
						// eight bytes at a time at any address (LDS takes unaligned accesses on this target), the last eight — or four — of a
						// match once more where they end; every read before the first write
						uint8_t const *src = B + from;
						uint8_t *to = B + pos;
						bool const small = length < 8;
						uint64_t r0, r1, r2, rt;
						__builtin_memcpy(&r0, src, 8);
						__builtin_memcpy(&r1, src + 8, 8);
						__builtin_memcpy(&r2, src + 16, 8);
						__builtin_memcpy(&rt, src + (small ? 0 : length - 8), 8);
						if (1 == dist) r0 = r1 = r2 = rt = (r0 & 0xff) * UINT64_C(0x0101010101010101);
						if (small)
						{
							uint32_t const lo = (uint32_t) r0;
							if (3 == length) { uint16_t const h = (uint16_t) lo; __builtin_memcpy(to, &h, 2); to[2] = (uint8_t) (lo >> 16); }
							else
							{
								uint32_t const hi = (uint32_t) (r0 >> (8 * (length - 4)));
								__builtin_memcpy(to, &lo, 4);
								__builtin_memcpy(to + length - 4, &hi, 4);
							}
						}
						else
						{
							__builtin_memcpy(to, &r0, 8);
							if (length >= 16) __builtin_memcpy(to + 8, &r1, 8);
							if (length >= 24) __builtin_memcpy(to + 16, &r2, 8);
							__builtin_memcpy(to + length - 8, &rt, 8);
						}
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
	uint32_t max_blocks{}, n_slots{}, grid_max{}, grid_full{}, grid_resolve{};
	pvc3_crc::tables *d_crc_tables{};
	// Everything else is a slot's own, the stream included: the groups of two slots are on the GPU at the same time.  A launch of a
	// thousand blocks takes as long as a launch of one (a block is one wave's work from its first bit to its last), so two launches
	// one after the other on one stream, as it was, left most of the chip idle twice.
	struct slot
	{
		hipStream_t stream{};
		uint8_t *d_in{}, *d_out{};
		pvc3_bgzf_block *d_blocks{};
		uint32_t *d_status{};
		// the matches of every block between the two kernels
		token_t *d_tokens{};
		size_t tokens_cap{};				// in tokens
		uint32_t *d_ntok{};
		// the device-resident form (pvc3_bgzf_inflater_submit_device): where the blocks' bytes went and their checksums
		uint64_t *d_off{};
		uint32_t *d_lens{}, *d_crc{};
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

// The two kernels of one group of blocks on the slot's stream (with room for the tokens of out_bytes of output).
int launch_inflate(pvc3_bgzf_inflater *h, pvc3_bgzf_inflater::slot &s, uint8_t *d_out, size_t out_bytes, uint32_t n_blocks)
{
	size_t const need = out_bytes / 3 + n_blocks + 64;
	if (need > s.tokens_cap)
	{
		HIP_TRY(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
		if (s.d_tokens) { (void) hipFree(s.d_tokens); s.d_tokens = nullptr; s.tokens_cap = 0; }
		HIP_TRY(hipMalloc((void **) &s.d_tokens, (need + need / 8) * sizeof(token_t)), "hipMalloc");
		s.tokens_cap = need + need / 8;
	}
	uint32_t const grid = (n_blocks < h->grid_max) ? n_blocks : h->grid_max, grid_resolve = (n_blocks < h->grid_resolve) ? n_blocks : h->grid_resolve;
	HIP_TRY(hipEventRecord(s.start, s.stream), "hipEventRecord");
	hipLaunchKernelGGL(k_inflate_decode, dim3(grid), dim3(WAVE), 0, s.stream, s.d_in, s.d_blocks, d_out, s.d_status, n_blocks, s.d_tokens, s.d_ntok);
	hipLaunchKernelGGL(k_inflate_resolve_lanes, dim3(grid_resolve), dim3(WAVE), 0, s.stream, s.d_blocks, d_out, s.d_status, n_blocks, s.d_tokens, s.d_ntok);
	HIP_TRY(hipGetLastError(), "k_inflate_decode");
	HIP_TRY(hipEventRecord(s.stop, s.stream), "hipEventRecord");
	return 0;
}

}

extern "C" {

void pvc3_bgzf_inflater_destroy(pvc3_bgzf_inflater *h)
{
	if (!h) return;
	(void) hipSetDevice(h->device);
	for (auto &s : h->slots)
	{
		if (s.stream) (void) hipStreamSynchronize(s.stream);
		for (void *p : {(void *) s.h_in, (void *) s.h_out, (void *) s.h_blocks, (void *) s.h_status, (void *) s.h_off, (void *) s.h_lens, (void *) s.h_crc}) if (p) (void) hipHostFree(p);
		for (void *p : {(void *) s.d_in, (void *) s.d_out, (void *) s.d_blocks, (void *) s.d_status, (void *) s.d_tokens, (void *) s.d_ntok, (void *) s.d_off, (void *) s.d_lens, (void *) s.d_crc}) if (p) (void) hipFree(p);
		if (s.start) (void) hipEventDestroy(s.start);
		if (s.stop) (void) hipEventDestroy(s.stop);
		if (s.done) (void) hipEventDestroy(s.done);
		if (s.stream) (void) hipStreamDestroy(s.stream);
	}
	if (h->d_crc_tables) (void) hipFree(h->d_crc_tables);
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
		check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_decode, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		if (per_cu < 1) per_cu = 1;
		h->grid_max = (uint32_t) prop.multiProcessorCount * (uint32_t) per_cu;
		h->grid_full = h->grid_max;
		int per_cu_b = 0;
		check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_inflate_resolve_lanes, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		h->grid_resolve = (uint32_t) prop.multiProcessorCount * (uint32_t) ((per_cu_b < 1) ? 1 : per_cu_b);
	}
	for (uint32_t i = 0; i < n_slots && 0 == rc; ++i)
	{
		auto &s = h->slots[i];
		check(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate");
		if (0 == rc) check(hipMalloc((void **) &s.d_in, max_in_bytes + IN_SLACK), "hipMalloc");
		if (0 == rc) check(hipMalloc((void **) &s.d_out, max_out_bytes), "hipMalloc");
		if (0 == rc) check(hipMalloc((void **) &s.d_blocks, max_blocks * sizeof(pvc3_bgzf_block)), "hipMalloc");
		if (0 == rc) check(hipMalloc((void **) &s.d_status, max_blocks * sizeof(uint32_t)), "hipMalloc");
		if (0 == rc) check(hipMalloc((void **) &s.d_ntok, max_blocks * sizeof(uint32_t)), "hipMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_in, max_in_bytes + IN_SLACK, hipHostMallocDefault), "hipHostMalloc");
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

// Fewer blocks of a launch at a time than the chip's LDS holds (0: as many as it holds again), for a caller whose other kernels run
// next to the inflater's.
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
	HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.d_blocks, s.h_blocks, n_blocks * sizeof(pvc3_bgzf_block), hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	if (int const rc = launch_inflate(h, s, s.d_out, h->max_out, n_blocks)) return rc;
	HIP_TRY(hipMemcpyAsync(s.h_status, s.d_status, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
	if (out_bytes) HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, out_bytes, hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.done, s.stream), "hipEventRecord");
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
	if (!h->d_crc_tables)
	{
		HIP_TRY(hipMalloc((void **) &h->d_crc_tables, sizeof(pvc3_crc::tables)), "hipMalloc");
		pvc3_crc::tables const t = pvc3_crc::make_tables();
		HIP_TRY(hipMemcpy(h->d_crc_tables, &t, sizeof(t), hipMemcpyHostToDevice), "hipMemcpy");
	}
	if (!s.d_off)
	{
		HIP_TRY(hipMalloc((void **) &s.d_off, h->max_blocks * sizeof(uint64_t)), "hipMalloc");
		HIP_TRY(hipMalloc((void **) &s.d_lens, h->max_blocks * sizeof(uint32_t)), "hipMalloc");
		HIP_TRY(hipMalloc((void **) &s.d_crc, h->max_blocks * sizeof(uint32_t)), "hipMalloc");
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
	HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.d_blocks, s.h_blocks, n_blocks * sizeof(pvc3_bgzf_block), hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.d_off, s.h_off, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.d_lens, s.h_lens, n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
	if (int const rc = launch_inflate(h, s, (uint8_t *) d_out, d_out_bytes, n_blocks)) return rc;
	hipLaunchKernelGGL(pvc3_crc::k_crc32_ranges, dim3((n_blocks < 4096) ? n_blocks : 4096), dim3(64), 0, s.stream, (uint8_t const *) d_out, s.d_off, s.d_lens, 0u, 0u, n_blocks, h->d_crc_tables, s.d_crc);
	HIP_TRY(hipGetLastError(), "k_crc32_ranges");
	HIP_TRY(hipMemcpyAsync(s.h_status, s.d_status, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.h_crc, s.d_crc, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.done, s.stream), "hipEventRecord");
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
