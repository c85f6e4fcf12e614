// deflate.hip — BGZF blocks compressed on the GPU, one wavefront per block (libpvc3_bgzf.so; C ABI in include/pvc3_bgzf.h).
//
// No reference code corresponds to this file: the reference writes BAM through seqan3 / zlib.  It exists because DEFLATE was
// 7 of the 15 core-seconds panvc3_project_alignments spent per four million records on the 16 host cores of a GPU box
// (DESIGN.md §8), while the GPU next to them was idle nine tenths of the time.
//
// A block (<= 65 280 bytes) is one wave's work from the first byte to the last bit, so that nothing is exchanged between
// waves and a launch of a few thousand blocks fills the chip with independent streams of dependent loads:
//
//   parse     64 positions at a time, lane = position.  The candidate of a position is the latest earlier position with the
//             same hash of its four bytes, from a 2048-entry table in LDS that holds positions of *earlier windows* only
//             (read by all lanes, then updated with ds_max: the largest position wins, whatever the order of the lanes), or
//             the position one byte back if that matches (runs of one quality value or base: the matches the table cannot
//             see because they begin inside the window).  Every lane extends its candidate eight bytes per step.  The greedy
//             choice (take the match at `next`, else a literal) is a scalar loop over the 64 lengths with v_readlane; the
//             chosen lanes append their tokens to the block's token list in global memory and count their symbols in LDS.
//   codes     Huffman's construction for the three alphabets: rank sort by all lanes, the two-queue merge by one lane, depths
//             by pointer chasing from every leaf; lengths over the limit: halve the frequencies and repeat (as the host's
//             encoder, fast_deflate.hpp); canonical codes, bit-reversed.  The code length sequence is sent without the
//             repeat symbols 16-18 (a tenth of a per cent of the output, and every symbol's position in the bit stream is a
//             prefix sum).
//   emit      64 tokens at a time: bit lengths, a wave scan, ds_or of every token's bits into a staging area in LDS, whole
//             dwords from there to the output.  The size is known before the first bit: a block that would not shrink is stored.
//
// Measured on MI355X (tools/bgzf_bench.py): see DESIGN.md §8.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/pvc3_bgzf.h"
#include "crc32_device.hpp"

thread_local char g_bgzf_error[256];		// the message of the calling thread's last failure (both files of the library)

namespace {

// 2 048 entries: on BAM records the output is as small as with 4 096 (0.3329 against 0.3334 of the input) and as fast, and a block
// being compressed holds 11.4 KB of its CU's LDS instead of 19.6 (tools/calib/deflate_hash_bits.py; the code construction's scratch
// needs 2 048 dwords of it).
#ifndef PVC3_DEFLATE_HASH_BITS
#define PVC3_DEFLATE_HASH_BITS 11
#endif
constexpr unsigned HASH_BITS = PVC3_DEFLATE_HASH_BITS;
constexpr unsigned WAVE = 64;
constexpr unsigned N_LL = 286, N_D = 30, N_CL = 19;
constexpr unsigned MAX_MATCH = 258, MAX_DIST = 32768;
constexpr unsigned TOKENS_PER_WAVE = PVC3_BGZF_BLOCK;
constexpr uint32_t SIZE_MISMATCH = 0x80000000u;		// out_len flag: the stream is not as long as computed (a bug)

struct lds_t
{
	uint32_t head[1u << HASH_BITS];		// parse: hash -> position + 1 (0: none); afterwards scratch of the code construction
	uint32_t f_ll[288], f_d[32], f_cl[32];
	uint32_t t_ll[288], t_d[32], t_cl[32];		// bit-reversed code | length << 16
	uint32_t stage[112];
};

// Scratch of build_code inside lds_t::head (in dwords).
constexpr unsigned S_FW = 0, S_ORD = 288, S_WT = 576, S_PAR = 1152, S_LEN = 1728, S_BL = 2016, S_NC = 2032;
static_assert(S_NC + 16 <= (1u << HASH_BITS), "scratch fits the hash table");

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
	unsigned const lane = threadIdx.x;
#pragma unroll
	for (unsigned d = 1; d < WAVE; d <<= 1)
	{
		uint32_t const y = __shfl_up(x, d);
		if (lane >= d) x += y;
	}
	return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
	for (unsigned d = WAVE / 2; d; d >>= 1) x += __shfl_xor(x, d);
	return x;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t x)
{
#pragma unroll
	for (unsigned d = WAVE / 2; d; d >>= 1) { uint32_t const y = __shfl_xor(x, d); x = (y > x) ? y : x; }
	return x;
}

// Eight bytes from any address: three dwords from the dword below and two byte alignments (a 16-byte load from an address
// that is not a multiple of four costs the texture addresser three to four times as much, profiles/r03_final/ta_cost.txt).
__device__ __forceinline__ uint64_t load8(uint8_t const *p)
{
	uintptr_t const a = reinterpret_cast <uintptr_t>(p);
	uint32_t const *q = reinterpret_cast <uint32_t const *>(a & ~uintptr_t(3));
	unsigned const s = (unsigned) (a & 3);
	uint32_t const w0 = q[0], w1 = q[1], w2 = q[2];
	uint32_t const lo = __builtin_amdgcn_alignbyte(w1, w0, s), hi = __builtin_amdgcn_alignbyte(w2, w1, s);
	return (uint64_t(hi) << 32) | lo;
}

__device__ __forceinline__ uint32_t load4(uint8_t const *p)
{
	uintptr_t const a = reinterpret_cast <uintptr_t>(p);
	uint32_t const *q = reinterpret_cast <uint32_t const *>(a & ~uintptr_t(3));
	return __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned) (a & 3));
}

// RFC 1951 3.2.5.  Length symbol - 257 of a match length - 3, number and value of its extra bits.
__device__ __forceinline__ void length_symbol(uint32_t l3, uint32_t &sym, uint32_t &n_extra, uint32_t &extra)
{
	if (l3 < 8) { sym = l3; n_extra = 0; extra = 0; return; }
	if (255 == l3) { sym = 28; n_extra = 0; extra = 0; return; }
	uint32_t const e = 29 - (uint32_t) __builtin_clz(l3);		// 8..15 -> 1, ..., 128..254 -> 5
	sym = 4 + 4 * e + ((l3 >> e) & 3);
	n_extra = e;
	extra = l3 & ((1u << e) - 1);
}

// Distance symbol of a distance - 1, number and value of its extra bits.
__device__ __forceinline__ void distance_symbol(uint32_t dd, uint32_t &sym, uint32_t &n_extra, uint32_t &extra)
{
	if (dd < 4) { sym = dd; n_extra = 0; extra = 0; return; }
	uint32_t const e = 31 - (uint32_t) __builtin_clz(dd);		// >= 2
	sym = 2 * e + ((dd >> (e - 1)) & 1);
	n_extra = e - 1;
	extra = dd & ((1u << (e - 1)) - 1);
}

__device__ __forceinline__ uint32_t length_extra_bits(uint32_t sym) { return (sym < 8 || 28 == sym) ? 0 : ((sym - 4) >> 2); }
__device__ __forceinline__ uint32_t distance_extra_bits(uint32_t sym) { return (sym < 4) ? 0 : ((sym >> 1) - 1); }

// Code lengths (<= limit) and canonical codes of freq[0 .. n) into table[s] = bit-reversed code | length << 16 (0 for an unused
// symbol).  At least min_used symbols get a code (inflate wants a complete code length code; RFC 1951 3.2.7 one distance code
// at least).  scr: lds_t::head.  Called by the whole wave.
__device__ void build_code(uint32_t const *freq, unsigned n, unsigned limit, unsigned min_used, uint32_t *table, uint32_t *scr)
{
	unsigned const lane = threadIdx.x;
	uint32_t *fw = scr + S_FW, *ord = scr + S_ORD, *wt = scr + S_WT, *par = scr + S_PAR, *len = scr + S_LEN, *bl = scr + S_BL, *nc = scr + S_NC;
	uint32_t mine = 0;
	for (unsigned s = lane; s < n; s += WAVE) { uint32_t const f = freq[s]; fw[s] = f; mine += (0 != f); }
	uint32_t used = wave_sum(mine);
	__syncthreads();
	if (used < min_used)
	{
		if (0 == lane)
			for (unsigned s = 0; s < n && used < min_used; ++s)
				if (0 == fw[s]) { fw[s] = 1; ++used; }
		__syncthreads();
		used = min_used;
	}
	for (;;)
	{
		// ascending by (weight, symbol): a symbol's rank is the number of used symbols before it
		for (unsigned s = lane; s < n; s += WAVE)
		{
			len[s] = 0;
			uint32_t const f = fw[s];
			if (0 == f) continue;
			uint32_t rank = 0;
			for (unsigned t = 0; t < n; ++t)
			{
				uint32_t const ft = fw[t];
				rank += (0 != ft) & ((ft < f) | ((ft == f) & (t < s)));
			}
			ord[rank] = s;
			wt[rank] = f;
		}
		__syncthreads();
		if (1 == used)
		{
			if (0 == lane) len[ord[0]] = 1;
			__syncthreads();
			break;
		}
		// Two queues: the leaves in ord / wt[0 .. used), the inner nodes come into being in non-decreasing weight behind them.
		if (0 == lane)
		{
			unsigned leaf = 0, inner = used, next = used;
			uint32_t w_leaf = wt[0], w_inner = 0;
			while ((used - leaf) + (next - inner) > 1)
			{
				unsigned pick[2];
				uint32_t w[2];
#pragma unroll
				for (unsigned k = 0; k < 2; ++k)
				{
					bool const take_leaf = leaf < used && (inner >= next || w_leaf <= w_inner);
					if (take_leaf) { pick[k] = leaf; w[k] = w_leaf; ++leaf; if (leaf < used) w_leaf = wt[leaf]; }
					else { pick[k] = inner; w[k] = w_inner; ++inner; if (inner < next) w_inner = wt[inner]; }
				}
				uint32_t const sum = w[0] + w[1];
				wt[next] = sum;
				par[pick[0]] = next; par[pick[1]] = next;
				if (inner == next) w_inner = sum;		// the queue of inner nodes was empty: this one is its head
				++next;
			}
		}
		__syncthreads();
		unsigned const root = 2 * used - 2;
		uint32_t deepest = 0;
		for (unsigned i = lane; i < used; i += WAVE)
		{
			uint32_t d = 0;
			for (unsigned j = i; j != root && d < 2 * N_LL; j = par[j]) ++d;		// (the bound: no endless loop whatever par holds)
			len[ord[i]] = d;
			deepest = (d > deepest) ? d : deepest;
		}
		deepest = wave_max(deepest);
		__syncthreads();
		if (deepest <= limit) break;
		for (unsigned s = lane; s < n; s += WAVE) fw[s] = (fw[s] + 1) >> 1;
		__syncthreads();
	}
	// canonical codes (RFC 1951 3.2.2)
	if (lane < 16) bl[lane] = 0;
	__syncthreads();
	for (unsigned s = lane; s < n; s += WAVE) if (len[s]) atomicAdd(&bl[len[s]], 1u);
	__syncthreads();
	if (0 == lane)
	{
		uint32_t c = 0, before = 0;
		for (unsigned l = 1; l < 16; ++l) { c = (c + before) << 1; nc[l] = c; before = bl[l]; }
	}
	__syncthreads();
	for (unsigned s = lane; s < n; s += WAVE)
	{
		uint32_t const l = len[s];
		if (0 == l) { table[s] = 0; continue; }
		uint32_t k = 0;
		for (unsigned t = 0; t < s; ++t) k += (len[t] == l);
		uint32_t const code = nc[l] + k;
		table[s] = (__brev(code) >> (32 - l)) | (l << 16);
	}
	__syncthreads();
}

__constant__ uint8_t c_cl_order[N_CL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The bit stream of one block: bits are or-ed into lds_t::stage, whole dwords go to the output.
struct bit_sink
{
	uint32_t *stage;
	uint32_t *out32;
	uint32_t bitpos{};			// bits in the stage, < 32 between calls
	uint32_t words{};			// dwords written

	// Every lane adds nb (<= 48) bits, in lane order.
	__device__ __forceinline__ void put(uint64_t bits, uint32_t nb)
	{
		unsigned const lane = threadIdx.x;
		uint32_t const incl = wave_scan_incl(nb);
		uint32_t const total = __shfl(incl, WAVE - 1);
		if (nb)
		{
			uint32_t const o = bitpos + incl - nb;
			uint32_t const word = o >> 5, sh = o & 31;
			uint64_t const lo = bits << sh;
			uint32_t const hi = sh ? (uint32_t) (bits >> (64 - sh)) : 0;
			atomicOr(&stage[word], (uint32_t) lo);
			if (lo >> 32) atomicOr(&stage[word + 1], (uint32_t) (lo >> 32));
			if (hi) atomicOr(&stage[word + 2], hi);
		}
		__syncthreads();
		bitpos += total;
		uint32_t const full = bitpos >> 5;
		uint32_t const carry = stage[full];
		for (unsigned w = lane; w < full; w += WAVE) out32[words + w] = stage[w];
		__syncthreads();
		for (unsigned w = lane; w <= full; w += WAVE) stage[w] = (0 == w) ? carry : 0;
		__syncthreads();
		words += full;
		bitpos &= 31;
	}
};

__global__ __launch_bounds__(WAVE) void k_deflate_blocks(
	uint8_t const *__restrict__ in, uint32_t const *__restrict__ in_len, uint8_t *__restrict__ out, uint32_t *__restrict__ out_len,
	uint32_t *__restrict__ tokens, uint32_t n_blocks)
{
	__shared__ lds_t L;
	unsigned const lane = threadIdx.x;
	uint32_t *tok = tokens + (size_t) blockIdx.x * TOKENS_PER_WAVE;

	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		uint8_t const *src = in + (size_t) b * PVC3_BGZF_IN_STRIDE + PVC3_BGZF_IN_PAD;
		uint8_t *dst = out + (size_t) b * PVC3_BGZF_OUT_STRIDE;
		uint32_t n = in_len[b];
		if (n > PVC3_BGZF_BLOCK) n = PVC3_BGZF_BLOCK;		// the host has refused such a length already
		if (0 == n)
		{
			if (0 == lane) { dst[0] = 3; dst[1] = 0; out_len[b] = 2; }		// fixed codes, end of block
			continue;
		}

		for (unsigned i = lane; i < (1u << HASH_BITS); i += WAVE) L.head[i] = 0;
		for (unsigned i = lane; i < 288; i += WAVE) L.f_ll[i] = 0;
		if (lane < 32) { L.f_d[lane] = 0; L.f_cl[lane] = 0; }
		for (unsigned i = lane; i < 112; i += WAVE) L.stage[i] = 0;
		__syncthreads();

		// ---- parse ------------------------------------------------------------------------------------------------------------
		uint32_t next = 0, n_tok = 0;
		for (uint32_t base = 0; base < n; base += WAVE)
		{
			uint32_t const p = base + lane;
			uint64_t const w = load8(src + p - 1);		// bytes p - 1 .. p + 6 (the pad in front of the block makes p = 0 safe)
			uint32_t const v = (uint32_t) (w >> 8);
			bool const hashable = p + 4 <= n;
			uint32_t const h = (v * 2654435761u) >> (32 - HASH_BITS);
			uint32_t const cand = hashable ? L.head[h] : 0;
			__syncthreads();
			if (hashable) atomicMax(&L.head[h], p + 1);

			bool const one_back = hashable && p >= 1 && (uint32_t) w == v;
			bool has = one_back;
			uint32_t c = p - 1;
			if (!one_back && cand && p + 1 - cand <= MAX_DIST)
			{
				c = cand - 1;
				has = load4(src + c) == v;
			}
			uint32_t m = 0;
			if (has)
			{
				uint32_t const max_len = (n - p < MAX_MATCH) ? (n - p) : MAX_MATCH;
				m = 4;
				while (m < max_len)
				{
					uint64_t const x = load8(src + p + m) ^ load8(src + c + m);
					if (x) { m += (uint32_t) (__builtin_ctzll(x) >> 3); break; }
					m += 8;
				}
				m = (m < max_len) ? m : max_len;
			}

			// the greedy choice: `next` is the first position no token covers yet
			uint32_t const lim = (base + WAVE < n) ? (base + WAVE) : n;
			uint64_t sel = 0;
			while (next < lim)
			{
				uint32_t const l = next - base;
				uint32_t const ml = __builtin_amdgcn_readlane(m, l);
				sel |= uint64_t(1) << l;
				next += ml ? ml : 1;
			}
			if ((sel >> lane) & 1)
			{
				uint32_t const idx = n_tok + (uint32_t) __popcll(sel & ((uint64_t(1) << lane) - 1));
				if (m)
				{
					uint32_t const dd = p - c - 1;
					uint32_t ls, le, lx, ds, de, dx;
					length_symbol(m - 3, ls, le, lx);
					distance_symbol(dd, ds, de, dx);
					tok[idx] = 0x80000000u | ((m - 3) << 16) | dd;
					atomicAdd(&L.f_ll[257 + ls], 1u);
					atomicAdd(&L.f_d[ds], 1u);
				}
				else
				{
					uint32_t const byte = v & 0xff;
					tok[idx] = byte;
					atomicAdd(&L.f_ll[byte], 1u);
				}
			}
			n_tok += (uint32_t) __popcll(sel);
		}
		__syncthreads();
		if (0 == lane) L.f_ll[256] = 1;
		__threadfence();		// the tokens are read back below
		__syncthreads();

		// ---- codes ------------------------------------------------------------------------------------------------------------
		build_code(L.f_ll, N_LL, 15, 2, L.t_ll, L.head);
		build_code(L.f_d, N_D, 15, 1, L.t_d, L.head);
		uint32_t n_ll = 0, n_d = 0;
		for (unsigned s = lane; s < N_LL; s += WAVE) if (L.t_ll[s] >> 16) n_ll = s + 1;
		if (lane < N_D && (L.t_d[lane] >> 16)) n_d = lane + 1;
		n_ll = wave_max(n_ll); n_d = wave_max(n_d);
		if (n_ll < 257) n_ll = 257;
		if (n_d < 1) n_d = 1;
		uint32_t const n_seq = n_ll + n_d;
		for (unsigned k = lane; k < n_seq; k += WAVE)
		{
			uint32_t const t = (k < n_ll) ? L.t_ll[k] : L.t_d[k - n_ll];
			atomicAdd(&L.f_cl[t >> 16], 1u);
		}
		__syncthreads();
		build_code(L.f_cl, N_CL, 7, 2, L.t_cl, L.head);
		uint32_t n_cl = 0;
		if (lane < N_CL && (L.t_cl[c_cl_order[lane]] >> 16)) n_cl = lane + 1;
		n_cl = wave_max(n_cl);
		if (n_cl < 4) n_cl = 4;

		// ---- the size, before a bit is written ------------------------------------------------------------------------------------
		uint32_t bits = 0;
		for (unsigned s = lane; s < N_LL; s += WAVE)
			bits += L.f_ll[s] * ((L.t_ll[s] >> 16) + ((s > 256) ? length_extra_bits(s - 257) : 0));
		if (lane < N_D) bits += L.f_d[lane] * ((L.t_d[lane] >> 16) + distance_extra_bits(lane));
		for (unsigned k = lane; k < n_seq; k += WAVE)
		{
			uint32_t const t = (k < n_ll) ? L.t_ll[k] : L.t_d[k - n_ll];
			bits += L.t_cl[t >> 16] >> 16;
		}
		bits = wave_sum(bits) + 17 + 3 * n_cl;
		uint32_t const size = (bits + 7) >> 3;
		if (size >= n + 5)
		{
			// stored (RFC 1951 3.2.4)
			if (0 == lane)
			{
				dst[0] = 1;
				dst[1] = (uint8_t) n; dst[2] = (uint8_t) (n >> 8); dst[3] = (uint8_t) ~n; dst[4] = (uint8_t) (~n >> 8);
				out_len[b] = n + 5;
			}
			for (uint32_t i = lane; i < n; i += WAVE) dst[5 + i] = src[i];
			__syncthreads();
			continue;
		}

		// ---- the bit stream --------------------------------------------------------------------------------------------------
		bit_sink sink{L.stage, reinterpret_cast <uint32_t *>(dst)};
		{
			uint64_t hb = 0; uint32_t hn = 0;
			if (0 == lane) { hb = 1u | (2u << 1) | ((n_ll - 257) << 3) | ((n_d - 1) << 8) | ((n_cl - 4) << 13); hn = 17; }
			else if (lane - 1 < n_cl) { hb = L.t_cl[c_cl_order[lane - 1]] >> 16; hn = 3; }
			sink.put(hb, hn);
		}
		for (uint32_t k0 = 0; k0 < n_seq; k0 += WAVE)
		{
			uint32_t const k = k0 + lane;
			uint64_t cb = 0; uint32_t cn = 0;
			if (k < n_seq)
			{
				uint32_t const t = (k < n_ll) ? L.t_ll[k] : L.t_d[k - n_ll];
				uint32_t const e = L.t_cl[t >> 16];
				cb = e & 0xffff; cn = e >> 16;
			}
			sink.put(cb, cn);
		}
		for (uint32_t t0 = 0; t0 <= n_tok; t0 += WAVE)
		{
			uint32_t const t = t0 + lane;
			uint64_t tb = 0; uint32_t tn = 0;
			if (t < n_tok)
			{
				uint32_t const tv = tok[t];
				if (tv >> 31)
				{
					uint32_t ls, le, lx, ds, de, dx;
					length_symbol((tv >> 16) & 0xff, ls, le, lx);
					distance_symbol(tv & 0xffff, ds, de, dx);
					uint32_t const el = L.t_ll[257 + ls], ed = L.t_d[ds];
					uint32_t const n1 = el >> 16, n2 = ed >> 16;
					tb = uint64_t(el & 0xffff) | (uint64_t(lx) << n1) | (uint64_t(ed & 0xffff) << (n1 + le)) | (uint64_t(dx) << (n1 + le + n2));
					tn = n1 + le + n2 + de;
				}
				else
				{
					uint32_t const e = L.t_ll[tv];
					tb = e & 0xffff; tn = e >> 16;
				}
			}
			else if (t == n_tok)
			{
				uint32_t const e = L.t_ll[256];
				tb = e & 0xffff; tn = e >> 16;
			}
			sink.put(tb, tn);
		}
		if (sink.bitpos) { if (0 == lane) sink.out32[sink.words] = L.stage[0]; }
		uint32_t const written = sink.words * 4 + ((sink.bitpos + 7) >> 3);
		if (0 == lane) out_len[b] = (written == size) ? written : (written | SIZE_MISMATCH);
		__syncthreads();
	}
}

// ---- the device-resident form: blocks whose bytes a kernel of the caller wrote (pvc3_bgzf_dev_*) ---------------------------------------
// k_block_lengths: the lengths of the blocks of n_bytes of input.  k_frame_blocks: every block as it lies in a BGZF file — the
// 18-byte gzip header with the BC field, the stream, CRC-32 and ISIZE (SAMv1 §4.1.1) — one behind the other, so that what
// comes back from the GPU is written to the file as it is.
__global__ void k_block_lengths(uint64_t n_bytes, uint32_t n_blocks, uint32_t *__restrict__ len)
{
	uint32_t const b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	uint64_t const lo = (uint64_t) b * PVC3_BGZF_BLOCK;
	len[b] = (uint32_t) ((n_bytes - lo < PVC3_BGZF_BLOCK) ? (n_bytes - lo) : PVC3_BGZF_BLOCK);
}

// Exclusive prefix sums of the framed sizes by one workgroup (a few thousand blocks); file_off[n_blocks] = total.
__global__ __launch_bounds__(1024) void k_frame_offsets(uint32_t const *__restrict__ out_len, uint32_t n_blocks, uint64_t *__restrict__ file_off)
{
	__shared__ uint64_t part[1024];
	__shared__ uint64_t carry;
	if (0 == threadIdx.x) carry = 0;
	__syncthreads();
	for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024)
	{
		uint32_t const b = b0 + threadIdx.x;
		uint64_t const v = (b < n_blocks) ? (uint64_t) (18 + (out_len[b] & ~SIZE_MISMATCH) + 8) : 0;
		part[threadIdx.x] = v;
		__syncthreads();
		for (uint32_t d = 1; d < 1024; d <<= 1)
		{
			uint64_t const t = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0;
			__syncthreads();
			part[threadIdx.x] += t;
			__syncthreads();
		}
		if (b < n_blocks) file_off[b] = carry + part[threadIdx.x] - v;
		__syncthreads();
		if (1023 == threadIdx.x) carry += part[1023];
		__syncthreads();
	}
	if (0 == threadIdx.x) file_off[n_blocks] = carry;
}

__global__ __launch_bounds__(256) void k_frame_blocks(uint8_t const *__restrict__ streams, uint32_t const *__restrict__ out_len, uint32_t const *__restrict__ in_len, uint32_t const *__restrict__ crc,
	uint64_t const *__restrict__ file_off, uint32_t n_blocks, uint8_t *__restrict__ file)
{
	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
	{
		uint32_t const clen = out_len[b] & ~SIZE_MISMATCH;
		uint8_t *dst = file + file_off[b];
		uint8_t const *src = streams + (size_t) b * PVC3_BGZF_OUT_STRIDE;
		uint32_t const bsize = 18 + clen + 8;
		if (threadIdx.x < 18)
		{
			uint8_t const head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t) ((bsize - 1) & 0xff), (uint8_t) ((bsize - 1) >> 8)};
			dst[threadIdx.x] = head[threadIdx.x];
		}
		else if (threadIdx.x < 26)
		{
			uint32_t const k = threadIdx.x - 18;
			uint32_t const v = (k < 4) ? crc[b] : in_len[b];
			dst[18 + clen + k] = (uint8_t) (v >> (8 * (k & 3)));
		}
		for (uint32_t x = threadIdx.x; x < clen; x += 256) dst[18 + x] = src[x];
	}
}

// ---- host ------------------------------------------------------------------------------------------------------------------------

int fail(int code, char const *what, hipError_t e = hipSuccess)
{
	if (hipSuccess != e) std::snprintf(g_bgzf_error, sizeof(g_bgzf_error), "%s: %s", what, hipGetErrorString(e));
	else std::snprintf(g_bgzf_error, sizeof(g_bgzf_error), "%s", what);
	return code;
}

enum { E_INVALID = -1, E_NOMEM = -2, E_HIP = -3, E_NO_DEVICE = -4, E_STATE = -6 };		// the PVC3_E_* values of pvc3.h

#define HIP_TRY(expr, what) do { hipError_t const e_ = (expr); if (hipSuccess != e_) return fail(E_HIP, what, e_); } while (0)

}

struct pvc3_bgzf
{
	int device{};
	uint32_t max_blocks{}, n_slots{}, grid_max{};
	hipStream_t stream{};
	uint8_t *d_in{}, *d_out{};
	uint32_t *d_len{}, *d_out_len{}, *d_tokens{};
	struct slot
	{
		uint8_t *h_in{}, *h_out{};
		uint32_t *h_len{}, *h_out_len{};
		hipEvent_t start{}, stop{}, done{};
		uint32_t n{};
		bool busy{};
	} slots[4];
};

extern "C" {

char const *pvc3_bgzf_last_error(void) { return g_bgzf_error; }

void pvc3_bgzf_destroy(pvc3_bgzf *h)
{
	if (!h) return;
	(void) hipSetDevice(h->device);
	if (h->stream) (void) hipStreamSynchronize(h->stream);
	for (auto &s : h->slots)
	{
		if (s.h_in) (void) hipHostFree(s.h_in);
		if (s.h_out) (void) hipHostFree(s.h_out);
		if (s.h_len) (void) hipHostFree(s.h_len);
		if (s.h_out_len) (void) hipHostFree(s.h_out_len);
		if (s.start) (void) hipEventDestroy(s.start);
		if (s.stop) (void) hipEventDestroy(s.stop);
		if (s.done) (void) hipEventDestroy(s.done);
	}
	if (h->d_in) (void) hipFree(h->d_in);
	if (h->d_out) (void) hipFree(h->d_out);
	if (h->d_len) (void) hipFree(h->d_len);
	if (h->d_out_len) (void) hipFree(h->d_out_len);
	if (h->d_tokens) (void) hipFree(h->d_tokens);
	if (h->stream) (void) hipStreamDestroy(h->stream);
	delete h;
}

int pvc3_bgzf_create(int device, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf **out)
{
	if (!out || 0 == max_blocks || max_blocks > (1u << 20) || 0 == n_slots || n_slots > 4) return fail(E_INVALID, "pvc3_bgzf_create: bad argument");
	*out = nullptr;
	int count = 0;
	if (hipSuccess != hipGetDeviceCount(&count) || device < 0 || device >= count) return fail(E_NO_DEVICE, "pvc3_bgzf_create: no such GPU");
	HIP_TRY(hipSetDevice(device), "hipSetDevice");
	pvc3_bgzf *h = new (std::nothrow) pvc3_bgzf;
	if (!h) return fail(E_NOMEM, "out of memory");
	h->device = device; h->max_blocks = max_blocks; h->n_slots = n_slots;
	int rc = 0;
	auto check = [&](hipError_t e, char const *what) { if (0 == rc && hipSuccess != e) rc = fail((hipErrorOutOfMemory == e) ? E_NOMEM : E_HIP, what, e); };
	hipDeviceProp_t prop;
	check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
	if (0 == rc)
	{
		// One wave per block, as many waves as the LDS of the chip holds (sizeof(lds_t) each): more blocks than that take turns.
		int per_cu = 0;
		check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate_blocks, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		if (per_cu < 1) per_cu = 1;
		h->grid_max = (uint32_t) prop.multiProcessorCount * (uint32_t) per_cu;
		if (h->grid_max > max_blocks) h->grid_max = max_blocks;
	}
	check(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate");
	size_t const in_bytes = (size_t) max_blocks * PVC3_BGZF_IN_STRIDE, out_bytes = (size_t) max_blocks * PVC3_BGZF_OUT_STRIDE;
	if (0 == rc) check(hipMalloc((void **) &h->d_in, in_bytes), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_out, out_bytes), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_len, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_out_len, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_tokens, (size_t) h->grid_max * TOKENS_PER_WAVE * sizeof(uint32_t)), "hipMalloc");
	for (uint32_t i = 0; i < n_slots && 0 == rc; ++i)
	{
		auto &s = h->slots[i];
		check(hipHostMalloc((void **) &s.h_in, in_bytes, hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_out, out_bytes, hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_len, max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_out_len, max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipEventCreate(&s.start), "hipEventCreate");
		if (0 == rc) check(hipEventCreate(&s.stop), "hipEventCreate");
		if (0 == rc) check(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
	}
	if (rc) { char keep[sizeof(g_bgzf_error)]; std::memcpy(keep, g_bgzf_error, sizeof(keep)); pvc3_bgzf_destroy(h); std::memcpy(g_bgzf_error, keep, sizeof(keep)); return rc; }
	*out = h;
	return 0;
}

uint32_t pvc3_bgzf_max_blocks(pvc3_bgzf const *h) { return h ? h->max_blocks : 0; }
uint8_t *pvc3_bgzf_slot_input(pvc3_bgzf *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].h_in : nullptr; }
uint32_t *pvc3_bgzf_slot_lengths(pvc3_bgzf *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].h_len : nullptr; }

int pvc3_bgzf_submit(pvc3_bgzf *h, uint32_t slot, uint32_t n_blocks)
{
	if (!h || slot >= h->n_slots || 0 == n_blocks || n_blocks > h->max_blocks) return fail(E_INVALID, "pvc3_bgzf_submit: bad argument");
	auto &s = h->slots[slot];
	if (s.busy) return fail(E_STATE, "pvc3_bgzf_submit: the slot has not been waited for");
	for (uint32_t i = 0; i < n_blocks; ++i)
		if (s.h_len[i] > PVC3_BGZF_BLOCK) return fail(E_INVALID, "pvc3_bgzf_submit: a block of more than 65280 bytes");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	// The device buffers are shared by the slots: one stream keeps the groups in order.
	HIP_TRY(hipMemcpyAsync(h->d_in, s.h_in, (size_t) n_blocks * PVC3_BGZF_IN_STRIDE, hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(h->d_len, s.h_len, n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream), "hipMemcpyAsync");
	uint32_t const grid = (n_blocks < h->grid_max) ? n_blocks : h->grid_max;
	HIP_TRY(hipEventRecord(s.start, h->stream), "hipEventRecord");
	hipLaunchKernelGGL(k_deflate_blocks, dim3(grid), dim3(WAVE), 0, h->stream, h->d_in, h->d_len, h->d_out, h->d_out_len, h->d_tokens, n_blocks);
	HIP_TRY(hipGetLastError(), "k_deflate_blocks");
	HIP_TRY(hipEventRecord(s.stop, h->stream), "hipEventRecord");
	HIP_TRY(hipMemcpyAsync(s.h_out_len, h->d_out_len, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.h_out, h->d_out, (size_t) n_blocks * PVC3_BGZF_OUT_STRIDE, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.done, h->stream), "hipEventRecord");
	s.n = n_blocks; s.busy = true;
	return 0;
}

int pvc3_bgzf_wait(pvc3_bgzf *h, uint32_t slot, uint8_t const **out, uint32_t const **out_len, float *kernel_ms)
{
	if (!h || slot >= h->n_slots) return fail(E_INVALID, "pvc3_bgzf_wait: bad argument");
	auto &s = h->slots[slot];
	if (!s.busy) return fail(E_STATE, "pvc3_bgzf_wait: nothing was submitted");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	s.busy = false;
	HIP_TRY(hipEventSynchronize(s.done), "hipEventSynchronize");
	if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s.start, s.stop), "hipEventElapsedTime");
	for (uint32_t i = 0; i < s.n; ++i)
		if (s.h_out_len[i] & SIZE_MISMATCH) return fail(E_STATE, "k_deflate_blocks: a stream is not as long as computed");
	if (out) *out = s.h_out;
	if (out_len) *out_len = s.h_out_len;
	return 0;
}

int pvc3_bgzf_deflate(pvc3_bgzf *h, uint8_t const *src, size_t n, uint32_t piece, uint8_t *dst, uint32_t *dst_len)
{
	if (!h || (!src && n) || !dst || !dst_len || 0 == piece || piece > PVC3_BGZF_BLOCK) return fail(E_INVALID, "pvc3_bgzf_deflate: bad argument");
	size_t const n_pieces = (n + piece - 1) / piece;
	for (size_t first = 0; first < n_pieces; first += h->max_blocks)
	{
		uint32_t const count = (uint32_t) ((n_pieces - first < h->max_blocks) ? (n_pieces - first) : h->max_blocks);
		auto &s = h->slots[0];
		for (uint32_t i = 0; i < count; ++i)
		{
			size_t const off = (first + i) * piece;
			uint32_t const len = (uint32_t) ((n - off < piece) ? (n - off) : piece);
			std::memcpy(s.h_in + (size_t) i * PVC3_BGZF_IN_STRIDE + PVC3_BGZF_IN_PAD, src + off, len);
			s.h_len[i] = len;
		}
		int rc = pvc3_bgzf_submit(h, 0, count);
		if (rc) return rc;
		uint8_t const *o; uint32_t const *ol;
		rc = pvc3_bgzf_wait(h, 0, &o, &ol, nullptr);
		if (rc) return rc;
		for (uint32_t i = 0; i < count; ++i)
		{
			std::memcpy(dst + (first + i) * PVC3_BGZF_OUT_STRIDE, o + (size_t) i * PVC3_BGZF_OUT_STRIDE, ol[i]);
			dst_len[first + i] = ol[i];
		}
	}
	return 0;
}

// ---- pvc3_bgzf_dev_* -------------------------------------------------------------------------------------------------------------------
}	// extern "C"

struct pvc3_bgzf_dev
{
	int device{};
	uint32_t max_blocks{}, n_slots{}, grid_max{};
	hipStream_t stream{};
	uint8_t *d_out{};
	uint32_t *d_len{}, *d_out_len{}, *d_tokens{}, *d_crc{};
	uint64_t *d_file_off{};
	pvc3_crc::tables *d_crc_tables{};
	size_t file_cap{}, stage_cap{};		// the framed blocks of a slot on the device at most; the page-locked bytes they are fetched through
	struct slot
	{
		uint8_t *d_in{};				// the caller's kernels write here
		uint8_t *d_file{};				// the framed blocks on the device (a slot's own: they are fetched when its sizes are known, which may be after the next slot was submitted)
		uint8_t *h_file{};				// page-locked: the framed blocks
		uint32_t *h_out_len{};
		uint64_t *h_file_off{};
		hipEvent_t start{}, stop{}, sizes{}, done{};
		uint32_t n{};
		uint64_t total{};
		bool busy{};
	} slots[4];
};

extern "C" {

void pvc3_bgzf_dev_destroy(pvc3_bgzf_dev *h)
{
	if (!h) return;
	(void) hipSetDevice(h->device);
	if (h->stream) (void) hipStreamSynchronize(h->stream);
	for (auto &s : h->slots)
	{
		for (hipEvent_t e : {s.start, s.stop, s.sizes, s.done}) if (e) (void) hipEventDestroy(e);
		if (s.d_in) (void) hipFree(s.d_in);
		if (s.d_file) (void) hipFree(s.d_file);
		if (s.h_file) (void) hipHostFree(s.h_file);
		if (s.h_out_len) (void) hipHostFree(s.h_out_len);
		if (s.h_file_off) (void) hipHostFree(s.h_file_off);
	}
	for (void *p : {(void *) h->d_out, (void *) h->d_len, (void *) h->d_out_len, (void *) h->d_tokens, (void *) h->d_crc, (void *) h->d_file_off, (void *) h->d_crc_tables})
		if (p) (void) hipFree(p);
	if (h->stream) (void) hipStreamDestroy(h->stream);
	delete h;
}

int pvc3_bgzf_dev_create(int device, uint32_t max_blocks, uint32_t n_slots, pvc3_bgzf_dev **out)
{
	if (!out || 0 == max_blocks || max_blocks > (1u << 20) || 0 == n_slots || n_slots > 4) return fail(E_INVALID, "pvc3_bgzf_dev_create: bad argument");
	*out = nullptr;
	int count = 0;
	if (hipSuccess != hipGetDeviceCount(&count) || device < 0 || device >= count) return fail(E_NO_DEVICE, "pvc3_bgzf_dev_create: no such GPU");
	HIP_TRY(hipSetDevice(device), "hipSetDevice");
	pvc3_bgzf_dev *h = new (std::nothrow) pvc3_bgzf_dev;
	if (!h) return fail(E_NOMEM, "out of memory");
	h->device = device; h->max_blocks = max_blocks; h->n_slots = n_slots;
	int rc = 0;
	auto check = [&](hipError_t e, char const *what) { if (0 == rc && hipSuccess != e) rc = fail((hipErrorOutOfMemory == e) ? E_NOMEM : E_HIP, what, e); };
	hipDeviceProp_t prop;
	check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
	if (0 == rc)
	{
		int per_cu = 0;
		check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate_blocks, WAVE, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
		if (per_cu < 1) per_cu = 1;
		h->grid_max = (uint32_t) prop.multiProcessorCount * (uint32_t) per_cu;
		if (h->grid_max > max_blocks) h->grid_max = max_blocks;
	}
	check(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate");
	size_t const in_bytes = (size_t) max_blocks * PVC3_BGZF_IN_STRIDE, out_bytes = (size_t) max_blocks * PVC3_BGZF_OUT_STRIDE;
	h->file_cap = (size_t) max_blocks * (PVC3_BGZF_BLOCK + 5 + 26);		// a block that does not shrink is stored: five bytes more
	// BAM records shrink to a third: the page-locked memory the bytes come back through is sized for that and a margin (what does
	// not fit one fetch takes another), since page-locked memory costs the process time when it is made and when it ends
	h->stage_cap = (h->file_cap / 2 > (size_t(16) << 20)) ? h->file_cap / 2 : ((h->file_cap < (size_t(16) << 20)) ? h->file_cap : (size_t(16) << 20));
	if (0 == rc) check(hipMalloc((void **) &h->d_out, out_bytes), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_len, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_out_len, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_crc, max_blocks * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_file_off, ((size_t) max_blocks + 1) * sizeof(uint64_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_tokens, (size_t) h->grid_max * TOKENS_PER_WAVE * sizeof(uint32_t)), "hipMalloc");
	if (0 == rc) check(hipMalloc((void **) &h->d_crc_tables, sizeof(pvc3_crc::tables)), "hipMalloc");
	if (0 == rc) { pvc3_crc::tables const t = pvc3_crc::make_tables(); check(hipMemcpy(h->d_crc_tables, &t, sizeof(t), hipMemcpyHostToDevice), "hipMemcpy"); }
	for (uint32_t i = 0; i < n_slots && 0 == rc; ++i)
	{
		auto &s = h->slots[i];
		check(hipMalloc((void **) &s.d_in, in_bytes), "hipMalloc");
		if (0 == rc) check(hipMalloc((void **) &s.d_file, h->file_cap), "hipMalloc");
		if (0 == rc) check(hipMemset(s.d_in, 0, in_bytes), "hipMemset");		// (the pads in front of the blocks are read)
		if (0 == rc) check(hipHostMalloc((void **) &s.h_file, h->stage_cap, hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_out_len, max_blocks * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipHostMalloc((void **) &s.h_file_off, ((size_t) max_blocks + 1) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc");
		if (0 == rc) check(hipEventCreate(&s.start), "hipEventCreate");
		if (0 == rc) check(hipEventCreate(&s.stop), "hipEventCreate");
		if (0 == rc) check(hipEventCreateWithFlags(&s.sizes, hipEventDisableTiming), "hipEventCreate");
		if (0 == rc) check(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
	}
	if (rc) { char keep[sizeof(g_bgzf_error)]; std::memcpy(keep, g_bgzf_error, sizeof(keep)); pvc3_bgzf_dev_destroy(h); std::memcpy(g_bgzf_error, keep, sizeof(keep)); return rc; }
	*out = h;
	return 0;
}

void *pvc3_bgzf_dev_input(pvc3_bgzf_dev *h, uint32_t slot) { return (h && slot < h->n_slots) ? h->slots[slot].d_in : nullptr; }
uint64_t pvc3_bgzf_dev_capacity(pvc3_bgzf_dev const *h) { return h ? (uint64_t) h->max_blocks * PVC3_BGZF_BLOCK : 0; }

int pvc3_bgzf_dev_submit(pvc3_bgzf_dev *h, uint32_t slot, uint64_t n_bytes)
{
	if (!h || slot >= h->n_slots || 0 == n_bytes || n_bytes > (uint64_t) h->max_blocks * PVC3_BGZF_BLOCK) return fail(E_INVALID, "pvc3_bgzf_dev_submit: bad argument");
	auto &s = h->slots[slot];
	if (s.busy) return fail(E_STATE, "pvc3_bgzf_dev_submit: the slot has not been waited for");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	uint32_t const n_blocks = (uint32_t) ((n_bytes + PVC3_BGZF_BLOCK - 1) / PVC3_BGZF_BLOCK);
	uint32_t const grid = (n_blocks < h->grid_max) ? n_blocks : h->grid_max;
	hipLaunchKernelGGL(k_block_lengths, dim3((n_blocks + 255) / 256), dim3(256), 0, h->stream, n_bytes, n_blocks, h->d_len);
	HIP_TRY(hipEventRecord(s.start, h->stream), "hipEventRecord");
	hipLaunchKernelGGL(k_deflate_blocks, dim3(grid), dim3(WAVE), 0, h->stream, s.d_in, h->d_len, h->d_out, h->d_out_len, h->d_tokens, n_blocks);
	HIP_TRY(hipGetLastError(), "k_deflate_blocks");
	HIP_TRY(hipEventRecord(s.stop, h->stream), "hipEventRecord");
	hipLaunchKernelGGL(pvc3_crc::k_crc32_ranges, dim3((n_blocks < 4096) ? n_blocks : 4096), dim3(64), 0, h->stream, s.d_in, nullptr, h->d_len, PVC3_BGZF_IN_STRIDE, PVC3_BGZF_IN_PAD, n_blocks, h->d_crc_tables, h->d_crc);
	hipLaunchKernelGGL(k_frame_offsets, dim3(1), dim3(1024), 0, h->stream, h->d_out_len, n_blocks, h->d_file_off);
	hipLaunchKernelGGL(k_frame_blocks, dim3((n_blocks < 2048) ? n_blocks : 2048), dim3(256), 0, h->stream, h->d_out, h->d_out_len, h->d_len, h->d_crc, h->d_file_off, n_blocks, s.d_file);
	HIP_TRY(hipGetLastError(), "k_frame_blocks");
	HIP_TRY(hipMemcpyAsync(s.h_out_len, h->d_out_len, n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipMemcpyAsync(s.h_file_off, h->d_file_off, ((size_t) n_blocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
	HIP_TRY(hipEventRecord(s.sizes, h->stream), "hipEventRecord");
	s.n = n_blocks; s.busy = true;
	return 0;
}

int pvc3_bgzf_dev_wait(pvc3_bgzf_dev *h, uint32_t slot, uint64_t *n_file_bytes, float *kernel_ms)
{
	if (!h || slot >= h->n_slots || !n_file_bytes) return fail(E_INVALID, "pvc3_bgzf_dev_wait: bad argument");
	auto &s = h->slots[slot];
	if (!s.busy) return fail(E_STATE, "pvc3_bgzf_dev_wait: nothing was submitted");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	s.busy = false;
	HIP_TRY(hipEventSynchronize(s.sizes), "hipEventSynchronize");
	if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, s.start, s.stop), "hipEventElapsedTime");
	for (uint32_t i = 0; i < s.n; ++i)
		if (s.h_out_len[i] & SIZE_MISMATCH) return fail(E_STATE, "k_deflate_blocks: a stream is not as long as computed");
	s.total = s.h_file_off[s.n];
	if (s.total > h->file_cap) return fail(E_STATE, "pvc3_bgzf_dev_wait: the framed blocks exceed their buffer");
	*n_file_bytes = s.total;
	return 0;
}

int pvc3_bgzf_dev_fetch(pvc3_bgzf_dev *h, uint32_t slot, uint64_t offset, uint8_t const **bytes, uint64_t *n_bytes)
{
	if (!h || slot >= h->n_slots || !bytes || !n_bytes) return fail(E_INVALID, "pvc3_bgzf_dev_fetch: bad argument");
	auto &s = h->slots[slot];
	if (s.busy || offset > s.total) return fail(E_STATE, "pvc3_bgzf_dev_fetch: nothing to fetch there");
	HIP_TRY(hipSetDevice(h->device), "hipSetDevice");
	uint64_t const n = (s.total - offset < h->stage_cap) ? (s.total - offset) : h->stage_cap;
	if (n)
	{
		HIP_TRY(hipMemcpyAsync(s.h_file, s.d_file + offset, n, hipMemcpyDeviceToHost, h->stream), "hipMemcpyAsync");
		HIP_TRY(hipEventRecord(s.done, h->stream), "hipEventRecord");
		HIP_TRY(hipEventSynchronize(s.done), "hipEventSynchronize");
	}
	*bytes = s.h_file; *n_bytes = n;
	return 0;
}

}
