// kernels.hip — gfx950 kernels of the project_alignments hot path.
//
//   k_project_fast / k_project_mid / k_project_general   project_*.hip (projection + mate projection)
//   k_pack_text4         4-bit copy of a destination text
//   k_reduce_stats       per-wave statistics -> counters
//   k_realign            realign.hip
//   k_project_positions  one thread per position: project_position (msa_index.hh:253-258)
//   k_scan_* / k_offsets_gather   pack the per-record results in record order.
//
// Integer / bit work, HBM- and latency-bound; no MFMA on this path.
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

__global__ __launch_bounds__(256) void k_project_positions(
	dev_entry const *__restrict__ entries, uint32_t n_entries, uint64_t n,
	uint32_t const *__restrict__ src_entry, uint32_t const *__restrict__ dst_entry, uint32_t const *__restrict__ pos, uint32_t *__restrict__ out
)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
	{
		uint32_t const s = src_entry[i], d = dst_entry[i];
		out[i] = (s < n_entries && d < n_entries) ? thread_project_position(entries[s], entries[d], d, pos[i]) : PVC3_NO_MATE;
	}
}


// The kernels read cigar + cigar_off, seq4 + seq_off and qual + qual_off without further checks; a batch whose offsets decrease,
// run past the arrays or leave a record fewer SEQ / QUAL bytes than its l_seq promises is refused as a whole.
__global__ __launch_bounds__(256) void k_validate_batch(dev_batch_in const in, uint64_t n_cigar, uint64_t n_seq, uint64_t n_qual, uint64_t const *__restrict__ qual_off, uint32_t *first_bad)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += stride)
	{
		uint64_t const c0 = in.cigar_off[i], c1 = in.cigar_off[i + 1], s0 = in.seq_off[i], s1 = in.seq_off[i + 1];
		uint32_t const l_seq = in.l_seq[i];
		bool bad = (c0 > c1) || (c1 > n_cigar) || (s0 > s1) || (s1 > n_seq) || (s1 - s0 < ((uint64_t) l_seq + 1) / 2);
		if (qual_off && (in.flags[i] & PVC3_FLAG_HAS_QUAL))
		{
			uint64_t const q0 = qual_off[i], q1 = qual_off[i + 1];
			bad = bad || (q0 > q1) || (q1 > n_qual) || (q1 - q0 < l_seq);
		}
		if (bad) atomicMin(first_bad, (uint32_t) i);
	}
}


// ---- packing: exclusive scans of the per-record counts, then a gather into record order ------------------------------
// A tile is 1024 consecutive records; thread t of the tile's block owns records 4 t .. 4 t + 3, so that a thread's counts and
// offsets are single 16-byte (32-byte) accesses, everything a thread needs is requested before anything is used, and
// neighbouring threads copy neighbouring pieces of the arenas.
constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_ITEMS = 4;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr uint32_t GATHER_STAGE = 8192;		// uint32 of LDS staging in k_offsets_gather (8 operations per record on average; more: direct copies)

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2_t;

// The counts of a thread's four records: one 16-byte load inside the array (it is a multiple of 4 long only by accident).
__device__ __forceinline__ u32x4_t load_counts(uint32_t const *__restrict__ a, uint64_t first, uint64_t n)
{
	if (first + SCAN_ITEMS <= n)
		return *(u32x4_t const *) (a + first);
	u32x4_t v{0, 0, 0, 0};
	if (first < n) v.x = a[first];
	if (first + 1 < n) v.y = a[first + 1];
	if (first + 2 < n) v.z = a[first + 2];
	return v;
}

__device__ __forceinline__ uint64_t block_reduce_add(uint64_t v, uint64_t *lds)
{
	for (int off = 32; off; off >>= 1)
		v += __shfl_down(v, off);
	uint32_t const w = threadIdx.x >> 6;
	if (0 == (threadIdx.x & 63)) lds[w] = v;
	__syncthreads();
	uint64_t total = 0;
	for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) total += lds[i];
	__syncthreads();
	return total;
}

// A block sums SUPER_TILES consecutive tiles: tile_sums[2 t + k] = sum of counts k (0 = cigar, 1 = runs) over the tiles of the
// same super-tile before tile t, super_sums[2 s + k] = sum over super-tile s.  (Two levels keep the single-block scan below short:
// 4 883 entries for 40 M records.)
constexpr uint32_t SUPER_TILES = 8;
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(uint64_t n, uint64_t n_tiles, uint32_t const *__restrict__ a, uint32_t const *__restrict__ b, uint64_t *__restrict__ tile_sums, uint64_t *__restrict__ super_sums)
{
	__shared__ uint64_t lds[SCAN_THREADS / 64][2 * SUPER_TILES];
	uint64_t const tile0 = (uint64_t) blockIdx.x * SUPER_TILES;
	u32x4_t va[SUPER_TILES], vb[SUPER_TILES];
#pragma unroll
	for (uint32_t j = 0; j < SUPER_TILES; ++j)
	{
		uint64_t const first = (tile0 + j) * SCAN_TILE + SCAN_ITEMS * threadIdx.x;
		va[j] = load_counts(a, first, n); vb[j] = load_counts(b, first, n);
	}
	uint32_t const lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
	for (uint32_t j = 0; j < SUPER_TILES; ++j)
	{
		uint64_t sa = (uint64_t) va[j].x + va[j].y + va[j].z + va[j].w, sb = (uint64_t) vb[j].x + vb[j].y + vb[j].z + vb[j].w;
		for (int off = 32; off; off >>= 1) { sa += __shfl_xor(sa, off); sb += __shfl_xor(sb, off); }
		if (0 == lane) { lds[w][2 * j] = sa; lds[w][2 * j + 1] = sb; }
	}
	__syncthreads();
	if (threadIdx.x < 2)
	{
		uint64_t run = 0;
		for (uint32_t j = 0; j < SUPER_TILES; ++j)
		{
			uint64_t t = 0;
			for (uint32_t x = 0; x < SCAN_THREADS / 64; ++x) t += lds[x][2 * j + threadIdx.x];
			if (tile0 + j < n_tiles) tile_sums[2 * (tile0 + j) + threadIdx.x] = run;
			run += t;
		}
		super_sums[2 * (uint64_t) blockIdx.x + threadIdx.x] = run;
	}
}

// In-place exclusive scan of the tile sums by one block of four wavefronts (a block of sixteen waits for a nearly empty CU when other kernels' waves are resident: next to the BGZF codecs of the command line it took 0.8 ms instead of 0.01): wave w owns a contiguous range of tiles and walks
// it 64 tiles at a time (coalesced); first the ranges' totals, then the scan with the carry of the ranges before.
constexpr uint32_t TS_WAVES = 4;
__global__ __launch_bounds__(64 * TS_WAVES) void k_scan_tile_sums(uint64_t n_tiles, uint64_t *__restrict__ tile_sums, uint64_t *__restrict__ totals)
{
	typedef unsigned long long __attribute__((ext_vector_type(2))) pair_t;
	__shared__ uint64_t wave_a[TS_WAVES], wave_b[TS_WAVES];
	uint32_t const lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint64_t const per = ((n_tiles + TS_WAVES - 1) / TS_WAVES + 63) / 64 * 64;		// tiles per wave, a multiple of 64
	uint64_t const begin = per * w, end = (begin + per < n_tiles) ? (begin + per) : n_tiles;
	pair_t const *in = (pair_t const *) tile_sums;
	pair_t *out = (pair_t *) tile_sums;
	uint64_t sa = 0, sb = 0;
	for (uint64_t i = begin + lane; i < end; i += 64) { pair_t const v = in[i]; sa += v.x; sb += v.y; }
	for (int off = 32; off; off >>= 1) { sa += __shfl_xor(sa, off); sb += __shfl_xor(sb, off); }
	if (0 == lane) { wave_a[w] = sa; wave_b[w] = sb; }
	__syncthreads();
	uint64_t ca = 0, cb = 0, ta = 0, tb = 0;
	for (uint32_t i = 0; i < TS_WAVES; ++i)
	{
		ca += (i < w) ? wave_a[i] : 0; cb += (i < w) ? wave_b[i] : 0;
		ta += wave_a[i]; tb += wave_b[i];
	}
	for (uint64_t base = begin; base < end; base += 64)
	{
		uint64_t const i = base + lane;
		pair_t v{0, 0};
		if (i < end) v = in[i];
		uint64_t ia = v.x, ib = v.y;
#pragma unroll
		for (uint32_t off = 1; off < 64; off <<= 1)
		{
			uint64_t const xa = __shfl_up(ia, off), xb = __shfl_up(ib, off);
			if (lane >= off) { ia += xa; ib += xb; }
		}
		if (i < end) out[i] = pair_t{ca + ia - v.x, cb + ib - v.y};
		ca += __shfl(ia, 63); cb += __shfl(ib, 63);
	}
	if (0 == threadIdx.x) { totals[0] = ta; totals[1] = tb; }
}

// Exclusive prefix of v over the block's threads (v of thread t counts before those of thread t + 1); `total` = block sum.
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *lds, uint64_t &total)
{
	uint32_t const lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint64_t inc = v;
#pragma unroll
	for (uint32_t off = 1; off < 64; off <<= 1)
	{
		uint64_t const t = __shfl_up(inc, off);
		if (lane >= off) inc += t;
	}
	if (63 == lane) lds[w] = inc;
	__syncthreads();
	uint64_t before = 0;
	total = 0;
	for (uint32_t i = 0; i < (SCAN_THREADS >> 6); ++i)
	{
		uint64_t const x = lds[i];
		before += (i < w) ? x : 0;
		total += x;
	}
	__syncthreads();
	return before + inc - v;
}

// Offsets of every record's operations / runs in the packed arrays (exclusive scan, tile prefix from k_scan_tile_sums) and,
// in the same pass, the copy of the operations and runs out of the arenas.  (The status output is the projection kernels'
// flag array itself: every record ends with its PVC3_ST_* value there.)
__global__ __launch_bounds__(SCAN_THREADS) void k_offsets_gather(uint64_t n, dev_batch_work const W, dev_batch_packed const P, uint64_t const *__restrict__ tile_sums, uint64_t const *__restrict__ super_sums, uint64_t const *__restrict__ totals)
{
	__shared__ uint64_t lds[8];
	__shared__ uint32_t stage[GATHER_STAGE];		// the operations of the tile, in output order
	uint64_t const first = (uint64_t) blockIdx.x * SCAN_TILE + SCAN_ITEMS * threadIdx.x;
	uint64_t const carry_a = tile_sums[2 * (uint64_t) blockIdx.x] + super_sums[2 * (uint64_t) (blockIdx.x / SUPER_TILES)];
	uint64_t const carry_b = tile_sums[2 * (uint64_t) blockIdx.x + 1] + super_sums[2 * (uint64_t) (blockIdx.x / SUPER_TILES) + 1];
	bool const whole = first + SCAN_ITEMS <= n;
	u32x4_t const vc = load_counts(W.cig_n, first, n), vr = load_counts(W.run_n, first, n);
	uint32_t const nc[SCAN_ITEMS] = {vc.x, vc.y, vc.z, vc.w}, nr[SCAN_ITEMS] = {vr.x, vr.y, vr.z, vr.w};
	uint64_t loc[SCAN_ITEMS] = {0, 0, 0, 0};
	if (whole)
	{
		u64x2_t const l0 = *(u64x2_t const *) (W.cig_loc + first), l1 = *(u64x2_t const *) (W.cig_loc + first + 2);
		loc[0] = l0.x; loc[1] = l0.y; loc[2] = l1.x; loc[3] = l1.y;
	}
	else
		for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
			if (first + k < n) loc[k] = W.cig_loc[first + k];
	// The first four operations of every record with one 16-byte load (the arena is padded): most records have no more.
	u32x4_t head[SCAN_ITEMS];
#pragma unroll
	for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
	{
		head[k] = u32x4_t{0, 0, 0, 0};
		if (nc[k]) __builtin_memcpy(&head[k], W.arena + loc[k], 16);
	}

	uint64_t total_a, total_b;
	uint64_t const mine_a = (uint64_t) nc[0] + nc[1] + nc[2] + nc[3], mine_b = (uint64_t) nr[0] + nr[1] + nr[2] + nr[3];
	uint64_t const local_a = block_exclusive_scan(mine_a, lds, total_a);
	uint64_t const local_b = block_exclusive_scan(mine_b, lds, total_b);
	uint64_t off_a[SCAN_ITEMS + 1], off_b[SCAN_ITEMS + 1];
	off_a[0] = carry_a + local_a; off_b[0] = carry_b + local_b;
#pragma unroll
	for (uint32_t k = 0; k < SCAN_ITEMS; ++k) { off_a[k + 1] = off_a[k] + nc[k]; off_b[k + 1] = off_b[k] + nr[k]; }
	if (whole)
	{
		*(u64x2_t *) (P.out_cigar_off + first) = u64x2_t{off_a[0], off_a[1]}; *(u64x2_t *) (P.out_cigar_off + first + 2) = u64x2_t{off_a[2], off_a[3]};
		*(u64x2_t *) (P.run_off + first) = u64x2_t{off_b[0], off_b[1]}; *(u64x2_t *) (P.run_off + first + 2) = u64x2_t{off_b[2], off_b[3]};
	}
	else
		for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
			if (first + k < n) { P.out_cigar_off[first + k] = off_a[k]; P.run_off[first + k] = off_b[k]; }

	// The operations of the tile's records form one contiguous piece of the output: every thread drops its own into LDS,
	// then the block writes the piece with coalesced stores (a piece that does not fit is written directly).
	bool const staged = total_a <= GATHER_STAGE;
#pragma unroll
	for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
	{
		uint32_t const cnt = nc[k];
		if (0 == cnt) continue;
		uint32_t const h[4] = {head[k].x, head[k].y, head[k].z, head[k].w};
		uint32_t const *src = W.arena + loc[k];
		if (staged)
		{
			uint32_t *dst = stage + (off_a[k] - carry_a);
#pragma unroll
			for (uint32_t x = 0; x < 4; ++x) if (x < cnt) dst[x] = h[x];
			for (uint32_t x = 4; x < cnt; ++x) dst[x] = src[x];
		}
		else
		{
			uint32_t *dst = P.out_cigar + off_a[k];
#pragma unroll
			for (uint32_t x = 0; x < 4; ++x) if (x < cnt) dst[x] = h[x];
			for (uint32_t x = 4; x < cnt; ++x) dst[x] = src[x];
		}
	}
	if (mine_b)
		for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
			if (nr[k])
			{
				pvc3_run const *rs = W.runs_arena + W.run_loc[first + k];
				pvc3_run *rd = P.runs + off_b[k];
				for (uint32_t x = 0; x < nr[k]; ++x) rd[x] = rs[x];
			}
	if (staged)
	{
		__syncthreads();
		for (uint32_t j = threadIdx.x; j < (uint32_t) total_a; j += SCAN_THREADS)
			P.out_cigar[carry_a + j] = stage[j];
	}
	if (0 == blockIdx.x && 0 == threadIdx.x) { P.out_cigar_off[n] = totals[0]; P.run_off[n] = totals[1]; }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)	// splitmix64 finalizer
{
	x += UINT64_C(0x9E3779B97F4A7C15);
	x = (x ^ (x >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
	x = (x ^ (x >> 27)) * UINT64_C(0x94D049BB133111EB);
	return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_checksum(uint64_t n, uint64_t first_record, dev_batch_work const W, dev_batch_packed const P, unsigned long long *out)
{
	__shared__ uint64_t lds[8];
	uint64_t acc = 0;
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t rec = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; rec < n; rec += stride)
	{
		uint64_t h = mix64(first_record + rec);
		h = mix64(h ^ W.dst_pos[rec]);
		h = mix64(h ^ W.mate_dst_pos[rec]);
		h = mix64(h ^ (uint64_t) (uint32_t) P.status[rec]);
		uint64_t const c0 = P.out_cigar_off[rec], c1 = P.out_cigar_off[rec + 1];
		h = mix64(h ^ (c1 - c0));
		for (uint64_t i = c0; i < c1; ++i) h = mix64(h ^ P.out_cigar[i]);
		uint64_t const r0 = P.run_off[rec], r1 = P.run_off[rec + 1];
		h = mix64(h ^ (r1 - r0));
		for (uint64_t i = r0; i < r1; ++i)
		{
			pvc3_run const rr = P.runs[i];
			h = mix64(h ^ rr.query_location); h = mix64(h ^ rr.query_length); h = mix64(h ^ rr.ref_position); h = mix64(h ^ rr.ref_length);
		}
		acc += h;
	}
	acc = block_reduce_add(acc, lds);
	if (0 == threadIdx.x) atomicAdd(out, (unsigned long long) acc);
}

// 4-bit copy of a destination text for k_project_fast.  The fast path compares dna5::to_char of a BAM base code with the
// raw text byte (rewrite_cigar.hh:96-104, quirk Q3): codes 1 2 4 8 read as A C G T, every other code as N.  With the text
// encoded as A C G T N -> 1 2 4 8 15 and any other byte (lower case, IUPAC ...) -> 0, the comparison becomes
// (code in {1,2,4,8} ? code : 15) != nibble.  One thread writes one byte: bases 2b - TEXT4_FRONT and 2b + 1 - TEXT4_FRONT.
__device__ __forceinline__ uint32_t text_code(uint8_t c)
{
	return ('A' == c) ? 1u : ('C' == c) ? 2u : ('G' == c) ? 4u : ('T' == c) ? 8u : ('N' == c) ? 15u : 0u;
}

__global__ __launch_bounds__(256) void k_pack_text4(uint8_t const *__restrict__ text, uint64_t len, uint8_t *__restrict__ out)
{
	uint64_t const n_bytes = (TEXT4_FRONT + len + 1) / 2;
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; b < n_bytes; b += stride)
	{
		uint64_t const n0 = 2 * b;
		uint32_t v = 0;
		if (n0 >= TEXT4_FRONT && n0 - TEXT4_FRONT < len) v |= text_code(text[n0 - TEXT4_FRONT]);
		if (n0 + 1 >= TEXT4_FRONT && n0 + 1 - TEXT4_FRONT < len) v |= text_code(text[n0 + 1 - TEXT4_FRONT]) << 4;
		out[b] = (uint8_t) v;
	}
}

// dev_entry::text4w: 16 bytes per thread.
__global__ __launch_bounds__(256) void k_window_text4(uint8_t const *__restrict__ text4, uint64_t n_lines, uint8_t *__restrict__ out)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < 8 * n_lines; i += stride)
	{
		uint64_t const k = i >> 3, j = 16 * (i & 7);
		uint64_t v[2];
		__builtin_memcpy(v, text4 + (TEXT_WINDOW_STEP / 2) * k + j, 16);
		__builtin_memcpy(out + 128 * k + j, v, 16);
	}
}

// dev_entry::paired when the paired blocks start where the plain ones do (PAIRED_STEP == BLOCK_COLS): block b of the copy is
// block b of either row, side by side — an interleave of what is already on the device (the host build of round 2 took
// 1.3 s per context on C3).  Blocks past a row's own (the copy has up to one more) are all-gap with the row's total as rank.
__global__ __launch_bounds__(256) void k_pair_blocks(uint64_t const *__restrict__ a, uint64_t const *__restrict__ b, uint64_t n_have, uint64_t n_paired, uint64_t zeros_a, uint64_t zeros_b, uint64_t *__restrict__ out)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_paired * 2 * BLOCK_WORDS; i += stride)
	{
		uint64_t const blk = i / (2 * BLOCK_WORDS), r = (i / BLOCK_WORDS) & 1, w = i % BLOCK_WORDS;
		uint64_t v;
		if (blk < n_have) v = (r ? b : a)[blk * BLOCK_WORDS + w];
		else v = (0 == w) ? (r ? zeros_b : zeros_a) : ~UINT64_C(0);
		out[i] = v;
	}
}

// ---- position windows (dev_entry::pwin) ---------------------------------------------------------------------------------------
// Headers: one thread per window.  c0 = select0(P + 1) of the source row from the select samples, r0 = rank0 of the
// destination row there, clean = the columns from c0 on without a gap in either row (the padding past the last column is
// all gaps, so the count stops there by itself).
__global__ __launch_bounds__(256) void k_pwin_headers(dev_entry const S, dev_entry const D, uint32_t shift, uint64_t n_windows, uint8_t *__restrict__ out)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	uint64_t const wbytes = UINT64_C(2) << shift;
	row_view const sv{S.blocks, BLOCK_WORDS, BLOCK_COLS}, dv{D.blocks, BLOCK_WORDS, BLOCK_COLS};
	for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < n_windows; k += stride)
	{
		uint32_t const P = (uint32_t) (k << shift);		// < n_zeros by the number of windows
		col_pos const g = locate(BLOCK_COLS, select_guess(S.sel, P, 0).col);
		uint32_t const c0 = dir_select0(sv, P, dir_load(sv, g.blk, pair_of(g)));
		col_pos const p0 = locate(BLOCK_COLS, c0);
		uint32_t const r0 = dir_rank(dv, dir_load(dv, p0.blk, pair_of(p0)), p0);
		uint32_t clean = 0, col = c0;
		while (clean < PWIN_CLEAN_CAP)
		{
			col_pos const p = locate(BLOCK_COLS, col);
			uint64_t const wi = (uint64_t) p.blk * BLOCK_WORDS + 1 + p.w;
			uint64_t const bad = (S.blocks[wi] | D.blocks[wi]) >> p.bit;
			uint32_t const avail = 64 - p.bit;
			uint32_t t = bad ? ctz64(bad) : 64;
			t = (t < avail) ? t : avail;
			clean += t; col += t;
			if (t < avail) break;
		}
		typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
		*(u32x4 *) (out + wbytes * k) = u32x4{c0, r0, (clean < PWIN_CLEAN_CAP) ? clean : PWIN_CLEAN_CAP, 0};
	}
}

// Text: 16 bytes per thread, 32 nibbles of the destination's packed text from base r0 - PWIN_TEXT_BACK + 32 j on.
__global__ __launch_bounds__(256) void k_pwin_text(uint8_t const *__restrict__ text4, uint32_t shift, uint64_t n_windows, uint8_t *__restrict__ out)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	uint32_t const wbytes = 2u << shift, pieces = wbytes / 16 - 1, lg = shift - 3;		// 8 or 16 sixteen-byte pieces per window, the first is the header
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < (n_windows << lg); i += stride)
	{
		uint64_t const k = i >> lg;
		uint32_t const j = (uint32_t) (i & ((1u << lg) - 1));
		if (j >= pieces) continue;
		uint32_t const r0 = *(uint32_t const *) (out + (uint64_t) wbytes * k + 4);
		uint64_t const n0 = (uint64_t) r0 + TEXT4_FRONT - PWIN_TEXT_BACK + 32 * j;
		uint8_t const *p = text4 + (n0 >> 1);
		uint64_t v[2];
		__builtin_memcpy(v, p, 16);
		if (n0 & 1)
		{
			uint64_t const next = p[16];
			v[0] = (v[0] >> 4) | (v[1] << 60);
			v[1] = (v[1] >> 4) | (next << 60);
		}
		__builtin_memcpy(out + (uint64_t) wbytes * k + PWIN_HEADER + 16 * j, v, 16);
	}
}

// Folds the per-wave statistics of k_project_fast / k_project_mid / k_project_general (blockIdx.y) into the counters:
// one thread per wave row, one atomic per block and counter.
__global__ __launch_bounds__(256) void k_reduce_stats(uint32_t n_waves, dev_batch_work const W)
{
	__shared__ uint64_t lds[8];
	uint32_t const row = blockIdx.y;
	uint32_t const w = blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc[WAVE_STATS];
	for (uint32_t k = 0; k < WAVE_STATS; ++k)
		acc[k] = (w < n_waves) ? W.wave_stats[((uint64_t) row * n_waves + w) * WAVE_STATS + k] : 0;
	for (uint32_t k = 0; k < WAVE_STATS; ++k)
		acc[k] = block_reduce_add(acc[k], lds);
	if (0 == threadIdx.x)
	{
		dev_counters *c = W.counters;
		auto add = [](unsigned long long *p, uint64_t v) { if (v) atomicAdd(p, (unsigned long long) v); };
		add(&c->bitvec_bytes, acc[WS_BITVEC]);
		add(&c->ref_bytes, acc[WS_REF]);
		add(&c->n_failed, acc[WS_FAILED]);
		add(&c->n_realigned_records, acc[WS_WITH_RUNS]);
		if (0 == row) { add(&c->n_fast, acc[WS_DONE]); add(&c->fast_bytes, acc[WS_BYTES]); }
		else { add(&c->general_bytes, acc[WS_BYTES]); if (1 == row) add(&c->n_mid, acc[WS_DONE]); }
	}
}

uint32_t grid_for(uint64_t n, uint32_t per_block, uint32_t max_blocks)
{
	uint64_t const b = (n + per_block - 1) / per_block;
	return (uint32_t) (b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

}	// anonymous namespace


int launch_reduce_stats(uint64_t n, dev_batch_work const &work, void *stream)
{
	if (n)
		hipLaunchKernelGGL(k_reduce_stats, dim3((project_wave_count(n) + 255) / 256, 3), dim3(256), 0, (hipStream_t) stream, project_wave_count(n), work);
	return (int) hipGetLastError();
}

int launch_validate(dev_batch_in const &in, uint64_t n_cigar, uint64_t n_seq, uint64_t n_qual, uint64_t const *qual_off, uint32_t *first_bad, void *stream)
{
	if (in.n)
		hipLaunchKernelGGL(k_validate_batch, dim3(grid_for(in.n, 256, 256 * 16)), dim3(256), 0, (hipStream_t) stream, in, n_cigar, n_seq, n_qual, qual_off, first_bad);
	return (int) hipGetLastError();
}

int launch_pack_text4(uint8_t const *text, uint64_t len, uint8_t *out, void *stream)
{
	hipLaunchKernelGGL(k_pack_text4, dim3(grid_for((TEXT4_FRONT + len + 1) / 2, 256, 256 * 32)), dim3(256), 0, (hipStream_t) stream, text, len, out);
	return (int) hipGetLastError();
}

int launch_project_positions(dev_entry const *entries, uint32_t n_entries, uint64_t n, uint32_t const *src_entry, uint32_t const *dst_entry, uint32_t const *pos, uint32_t *out, void *stream)
{
	if (0 == n) return 0;
	hipLaunchKernelGGL(k_project_positions, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t) stream, entries, n_entries, n, src_entry, dst_entry, pos, out);
	return (int) hipGetLastError();
}

int launch_window_text4(uint8_t const *text4, uint64_t n_lines, uint8_t *out, void *stream)
{
	if (n_lines)
		hipLaunchKernelGGL(k_window_text4, dim3(grid_for(8 * n_lines, 256, 256 * 32)), dim3(256), 0, (hipStream_t) stream, text4, n_lines, out);
	return (int) hipGetLastError();
}

int launch_pair_blocks(uint64_t const *a, uint64_t const *b, uint64_t n_have, uint64_t n_paired, uint64_t zeros_a, uint64_t zeros_b, uint64_t *out, void *stream)
{
	hipLaunchKernelGGL(k_pair_blocks, dim3(grid_for(n_paired * 2 * BLOCK_WORDS, 256, 256 * 32)), dim3(256), 0, (hipStream_t) stream, a, b, n_have, n_paired, zeros_a, zeros_b, out);
	return (int) hipGetLastError();
}

int launch_build_pwin(dev_entry const &src, dev_entry const &dst, uint32_t shift, uint64_t n_windows, uint8_t *out, void *stream)
{
	if (0 == n_windows) return 0;
	hipLaunchKernelGGL(k_pwin_headers, dim3(grid_for(n_windows, 256, 256 * 32)), dim3(256), 0, (hipStream_t) stream, src, dst, shift, n_windows, out);
	hipLaunchKernelGGL(k_pwin_text, dim3(grid_for(n_windows << (shift - 3), 256, 256 * 64)), dim3(256), 0, (hipStream_t) stream, dst.text4, shift, n_windows, out);
	return (int) hipGetLastError();
}

// scan_tmp: [2 per tile][2 per super-tile][2 totals]
uint64_t pack_scan_tmp_words(uint64_t n)
{
	uint64_t const n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE, n_super = (n_tiles + SUPER_TILES - 1) / SUPER_TILES;
	return 2 * n_tiles + 2 * n_super + 2 + 2;
}

int launch_pack(uint64_t n, dev_batch_work const &work, dev_batch_packed const &packed, uint64_t *scan_tmp, void *stream)
{
	uint64_t const n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE, n_super = (n_tiles + SUPER_TILES - 1) / SUPER_TILES;
	uint64_t *super_sums = scan_tmp + 2 * n_tiles, *totals = super_sums + 2 * n_super;
	hipStream_t const s = (hipStream_t) stream;
	if (n_super)
		hipLaunchKernelGGL(k_scan_tiles, dim3((uint32_t) n_super), dim3(SCAN_THREADS), 0, s, n, n_tiles, work.cig_n, work.run_n, scan_tmp, super_sums);
	hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(64 * TS_WAVES), 0, s, n_super, super_sums, totals);
	hipLaunchKernelGGL(k_offsets_gather, dim3((uint32_t) (n_tiles ? n_tiles : 1)), dim3(SCAN_THREADS), 0, s, n, work, packed, scan_tmp, super_sums, totals);
	return (int) hipGetLastError();
}

int launch_checksum(uint64_t n, uint64_t first_record, dev_batch_work const &work, dev_batch_packed const &packed, unsigned long long *out, void *stream)
{
	(void) hipMemsetAsync(out, 0, sizeof(unsigned long long), (hipStream_t) stream);
	if (n)
		hipLaunchKernelGGL(k_checksum, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, (hipStream_t) stream, n, first_record, work, packed, out);
	return (int) hipGetLastError();
}

}
