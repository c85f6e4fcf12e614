// kernels.hip — gfx950 kernels of the project_alignments hot path.
//
//   k_project    one wavefront per record: select0 on the source vector, then a 64-columns-per-step walk over the src and
//                dst gap words; the per-lane token (= X I D) masks are run-length merged with cigar_buffer semantics.
//                Restates rewrite_cigar (include/panvc3/rewrite_cigar.hh:18-157) through the column-walk equivalence
//                 and detects indel runs on the fly (libpanvc3/indel_run_checker.cc:31-121).
//   k_mates      one thread per record: project_position of the mate (project_alignments.cc:897-901).
//   k_realign    realign.hip.
//   k_scan_* / k_gather   pack the per-record results in record order.
//
// Integer / bit work, HBM- and latency-bound; no MFMA on this path.
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

// Emits the tokens of one 64-column step in column order (bit i of m? = column i of the step).
__device__ __forceinline__ void emit_tokens(sink &s, uint64_t mE, uint64_t mX, uint64_t mI, uint64_t mD, uint32_t lane)
{
	uint64_t valid = mE | mX | mI | mD;
	if (valid == mE)
	{
		s.push(OP_EQ, popc64(mE), lane);
		return;
	}
	while (valid)
	{
		uint64_t const pb = valid & (~valid + 1);	// lowest set bit
		uint32_t op; uint64_t mt;
		if (mE & pb) { op = OP_EQ; mt = mE; }
		else if (mX & pb) { op = OP_X; mt = mX; }
		else if (mI & pb) { op = OP_I; mt = mI; }
		else { op = OP_D; mt = mD; }
		uint64_t const others = valid & ~mt;
		uint64_t const below = others ? ((others & (~others + 1)) - 1) : ~UINT64_C(0);	// columns before the next other token
		s.push(op, popc64(mt & valid & below), lane);
		valid &= ~below;
	}
}

// Input CIGAR operations, 64 at a time in one VGPR.
struct cigar_reader
{
	uint32_t v;
	uint32_t base;		// wave-uniform
	uint32_t const *ops;
	uint32_t n;

	__device__ __forceinline__ void init(uint32_t const *ops_, uint32_t n_, uint32_t lane)
	{
		ops = ops_; n = n_; base = 0;
		v = (lane < n) ? ops[lane] : 0;
	}

	__device__ __forceinline__ uint32_t get(uint32_t i, uint32_t lane)
	{
		if (i - base >= 64)
		{
			base = i & ~63u;
			v = (base + lane < n) ? ops[base + lane] : 0;
		}
		return readlane(v, i - base);
	}
};

// BAM 4-bit code -> seqan3::dna5::to_char() ("=ACMGRSVTWYHKDBN" read through dna5::assign_char).
__device__ __forceinline__ uint32_t seq4_to_dna5_char(uint32_t code)
{
	return (1 == code) ? 'A' : (2 == code) ? 'C' : (4 == code) ? 'G' : (8 == code) ? 'T' : 'N';
}

struct record_result
{
	uint32_t status;
	uint32_t dst_pos;
	uint32_t n_out;
	uint32_t has_run;
	uint64_t loc;
	uint32_t cols;			// MSA columns walked
	uint32_t ref_span;		// destination characters spanned
};

__device__ __forceinline__ void project_record(
	dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &W,
	uint64_t rec, arena_alloc &arena, uint32_t lane, record_result &res
)
{
	res.status = PVC3_ST_OK; res.dst_pos = 0; res.n_out = 0; res.has_run = 0; res.loc = 0; res.cols = 0; res.ref_span = 0;

	uint32_t const src_pos = in.src_pos[rec];
	uint32_t const src_id = in.src_entry[rec];
	uint32_t const dst_id = in.dst_entry[rec];
	uint64_t const cig_begin = in.cigar_off[rec];
	uint32_t const n_cigar = (uint32_t) (in.cigar_off[rec + 1] - cig_begin);
	uint64_t const seq_begin = in.seq_off[rec];
	uint32_t const l_seq = in.l_seq[rec];
	if (src_id >= n_entries || dst_id >= n_entries) { res.status = PVC3_ST_SRC_RANGE; return; }
	dev_entry const S = entries[src_id];
	dev_entry const D = entries[dst_id];

	cigar_reader cig;
	cig.init(in.cigar + cig_begin, n_cigar, lane);

	// Query bases: lane l keeps bytes [4 l, 4 l + 4) of the record's packed SEQ (bases [8 l, 8 l + 8)).
	uint8_t const *seq4 = in.seq4 + seq_begin;
	uint32_t seqw = 0;
	if (8 * lane < l_seq)
		seqw = load_u32_unaligned(seq4 + 4 * lane);	// the SEQ buffer is padded by the host side

	// Validation pass (defines the per-record status; same order as the oracle's batch driver).
	uint32_t ref_consumed = 0, query_consumed = 0;
	{
		bool bad_op = false;
		for (uint32_t i = 0; i < n_cigar; ++i)
		{
			uint32_t const c = cig.get(i, lane);
			uint32_t const op = c & 0xf, cnt = c >> 4;
			if (op > OP_X) { bad_op = true; break; }
			if (OP_M == op || OP_EQ == op || OP_X == op) { ref_consumed += cnt; query_consumed += cnt; }
			else if (OP_D == op || OP_N == op) ref_consumed += cnt;
			else if (OP_I == op || OP_S == op) query_consumed += cnt;
		}
		if (bad_op) { res.status = PVC3_ST_CIGAR_OP; return; }
		if (S.n_cols != D.n_cols || (uint64_t) src_pos + (ref_consumed ? ref_consumed : 1u) > S.n_zeros) { res.status = PVC3_ST_SRC_RANGE; return; }
		if (query_consumed > l_seq) { res.status = PVC3_ST_QUERY_RANGE; return; }
	}

	// aligned_position + project_aligned_position (rewrite_cigar.hh:51-56).
	window win;
	uint32_t col = wave_select0(S, D.blocks, src_pos, win, lane);
	uint32_t const col_start = col;
	uint32_t r = window_rank_dst(win, split_col(col));		// dst rank at col = prev_excess
	res.dst_pos = r;

	// Destination text window: lane l keeps bytes [ref_base + 4 l, ref_base + 4 l + 4).
	uint32_t ref_base = r & ~3u;
	uint32_t refw = 0;
	if (D.text)
		refw = *(uint32_t const *) (D.text + ref_base + 4 * lane);	// device texts are padded by TEXT_PAD bytes

	sink out;
	out.init();
	uint32_t q = 0;		// query position
	uint64_t spill_loc = 0;

	for (uint32_t ci = 0; ci < n_cigar; ++ci)
	{
		uint32_t const c = cig.get(ci, lane);
		uint32_t const op = c & 0xf;
		uint32_t n_rem = c >> 4;

		if (OP_I == op || OP_S == op)		// rewrite_cigar.hh:69-73
		{
			out.push(op, n_rem, lane);
			q += n_rem;
			continue;
		}
		if (OP_H == op || OP_P == op)		// :75-78
		{
			out.push(op, n_rem, lane);
			continue;
		}

		bool const m_type = !(OP_D == op || OP_N == op);	// M = X (:80-118) vs D N (:121-147)
		while (n_rem)
		{
			// Rare: more runs than the lane buffers hold => reserve an upper bound in the arena and store directly.
			if (!out.spill && out.n_out > SINK_SPILL_AT)
			{
				window tmp;
				uint32_t const col_end = wave_select0(S, D.blocks, src_pos + ref_consumed - 1, tmp, lane);
				uint32_t const bound = out.n_out + (col_end - col + 1) + (n_cigar - ci) + 2;
				uint64_t loc;
				if (!arena_take(arena, bound, W, lane, loc)) { res.status = PVC3_ST_OK; res.n_out = 0; res.loc = 0; return; }
				uint32_t *dst = W.arena + loc;
				if (lane < out.n_out) dst[lane] = out.buf0;
				if (64 + lane < out.n_out) dst[64 + lane] = out.buf1;
				if (128 + lane < out.n_out) dst[128 + lane] = out.buf2;
				out.spill = dst;
				spill_loc = loc;
			}

			col_pos const p = split_col(col);
			if (p.blk - win.b0 >= 2)
				window_load(win, S.blocks, D.blocks, p.blk, lane);
			uint32_t const wi = (p.blk - win.b0) * BLOCK_WORDS + 1 + p.w;
			uint64_t const sw = window_word(win, wi);
			uint64_t const dw = window_word(win, 16 + wi);
			uint64_t const ge = ~UINT64_C(0) << p.bit;
			uint64_t const z = ~sw & ge;				// source non-gap columns at or after col
			uint32_t const nz = popc64(z);
			uint64_t act;								// columns handled in this step
			uint32_t consumed;
			uint32_t const word_col = col - p.bit;
			if (nz >= n_rem)
			{
				uint32_t const e = wave_select_in_word(z, n_rem - 1, lane);	// last base of this operation
				act = ge & low_mask(e + 1);
				consumed = n_rem;
				col = word_col + e + 1;
			}
			else
			{
				act = ge;
				consumed = nz;
				col = word_col + 64;
			}

			uint64_t const dz = ~dw & act;				// destination non-gap columns
			uint64_t const Z = z & act;					// consumed source bases
			uint64_t const G = sw & act;				// source gaps in between
			uint32_t const nd = popc64(dz);
			uint64_t mE = 0, mX = 0, mI, mD;
			if (m_type)
			{
				mI = Z & ~dz;							// dst gap: match becomes insertion (:93-98)
				mD = G & dz;							// dst characters between source bases: deletion (:90-91)
				uint64_t const mC = Z & dz;				// compare query with the dst text (:99-113)
				if (mC)
				{
					// Highest text index needed by this step.
					uint32_t const hi = 63 - (uint32_t) __builtin_clzll(mC);
					uint32_t const max_ri = r + popc64(dz & low_mask(hi));
					if (!D.text || max_ri >= D.text_len) { res.status = PVC3_ST_REF_RANGE; res.n_out = 0; return; }
					if (r + nd > ref_base + REF_WINDOW_BYTES)
					{
						ref_base = r & ~3u;
						refw = *(uint32_t const *) (D.text + ref_base + 4 * lane);
					}
					bool const in_c = (mC >> lane) & 1;
					uint32_t const qi = q + mbcnt64(Z);
					uint32_t const ri = r + mbcnt64(dz) - ref_base;
					uint32_t code;
					if (q + 64 <= SEQ_WINDOW_BASES)
					{
						uint32_t const wv = (uint32_t) __shfl((int) seqw, (int) (qi >> 3));
						uint32_t const byte = (wv >> (((qi >> 1) & 3) * 8)) & 0xff;
						code = (qi & 1) ? (byte & 0xf) : (byte >> 4);
					}
					else
					{
						uint32_t byte = 0;
						if (in_c) byte = seq4[qi >> 1];
						code = (qi & 1) ? (byte & 0xf) : (byte >> 4);
					}
					uint32_t const rw = (uint32_t) __shfl((int) refw, (int) ((ri >> 2) & 63));
					uint32_t const rb = (rw >> ((ri & 3) * 8)) & 0xff;
					mE = ballot(in_c && seq4_to_dna5_char(code) == rb);		// raw byte compare (quirk Q3)
					mX = mC & ~mE;
				}
				q += consumed;
			}
			else
			{
				mI = 0;
				mD = dz;								// (:133-137) kept where dst has a character, dropped otherwise
			}
			emit_tokens(out, mE, mX, mI, mD, lane);
			r += nd;
			n_rem -= consumed;
		}
	}
	out.finish(lane);

	res.n_out = out.n_out;
	res.has_run = out.has_run;
	res.cols = col - col_start;
	res.ref_span = r - res.dst_pos;
	if (out.spill)
		res.loc = spill_loc;
	else if (out.n_out)
	{
		uint64_t loc;
		if (!arena_take(arena, out.n_out, W, lane, loc)) { res.n_out = 0; res.has_run = 0; return; }
		uint32_t *dst = W.arena + loc;
		if (lane < out.n_out) dst[lane] = out.buf0;
		if (out.n_out > 64)
		{
			if (64 + lane < out.n_out) dst[64 + lane] = out.buf1;
			if (128 + lane < out.n_out) dst[128 + lane] = out.buf2;
		}
		res.loc = loc;
	}
}

__global__ __launch_bounds__(256) void k_project(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	// Contiguous record range per wave: sequential descriptor reads and a sequentially filled arena chunk.
	uint64_t const begin = in.n * wave / n_waves;
	uint64_t const end = in.n * (wave + 1) / n_waves;
	arena_alloc arena{0, 0};
	uint32_t n_failed = 0, n_with_runs = 0;
	uint64_t bitvec_bytes = 0, ref_bytes = 0;

	for (uint64_t rec = begin; rec < end; ++rec)
	{
		record_result res;
		project_record(entries, n_entries, in, W, rec, arena, lane, res);
		if (0 == lane)
		{
			W.dst_pos[rec] = res.status ? 0 : res.dst_pos;		// a rejected record has empty outputs
			W.cig_loc[rec] = res.loc;
			W.cig_n[rec] = res.status ? 0 : res.n_out;
			W.run_n[rec] = 0;
			W.rec_flags[rec] = res.status | ((res.has_run && !res.status) ? REC_HAS_RUN : 0);
		}
		n_failed += (0 != res.status);
		n_with_runs += (res.has_run && !res.status);
		bitvec_bytes += 2 * ((res.cols + 8) / 8);		// src + dst windows, at least the first column's byte
		ref_bytes += res.ref_span;
	}
	if (0 == lane)
	{
		atomicAdd(&W.counters->bitvec_bytes, (unsigned long long) bitvec_bytes);
		atomicAdd(&W.counters->ref_bytes, (unsigned long long) ref_bytes);
		if (n_failed) atomicAdd(&W.counters->n_failed, (unsigned long long) n_failed);
		if (n_with_runs) atomicAdd(&W.counters->n_realigned_records, (unsigned long long) n_with_runs);
	}
}

__global__ __launch_bounds__(256) void k_mates(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t rec = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; rec < in.n; rec += stride)
	{
		uint32_t res = PVC3_NO_MATE;
		if (in.mate_pos && (in.flags[rec] & PVC3_FLAG_HAS_MATE))
		{
			uint32_t const s = in.mate_src_entry[rec], d = in.mate_dst_entry[rec];
			if (s < n_entries && d < n_entries)
				res = thread_project_position(entries[s], entries[d], in.mate_pos[rec]);
		}
		W.mate_dst_pos[rec] = res;
	}
}

__global__ __launch_bounds__(256) void k_project_positions(
	dev_entry const *__restrict__ entries, uint32_t n_entries, uint64_t n,
	uint32_t const *__restrict__ src_entry, uint32_t const *__restrict__ dst_entry, uint32_t const *__restrict__ pos, uint32_t *__restrict__ out
)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
	{
		uint32_t const s = src_entry[i], d = dst_entry[i];
		out[i] = (s < n_entries && d < n_entries) ? thread_project_position(entries[s], entries[d], pos[i]) : PVC3_NO_MATE;
	}
}


// ---- packing: exclusive scans of the per-record counts, then a gather into record order ------------------------------
constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_ITEMS = 16;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

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

// tile_sums[2 t + k] = sum of counts k (0 = cigar, 1 = runs) over tile t
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(uint64_t n, uint32_t const *__restrict__ a, uint32_t const *__restrict__ b, uint64_t *__restrict__ tile_sums)
{
	__shared__ uint64_t lds[8];
	uint64_t const base = (uint64_t) blockIdx.x * SCAN_TILE;
	uint64_t sa = 0, sb = 0;
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
	{
		uint64_t const idx = base + (uint64_t) i * SCAN_THREADS + threadIdx.x;
		if (idx < n) { sa += a[idx]; sb += b[idx]; }
	}
	sa = block_reduce_add(sa, lds);
	sb = block_reduce_add(sb, lds);
	if (0 == threadIdx.x) { tile_sums[2 * (uint64_t) blockIdx.x] = sa; tile_sums[2 * (uint64_t) blockIdx.x + 1] = sb; }
}

// In-place exclusive scan of the tile sums by one block (tile counts are small: n / 4096).
__global__ __launch_bounds__(1024) void k_scan_tile_sums(uint64_t n_tiles, uint64_t *__restrict__ tile_sums, uint64_t *__restrict__ totals)
{
	__shared__ uint64_t lds_a[1024], lds_b[1024];
	__shared__ uint64_t carry[2];
	if (0 == threadIdx.x) { carry[0] = 0; carry[1] = 0; }
	__syncthreads();
	for (uint64_t base = 0; base < n_tiles; base += 1024)
	{
		uint64_t const idx = base + threadIdx.x;
		uint64_t va = (idx < n_tiles) ? tile_sums[2 * idx] : 0, vb = (idx < n_tiles) ? tile_sums[2 * idx + 1] : 0;
		lds_a[threadIdx.x] = va; lds_b[threadIdx.x] = vb;
		__syncthreads();
		for (uint32_t off = 1; off < 1024; off <<= 1)	// Hillis-Steele inclusive scan
		{
			uint64_t ta = 0, tb = 0;
			if (threadIdx.x >= off) { ta = lds_a[threadIdx.x - off]; tb = lds_b[threadIdx.x - off]; }
			__syncthreads();
			lds_a[threadIdx.x] += ta; lds_b[threadIdx.x] += tb;
			__syncthreads();
		}
		if (idx < n_tiles)
		{
			tile_sums[2 * idx] = carry[0] + lds_a[threadIdx.x] - va;
			tile_sums[2 * idx + 1] = carry[1] + lds_b[threadIdx.x] - vb;
		}
		__syncthreads();
		if (1023 == threadIdx.x) { carry[0] += lds_a[1023]; carry[1] += lds_b[1023]; }
		__syncthreads();
	}
	if (0 == threadIdx.x) { totals[0] = carry[0]; totals[1] = carry[1]; }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(
	uint64_t n, uint32_t const *__restrict__ a, uint32_t const *__restrict__ b, uint64_t const *__restrict__ tile_sums,
	uint64_t const *__restrict__ totals, uint64_t *__restrict__ off_a, uint64_t *__restrict__ off_b
)
{
	// Each thread owns SCAN_ITEMS consecutive records of the tile.
	__shared__ uint64_t lds_a[SCAN_THREADS], lds_b[SCAN_THREADS];
	uint64_t const base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
	uint32_t ca[SCAN_ITEMS], cb[SCAN_ITEMS];
	uint64_t sa = 0, sb = 0;
#pragma unroll
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
	{
		uint64_t const idx = base + i;
		ca[i] = (idx < n) ? a[idx] : 0; cb[i] = (idx < n) ? b[idx] : 0;
		sa += ca[i]; sb += cb[i];
	}
	lds_a[threadIdx.x] = sa; lds_b[threadIdx.x] = sb;
	__syncthreads();
	for (uint32_t off = 1; off < SCAN_THREADS; off <<= 1)
	{
		uint64_t ta = 0, tb = 0;
		if (threadIdx.x >= off) { ta = lds_a[threadIdx.x - off]; tb = lds_b[threadIdx.x - off]; }
		__syncthreads();
		lds_a[threadIdx.x] += ta; lds_b[threadIdx.x] += tb;
		__syncthreads();
	}
	uint64_t xa = tile_sums[2 * (uint64_t) blockIdx.x] + lds_a[threadIdx.x] - sa;
	uint64_t xb = tile_sums[2 * (uint64_t) blockIdx.x + 1] + lds_b[threadIdx.x] - sb;
#pragma unroll
	for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
	{
		uint64_t const idx = base + i;
		if (idx < n) { off_a[idx] = xa; off_b[idx] = xb; }
		xa += ca[i]; xb += cb[i];
	}
	if (0 == blockIdx.x && 0 == threadIdx.x) { off_a[n] = totals[0]; off_b[n] = totals[1]; }
}

__global__ __launch_bounds__(256) void k_gather(uint64_t n, dev_batch_work const W, dev_batch_packed const P)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t rec = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; rec < n; rec += stride)
	{
		uint32_t const nc = W.cig_n[rec];
		uint32_t const *src = W.arena + W.cig_loc[rec];
		uint32_t *dst = P.out_cigar + P.out_cigar_off[rec];
		for (uint32_t i = 0; i < nc; ++i) dst[i] = src[i];
		uint32_t const nr = W.run_n[rec];
		if (nr)
		{
			pvc3_run const *rs = W.runs_arena + W.run_loc[rec];
			pvc3_run *rd = P.runs + P.run_off[rec];
			for (uint32_t i = 0; i < nr; ++i) rd[i] = rs[i];
		}
		P.status[rec] = (int32_t) (W.rec_flags[rec] & REC_STATUS_MASK);
	}
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

uint32_t grid_for(uint64_t n, uint32_t per_block, uint32_t max_blocks)
{
	uint64_t const b = (n + per_block - 1) / per_block;
	return (uint32_t) (b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

}	// anonymous namespace


// Persistent waves: 4 waves per block, at most 8 blocks per CU; at least 8 records per wave.
uint32_t project_wave_count(uint64_t n) { return 4 * grid_for(n, 4 * 8, 256 * 8); }

int launch_project(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream)
{
	if (0 == in.n) return 0;
	uint32_t const grid = project_wave_count(in.n) / 4;
	hipLaunchKernelGGL(k_project, dim3(grid), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work);
	return (int) hipGetLastError();
}

int launch_mates(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream)
{
	if (0 == in.n) return 0;
	hipLaunchKernelGGL(k_mates, dim3(grid_for(in.n, 256, 256 * 16)), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work);
	return (int) hipGetLastError();
}

int launch_project_positions(dev_entry const *entries, uint32_t n_entries, uint64_t n, uint32_t const *src_entry, uint32_t const *dst_entry, uint32_t const *pos, uint32_t *out, void *stream)
{
	if (0 == n) return 0;
	hipLaunchKernelGGL(k_project_positions, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t) stream, entries, n_entries, n, src_entry, dst_entry, pos, out);
	return (int) hipGetLastError();
}

uint64_t pack_scan_tmp_words(uint64_t n)
{
	return 2 * ((n + SCAN_TILE - 1) / SCAN_TILE) + 2 + 2;
}

int launch_pack(uint64_t n, dev_batch_work const &work, dev_batch_packed const &packed, uint64_t *scan_tmp, void *stream)
{
	uint64_t const n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	uint64_t *totals = scan_tmp + 2 * n_tiles + 2;
	hipStream_t const s = (hipStream_t) stream;
	if (n_tiles)
		hipLaunchKernelGGL(k_scan_tiles, dim3((uint32_t) n_tiles), dim3(SCAN_THREADS), 0, s, n, work.cig_n, work.run_n, scan_tmp);
	hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, s, n_tiles, scan_tmp, totals);
	hipLaunchKernelGGL(k_scan_apply, dim3((uint32_t) (n_tiles ? n_tiles : 1)), dim3(SCAN_THREADS), 0, s, n, work.cig_n, work.run_n, scan_tmp, totals, packed.out_cigar_off, packed.run_off);
	if (n)
		hipLaunchKernelGGL(k_gather, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, n, work, packed);
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
