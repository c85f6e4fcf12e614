// records.hip — the record path of project_alignments on the GPU (include/pvc3_records.h): BAM records found in the inflated
// stream, filtered, scattered into the batch arrays of pvc3.h; and, after the projection, written anew as BAM bytes.
//
// Reference behaviour followed (project-alignments/project_alignments.cc): record filters :528-570, four records per task
// :600-627, tag filter :775-802, OA :805-844, realigned ranges :848-861, record index :864-869, original RNAME / POS :872-877,
// mate :880-911, POS / CIGAR / RNAME :915-918.  The record layout is SAMv1 §4.2.
//
// Everything here is byte and integer work on a few hundred bytes per record: the kernels are lane-per-record (or
// lane-per-BGZF-block for the framing) and bound by the latency of their dependent loads, not by HBM; a piece of 100 MB
// takes them well under a millisecond next to the 5 - 10 ms the two BGZF codecs need for the same piece.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "records_internal.hpp"

namespace pvc3 {
namespace {

constexpr uint32_t WAVE = 64;
constexpr uint32_t NONE = 0xFFFFFFFFu;			// a block without a record start
constexpr uint32_t END_MARK = 0xFFFFFFFEu;		// the chain has reached the incomplete last record
constexpr uint32_t MAX_BLOCK_SIZE = 1u << 28;	// a record's block_size beyond this is taken for garbage

__device__ inline uint32_t ld32(uint8_t const *p)
{
	// two aligned dwords shifted into place (the buffers have slack behind their last byte); both are requested at once, whatever
	// the alignment: a second load that waits for a branch on the first doubles the latency of a hop along a chain of records
	uintptr_t const a = (uintptr_t) p;
	uint32_t const *q = (uint32_t const *) (a & ~(uintptr_t) 3);
	uint32_t const lo = q[0], hi = q[1];
	return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t) (a & 3));
}

__device__ inline uint32_t ld16(uint8_t const *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8); }

// What of a record's fixed fields must hold for any record of a file with n_refs references (p: its block_size field).
__device__ inline bool plausible_record(uint8_t const *U, uint64_t p, uint64_t end, uint32_t n_refs)
{
	if (p + 36 > end) return false;
	uint8_t const *r = U + p;
	uint32_t const bs = ld32(r);
	if (bs < 32 || bs > MAX_BLOCK_SIZE) return false;
	int32_t const ref_id = (int32_t) ld32(r + 4), pos = (int32_t) ld32(r + 8), next_ref = (int32_t) ld32(r + 24), next_pos = (int32_t) ld32(r + 28);
	uint32_t const l_name = r[12], n_cigar = ld16(r + 16), l_seq = ld32(r + 20);
	if (ref_id < -1 || ref_id >= (int32_t) n_refs || next_ref < -1 || next_ref >= (int32_t) n_refs || pos < -1 || next_pos < -1) return false;
	if (0 == l_name || l_seq > MAX_BLOCK_SIZE) return false;
	uint64_t const need = 32 + (uint64_t) l_name + 4 * (uint64_t) n_cigar + ((uint64_t) l_seq + 1) / 2 + l_seq;
	if (need > bs) return false;
	uint64_t const nul = p + 36 + l_name - 1;
	if (nul < end && 0 != U[nul]) return false;
	return true;
}

struct walk_result { uint32_t cnt, exit, tail; bool invalid; };

// The chain of records from `pos` on while they begin before blk_end.  exit: where the first record that begins at or behind
// blk_end begins — or END_MARK when the chain meets a record that is not whole (tail: where that one begins).
template <typename t_emit>
__device__ inline walk_result walk_records(uint8_t const *U, uint64_t pos, uint64_t blk_end, uint64_t end, t_emit &&emit)
{
	walk_result w{0, 0, 0, false};
	while (pos < blk_end)
	{
		if (pos + 4 > end) { w.exit = END_MARK; w.tail = (uint32_t) pos; return w; }
		uint32_t const bs = ld32(U + pos);
		if (bs < 32 || bs > MAX_BLOCK_SIZE) { w.invalid = true; w.exit = END_MARK; w.tail = (uint32_t) pos; return w; }
		if (pos + 4 + bs > end) { w.exit = END_MARK; w.tail = (uint32_t) pos; return w; }
		emit(w.cnt, (uint32_t) pos);
		++w.cnt;
		pos += 4 + (uint64_t) bs;
	}
	w.exit = (uint32_t) pos;
	return w;
}

// ---- framing ---------------------------------------------------------------------------------------------------------------------
// Lane = BGZF block.  Writers that never let a record straddle two blocks (htslib: bam_write1 flushes first) make every block
// begin with a record; in general the first record start inside a block is guessed — the first place where a plausible record
// begins whose successor is plausible too — and the chain walked from there.  A walk that starts on a record start stays on
// record starts, so the guesses only have to be CHECKED: k_frame_fix follows the chain through the blocks (the first record
// start in a block is where the previous block's walk ended) and walks again, from the true place, every block whose guess
// was not that.  The result is exact whatever the guesses were; they only decide how parallel it was.
__global__ __launch_bounds__(WAVE) void k_frame_walk(uint8_t const *__restrict__ U, uint32_t const *__restrict__ blk_off, uint32_t n_blocks, uint32_t start0, uint32_t end, uint32_t n_refs,
	uint32_t *__restrict__ guess, uint32_t *__restrict__ cnt, uint32_t *__restrict__ exits, uint32_t *__restrict__ tails)
{
	uint32_t const b = blockIdx.x * WAVE + threadIdx.x;
	if (b >= n_blocks) return;
	uint64_t const lo = (0 == b) ? start0 : blk_off[b], hi = blk_off[b + 1];
	uint64_t g = NONE;
	if (0 == b) g = lo;
	else
		for (uint64_t p = lo; p < hi; ++p)
		{
			if (!plausible_record(U, p, end, n_refs)) continue;
			uint64_t const q = p + 4 + (uint64_t) ld32(U + p);
			if (q + 36 <= end && !plausible_record(U, q, end, n_refs)) continue;
			g = p;
			break;
		}
	walk_result w{0, NONE, 0, false};
	if (NONE != g)
	{
		w = walk_records(U, g, hi, end, [](uint32_t, uint32_t) {});
		if (w.invalid) { g = NONE; w = walk_result{0, NONE, 0, false}; }		// a guess that led into garbage: no guess (k_frame_fix walks from the true place, and reports what it meets there)
	}
	guess[b] = (uint32_t) g; cnt[b] = w.cnt; exits[b] = w.exit; tails[b] = w.tail;
}

// One wave follows the chain through the blocks, 64 at a time.
__global__ __launch_bounds__(WAVE) void k_frame_fix(uint8_t const *__restrict__ U, uint32_t const *__restrict__ blk_off, uint32_t n_blocks, uint32_t start0, uint32_t end,
	uint32_t *__restrict__ guess, uint32_t *__restrict__ cnt, uint32_t *__restrict__ exits, uint32_t *__restrict__ tails, uint32_t *__restrict__ rec_base, rec_state *__restrict__ st)
{
	uint32_t const lane = threadIdx.x;
	uint32_t carry_in = start0;		// where the first record that begins at or behind the group's first block begins
	uint32_t tail = end;				// where the incomplete last record begins, once the chain has met it
	uint32_t total = 0;
	uint32_t invalid_at = NONE;
	for (uint32_t g0 = 0; g0 < n_blocks; g0 += WAVE)
	{
		uint32_t const b = g0 + lane;
		bool const have = b < n_blocks;
		uint32_t const blk_end = have ? blk_off[b + 1] : 0;
		for (;;)
		{
			uint32_t const ge = have ? guess[b] : NONE, ex = have ? exits[b] : NONE;
			bool const defined = have && NONE != ge;
			unsigned long long const mask = __ballot(defined);
			unsigned long long const before = mask & ((1ull << lane) - 1ull);
			int const src = before ? (63 - __clzll((long long) before)) : -1;
			uint32_t const from_prev = (uint32_t) __shfl((int) ex, src < 0 ? 0 : src);
			uint32_t const in = (src < 0) ? carry_in : from_prev;
			bool const ok = !have || (defined ? (in == ge) : (in >= blk_end));
			unsigned long long const bad = __ballot(!ok);
			if (0 == bad)
			{
				// the group is consistent: the chain's value behind it
				if (mask)
				{
					int const last = 63 - __clzll((long long) mask);
					uint32_t const out = (uint32_t) __shfl((int) ex, last);
					uint32_t const t = (uint32_t) __shfl((int) (have ? tails[b] : 0u), last);
					if (END_MARK == out && END_MARK != carry_in) tail = t;
					carry_in = out;
				}
				break;
			}
			uint32_t const f = (uint32_t) __ffsll((long long) bad) - 1;
			if (lane == f)
			{
				// this block again, from where the chain really enters it
				if (in >= blk_end) { guess[b] = NONE; cnt[b] = 0; exits[b] = NONE; tails[b] = 0; }
				else
				{
					walk_result const w = walk_records(U, in, blk_end, end, [](uint32_t, uint32_t) {});
					guess[b] = in; cnt[b] = w.cnt; exits[b] = w.exit; tails[b] = w.tail;
					if (w.invalid) invalid_at = w.tail;
				}
			}
			__threadfence_block();
		}
		// exclusive prefix sums of the record counts
		uint32_t const c = have ? cnt[b] : 0;
		uint32_t s = c;
		for (uint32_t d = 1; d < WAVE; d <<= 1)
		{
			uint32_t const v = (uint32_t) __shfl_up((int) s, d);
			if (lane >= d) s += v;
		}
		if (have) rec_base[b] = total + s - c;
		total += (uint32_t) __shfl((int) s, 63);
	}
	unsigned long long const inv = __ballot(NONE != invalid_at);
	if (inv)
	{
		uint32_t const at = (uint32_t) __shfl((int) invalid_at, (int) __ffsll((long long) inv) - 1);
		if (0 == lane) st->frame_error_at = at;
	}
	if (0 == lane)
	{
		st->n_records = total;
		st->tail_off = (END_MARK == carry_in) ? tail : ((carry_in > end) ? end : carry_in);
		if (END_MARK != carry_in && carry_in > end) st->frame_error_at = carry_in;		// (cannot happen: a record that passes `end` ends the chain)
	}
}

__global__ __launch_bounds__(WAVE) void k_frame_emit(uint8_t const *__restrict__ U, uint32_t const *__restrict__ blk_off, uint32_t n_blocks, uint32_t end,
	uint32_t const *__restrict__ guess, uint32_t const *__restrict__ rec_base, uint32_t *__restrict__ rec_off)
{
	uint32_t const b = blockIdx.x * WAVE + threadIdx.x;
	if (b >= n_blocks) return;
	uint32_t const g = guess[b];
	if (NONE == g) return;
	uint32_t *dst = rec_off + rec_base[b];
	(void) walk_records(U, g, blk_off[b + 1], end, [dst](uint32_t i, uint32_t pos) { dst[i] = pos; });
}

// ---- filters (project_alignments.cc:528-570) ----------------------------------------------------------------------------------------
__device__ inline void note_error(rec_state *st, uint64_t input_index, uint32_t code)
{
	atomicMin(&st->err_key, (unsigned long long) ((input_index << 8) | code));
}

__device__ inline uint64_t input_index_of(rec_state const *st, uint32_t i, uint32_t n_pending, uint32_t parity, uint64_t seen)
{
	return (i < n_pending) ? st->pending_idx[parity][i] : (seen + (i - n_pending));
}

__global__ __launch_bounds__(256) void k_filter(uint8_t const *__restrict__ U, uint32_t const *__restrict__ rec_off, rec_state *__restrict__ st, uint32_t n_refs,
	uint32_t n_pending, uint32_t parity, uint64_t seen, uint32_t *__restrict__ keep)
{
	uint32_t const n = st->n_records;
	uint32_t const i = blockIdx.x * 256 + threadIdx.x;
	uint32_t cat = 0;		// 1 matched, 2 flags not matched, 3 reference id missing, 4 sequence missing
	bool fresh = false;
	if (i < n)
	{
		uint8_t const *r = U + rec_off[i];
		uint32_t const bs = ld32(r);
		int32_t const ref_id = (int32_t) ld32(r + 4), pos = (int32_t) ld32(r + 8), next_ref = (int32_t) ld32(r + 24);
		uint32_t const l_name = r[12], n_cigar = ld16(r + 16), flag = ld16(r + 18), l_seq = ld32(r + 20);
		uint64_t const need = 32 + (uint64_t) l_name + 4 * (uint64_t) n_cigar + ((uint64_t) l_seq + 1) / 2 + l_seq;
		uint64_t const idx = input_index_of(st, i, n_pending, parity, seen);
		if (need > bs) note_error(st, idx, PVC3_REC_E_TRUNCATED);
		if (flag & (0x4 | 0x200 | 0x400)) cat = 2;			// unmapped, QC fail, duplicate (:528-537)
		else if (ref_id < 0) cat = 3;						// :546-551
		else if (pos < 0) cat = 2;							// :553-564
		else if (0 == l_seq) cat = 4;						// :566-570
		else
		{
			cat = 1;
			if (ref_id >= (int32_t) n_refs || next_ref >= (int32_t) n_refs) { note_error(st, idx, PVC3_REC_E_REF_ID); cat = 2; }
		}
		if (need > bs) cat = 2;		// (reported above; nothing downstream reads a record that is reported)
		keep[i] = (1 == cat) ? 1u : 0u;
		fresh = i >= n_pending;		// the records that waited for their chunk were counted when they arrived
	}
	for (uint32_t c = 1; c <= 4; ++c)
	{
		unsigned long long const m = __ballot(fresh && cat == c);
		if (m && 0 == (threadIdx.x & 63))
			atomicAdd(&st->counts[c - 1], (unsigned long long) __popcll(m));
	}
}

// ---- exclusive prefix sums (uint32 in, uint64 out, out[n] = total) ---------------------------------------------------------------------
// (Blocks of 256 threads: next to the two BGZF codecs' waves, which hold most of a CU's registers and LDS for milliseconds, a block
// of sixteen waves waits until a CU is nearly empty — a scan of half a million numbers took a third of a millisecond that way.)
constexpr uint32_t SCAN_BLOCK = 256, SCAN_PER_THREAD = 16, SCAN_TILE = SCAN_BLOCK * SCAN_PER_THREAD;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &block_total)
{
	uint32_t const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
	uint32_t s = v;
	for (uint32_t d = 1; d < WAVE; d <<= 1)
	{
		uint32_t const t = (uint32_t) __shfl_up((int) s, d);
		if (lane >= d) s += t;
	}
	if (63 == lane) lds[wave] = s;
	__syncthreads();
	if (0 == wave)
	{
		uint32_t w = (lane < n_waves) ? lds[lane] : 0, ws = w;
		for (uint32_t d = 1; d < WAVE; d <<= 1)
		{
			uint32_t const t = (uint32_t) __shfl_up((int) ws, d);
			if (lane >= d) ws += t;
		}
		if (lane < n_waves) lds[lane] = ws - w;
		if (lane == n_waves - 1) lds[32] = ws;
	}
	__syncthreads();
	uint32_t const r = lds[wave] + s - v;
	block_total = lds[32];
	__syncthreads();
	return r;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tiles(uint32_t const *__restrict__ in, uint32_t const *__restrict__ n_ptr, uint32_t n_fixed, uint64_t *__restrict__ out, uint32_t *__restrict__ tile_sums)
{
	__shared__ uint32_t lds[40];
	uint32_t const n = n_ptr ? *n_ptr : n_fixed;
	uint32_t const base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
	uint32_t v[SCAN_PER_THREAD], sum = 0;
	for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; sum += v[k]; }
	uint32_t total;
	uint32_t run = block_exclusive_scan(sum, lds, total);
	for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
	if (0 == threadIdx.x) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(uint32_t *__restrict__ tile_sums, uint32_t n_tiles, uint64_t *__restrict__ tile_base)
{
	__shared__ uint32_t lds[40];
	uint64_t carry = 0;
	for (uint32_t t0 = 0; t0 < n_tiles; t0 += SCAN_BLOCK)
	{
		uint32_t const t = t0 + threadIdx.x;
		uint32_t const v = (t < n_tiles) ? tile_sums[t] : 0;
		uint32_t total;
		uint32_t const ex = block_exclusive_scan(v, lds, total);
		if (t < n_tiles) tile_base[t] = carry + ex;
		carry += total;
	}
	if (0 == threadIdx.x) tile_base[n_tiles] = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(uint32_t const *__restrict__ n_ptr, uint32_t n_fixed, uint64_t *__restrict__ out, uint64_t const *__restrict__ tile_base, uint32_t n_tiles)
{
	uint32_t const n = n_ptr ? *n_ptr : n_fixed;
	uint64_t const add = tile_base[blockIdx.x];
	uint32_t const base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
	for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
		if (base + k < n) out[base + k] += add;
	// the total behind the last element (tiles beyond n add nothing)
	if (0 == blockIdx.x && 0 == threadIdx.x) out[n] = tile_base[n_tiles];
}

// ---- the batch: kept records, a multiple of four of them (project_alignments.cc:600-627) -----------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(uint32_t const *__restrict__ rec_off, uint32_t const *__restrict__ keep, uint64_t const *__restrict__ kept_pos, rec_state *__restrict__ st,
	uint32_t n_pending, uint32_t parity, uint64_t seen, int last, uint32_t *__restrict__ kept_off, uint64_t *__restrict__ kept_input)
{
	uint32_t const n = st->n_records;
	uint32_t const n_kept = (uint32_t) kept_pos[n];
	uint32_t const n_take = last ? n_kept : (n_kept & ~3u);
	uint32_t const i = blockIdx.x * 256 + threadIdx.x;
	if (0 == i) { st->n_kept = n_kept; st->n_take = n_take; st->n_deferred = n_kept - n_take; }
	if (i >= n || !keep[i]) return;
	uint32_t const j = (uint32_t) kept_pos[i];
	uint64_t const idx = input_index_of(st, i, n_pending, parity, seen);
	if (j < n_take) { kept_off[j] = rec_off[i]; kept_input[j] = idx; }
	else { st->deferred_off[j - n_take] = rec_off[i]; st->pending_idx[parity ^ 1][j - n_take] = idx; }
}

// What the next piece begins with: the (up to three) kept records that wait for their chunk of four, then the incomplete last record.
__global__ __launch_bounds__(256) void k_carry(uint8_t const *__restrict__ U, uint32_t end, rec_state *__restrict__ st, uint8_t *__restrict__ carry, uint32_t carry_cap)
{
	__shared__ uint32_t src[4], len[4], dst[5];
	if (0 == threadIdx.x)
	{
		uint32_t at = 0;
		uint32_t const nd = st->n_deferred;
		for (uint32_t k = 0; k < 4; ++k)
		{
			if (k < nd) { src[k] = st->deferred_off[k]; len[k] = 4 + ld32(U + src[k]); }
			else if (k == nd) { src[k] = st->tail_off; len[k] = end - st->tail_off; }
			else { src[k] = 0; len[k] = 0; }
			dst[k] = at;
			at = (len[k] > carry_cap - at) ? (carry_cap + 1) : (at + len[k]);
			if (at > carry_cap) break;
		}
		dst[4] = at;
		st->carry_len = at;
	}
	__syncthreads();
	if (dst[4] > carry_cap) return;		// (the host reports PVC3_REC_E_CARRY)
	for (uint32_t k = 0; k < 4; ++k)
		for (uint32_t x = threadIdx.x; x < len[k]; x += 256)
			carry[dst[k] + x] = U[src[k] + x];
}

// The batch arrays of pvc3.h from the kept records, in place: SEQ and QUAL stay where they are (seq_off / qual_off are offsets
// from the stream buffer's base), the CIGAR operations are copied because they must be aligned.
__global__ __launch_bounds__(256) void k_scatter(uint8_t const *__restrict__ U, uint32_t end, uint32_t const *__restrict__ kept_off, rec_state const *__restrict__ st,
	uint32_t const *__restrict__ t_src, uint32_t const *__restrict__ t_dst,
	uint32_t *__restrict__ src_pos, uint32_t *__restrict__ src_entry, uint32_t *__restrict__ dst_entry, uint32_t *__restrict__ flags, uint32_t *__restrict__ l_seq_out,
	uint64_t *__restrict__ seq_off, uint64_t *__restrict__ qual_off, uint32_t *__restrict__ mate_se, uint32_t *__restrict__ mate_de, uint32_t *__restrict__ mate_pos, uint32_t *__restrict__ n_cigar_out)
{
	uint32_t const n = st->n_take;
	uint32_t const j = blockIdx.x * 256 + threadIdx.x;
	if (0 == j) { seq_off[n] = end; qual_off[n] = end; }
	if (j >= n) return;
	uint32_t const at = kept_off[j];
	uint8_t const *r = U + at;
	int32_t const ref_id = (int32_t) ld32(r + 4), next_ref = (int32_t) ld32(r + 24), next_pos = (int32_t) ld32(r + 28);
	uint32_t const l_name = r[12], n_cigar = ld16(r + 16), l_seq = ld32(r + 20);
	uint32_t const seq_at = at + 36 + l_name + 4 * n_cigar, qual_at = seq_at + (l_seq + 1) / 2;
	src_pos[j] = ld32(r + 8);
	src_entry[j] = t_src[ref_id];
	dst_entry[j] = t_dst[ref_id];
	l_seq_out[j] = l_seq;
	uint32_t fl = (l_seq && 0xff != U[qual_at]) ? PVC3_FLAG_HAS_QUAL : 0;
	uint32_t mse = 0, mde = 0, mp = 0;
	if (next_ref >= 0 && next_pos >= 0) { fl |= PVC3_FLAG_HAS_MATE; mse = t_src[next_ref]; mde = t_dst[next_ref]; mp = (uint32_t) next_pos; }
	flags[j] = fl; mate_se[j] = mse; mate_de[j] = mde; mate_pos[j] = mp;
	seq_off[j] = seq_at; qual_off[j] = qual_at;
	n_cigar_out[j] = n_cigar;
}

__global__ __launch_bounds__(256) void k_scatter_cigar(uint8_t const *__restrict__ U, uint32_t const *__restrict__ kept_off, rec_state const *__restrict__ st, uint64_t const *__restrict__ cigar_off, uint64_t cigar_cap, uint32_t *__restrict__ cigar)
{
	uint32_t const n = st->n_take;
	uint32_t const j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n || cigar_off[n] > cigar_cap) return;		// (more operations than the batch object holds: the host makes a larger one and comes back)
	uint8_t const *r = U + kept_off[j];
	uint32_t const l_name = r[12], n_cigar = ld16(r + 16);
	uint8_t const *c = r + 36 + l_name;
	uint32_t *dst = cigar + cigar_off[j];
	for (uint32_t k = 0; k < n_cigar; ++k) dst[k] = ld32(c + 4 * k);
	if (j + 1 == n)
		for (uint32_t k = 0; k < 64; ++k) dst[n_cigar + k] = 0;		// the kernels read whole 16-byte pieces behind the last operation
}

// ---- rewriting -----------------------------------------------------------------------------------------------------------------------
struct count_sink
{
	uint32_t n{};
	__device__ void byte(uint32_t) { ++n; }
	__device__ void copy(uint8_t const *, uint32_t k) { n += k; }
	__device__ void copy_seq(uint8_t const *, uint32_t l_seq) { n += (l_seq + 1) / 2; }
	__device__ void copy_qual(uint8_t const *, uint32_t l_seq, bool) { n += l_seq; }
};

struct write_sink
{
	uint8_t *p;
	uint32_t left, block_bytes, gap;		// gap: bytes between the end of a block's data and the beginning of the next block's
	uint8_t *base;

	__device__ write_sink(uint8_t *out, uint64_t o, uint32_t bb, uint32_t stride, uint32_t pad): block_bytes(bb), gap(stride - bb), base(out)
	{
		uint64_t const blk = o / bb;
		uint32_t const in = (uint32_t) (o - blk * bb);
		p = out + blk * stride + pad + in;
		left = bb - in;
	}
	__device__ void byte(uint32_t v)
	{
		*p++ = (uint8_t) v;
		if (0 == --left) { p += gap; left = block_bytes; }
	}
	// Four bytes at a time where the output allows it: the bytes in front of a dword boundary of the output one by one, then
	// whole dwords (the source read as two aligned dwords shifted into place), then the rest; f transforms four bytes.
	template <typename t_fn>
	__device__ void copy4(uint8_t const *s, uint32_t k, t_fn &&f)
	{
		uint32_t x = 0;
		while (x < k && (((uintptr_t) p) & 3)) { byte(f((uint32_t) s[x]) & 0xff); ++x; }
		while (x + 4 <= k)
		{
			if (left < 4) { byte(f((uint32_t) s[x]) & 0xff); ++x; continue; }		// (a block of the output ends inside these four bytes; the next block's data begin on a dword boundary again or the loop above takes over)
			if (((uintptr_t) p) & 3) { byte(f((uint32_t) s[x]) & 0xff); ++x; continue; }
			*(uint32_t *) p = f(ld32(s + x));
			p += 4; left -= 4; x += 4;
			if (0 == left) { p += gap; left = block_bytes; }
		}
		for (; x < k; ++x) byte(f((uint32_t) s[x]) & 0xff);
	}
	__device__ void copy(uint8_t const *s, uint32_t k) { copy4(s, k, [](uint32_t v) { return v; }); }
	// SEQ through seqan3::dna5: a code that is not A, C, G or T becomes N; the unused low nibble of an odd-length SEQ stays as it was
	__device__ void copy_seq(uint8_t const *s, uint32_t l_seq)
	{
		uint32_t const nb = (l_seq + 1) / 2;
		// every nibble of the (up to four) bytes: one-hot (1, 2, 4, 8) stays, anything else becomes 15
		auto canon = [](uint32_t v) {
			uint32_t out = 0;
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k)
			{
				uint32_t const c = (v >> (4 * k)) & 15;
				out |= ((1 == c || 2 == c || 4 == c || 8 == c) ? c : 15u) << (4 * k);
			}
			return out;
		};
		if (l_seq & 1)
		{
			// all but the last byte, then the last one with its low nibble untouched
			copy4(s, nb - 1, canon);
			uint32_t const v = s[nb - 1], hi = v >> 4;
			byte((((1 == hi || 2 == hi || 4 == hi || 8 == hi) ? hi : 15u) << 4) | (v & 15));
		}
		else
			copy4(s, nb, canon);
	}
	// QUAL through seqan3::phred42: values beyond 41 saturate (a missing QUAL, 0xff ..., is copied as it is)
	__device__ void copy_qual(uint8_t const *s, uint32_t l_seq, bool has_qual)
	{
		if (!has_qual) { copy(s, l_seq); return; }
		copy4(s, l_seq, [](uint32_t v) {
			uint32_t out = 0;
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
			{
				uint32_t const q = (v >> (8 * k)) & 0xff;
				out |= ((q > 41) ? 41u : q) << (8 * k);
			}
			return out;
		});
	}
};

__device__ inline uint32_t dec_digits(uint64_t v)
{
	uint32_t d = 1;
	while (v >= 10) { v /= 10; ++d; }
	return d;
}

__constant__ uint64_t POW10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
	10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};

template <typename t_sink>
__device__ inline void put_dec(t_sink &s, int64_t sv)
{
	uint64_t v = (sv < 0) ? (uint64_t) 0 - (uint64_t) sv : (uint64_t) sv;
	if (sv < 0) s.byte('-');
	for (int k = (int) dec_digits(v) - 1; k >= 0; --k)
	{
		uint64_t const q = v / POW10[k];
		s.byte('0' + (uint32_t) (q % 10));
	}
}

template <typename t_sink>
__device__ inline void put32(t_sink &s, uint32_t v) { s.byte(v & 0xff); s.byte((v >> 8) & 0xff); s.byte((v >> 16) & 0xff); s.byte(v >> 24); }

// An integer tag in the smallest type that holds it (non-negative values only occur here).
template <typename t_sink>
__device__ inline void put_int_tag(t_sink &s, uint32_t key, uint64_t v)
{
	s.byte(key >> 8); s.byte(key & 0xff);
	if (v <= 0xff) { s.byte('C'); s.byte((uint32_t) v); }
	else if (v <= 0xffff) { s.byte('S'); s.byte((uint32_t) v & 0xff); s.byte((uint32_t) v >> 8); }
	else { s.byte('I'); put32(s, (uint32_t) v); }
}

__device__ inline uint32_t reg2bin(int64_t beg, int64_t end)		// SAMv1 §5.3
{
	--end;
	if (beg >> 14 == end >> 14) return (uint32_t) (((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (uint32_t) (((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (uint32_t) (((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (uint32_t) (((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (uint32_t) (((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

__device__ inline uint32_t scalar_size(uint32_t t)
{
	switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; }
}

constexpr uint32_t KMAX = PVC3_RECORDS_MAX_TAGS;

// One wave, lane = record of the batch.  EMIT = false: the sizes of the new records (and the histogram of the removed tags);
// EMIT = true: their bytes.  Both walk the old record's tags the same way: the preserved ones are listed in LDS in key order
// (seqan3's tag dictionary is an ordered map: tags come out in key order, a later duplicate replaces an earlier one), and are
// then merged with the new tags, which win over an old tag of the same name.
template <bool EMIT>
__global__ __launch_bounds__(WAVE) void k_rewrite(uint8_t const *__restrict__ U, uint32_t const *__restrict__ kept_off, uint64_t const *__restrict__ kept_input, rec_state *__restrict__ st,
	rec_tables const T, rec_results const R, uint32_t *__restrict__ out_size, uint64_t const *__restrict__ out_off, uint8_t *__restrict__ out, uint32_t block_bytes, uint32_t block_stride, uint32_t block_pad)
{
	__shared__ uint32_t k_key[KMAX][WAVE], k_off[KMAX][WAVE], k_len[KMAX][WAVE];
	uint32_t const lane = threadIdx.x;
	uint32_t const n = st->n_take;
	uint32_t const j = blockIdx.x * WAVE + lane;
	bool const have = j < n;
	bool bad = false;
	uint32_t err = 0;

	uint8_t const *r = U;		// the old record (behind its block_size field)
	uint32_t bs = 0, aux_at = 0, l_name = 0, n_cigar_old = 0, l_seq = 0, flag = 0, mapq = 0;
	int32_t ref_id = 0, pos = 0, next_ref = -1, next_pos = -1;
	if (have)
	{
		uint8_t const *q = U + kept_off[j];
		bs = ld32(q); r = q + 4;
		ref_id = (int32_t) ld32(r); pos = (int32_t) ld32(r + 4); l_name = r[8]; mapq = r[9]; n_cigar_old = ld16(r + 12); flag = ld16(r + 14); l_seq = ld32(r + 16);
		next_ref = (int32_t) ld32(r + 20); next_pos = (int32_t) ld32(r + 24);
		aux_at = 32 + l_name + 4 * n_cigar_old + (l_seq + 1) / 2 + l_seq;
	}

	// ---- the old tags -----------------------------------------------------------------------------------------------------------
	uint32_t n_kept = 0;
	bool have_nm = false; int64_t nm = 0;
	uint32_t old_oa_at = 0, old_oa_len = 0;
	uint32_t q = aux_at;
	for (;;)
	{
		bool const active = have && !bad && q < bs;
		if (!__any(active)) break;
		bool removed = false;
		uint32_t key = 0;
		if (active)
		{
			uint32_t size = 0;
			uint32_t const v = q + 3;
			uint32_t ty = 0;
			if (v > bs) bad = true;
			else
			{
				key = ((uint32_t) r[q] << 8) | r[q + 1]; ty = r[q + 2];
				if (uint32_t const s = scalar_size(ty)) size = s;
				else if ('Z' == ty || 'H' == ty)
				{
					uint32_t e = v;
					while (e < bs && 0 != r[e]) ++e;
					if (e >= bs) bad = true;
					size = e - v + 1;
				}
				else if ('B' == ty)
				{
					if (v + 5 > bs) bad = true;
					else
					{
						uint32_t const es = scalar_size(r[v]);
						uint64_t const sz = 5 + (uint64_t) es * ld32(r + v + 1);
						if (0 == es || sz > bs) bad = true;
						size = (uint32_t) sz;
					}
				}
				else bad = true;
				if (!bad && size > bs - v) bad = true;
			}
			if (bad) err = PVC3_REC_E_AUX;
			else
			{
				if (('N' << 8 | 'M') == key && ('c' == ty || 'C' == ty || 's' == ty || 'S' == ty || 'i' == ty || 'I' == ty))
				{
					have_nm = true;
					switch (ty)
					{
						case 'c': nm = (int8_t) r[v]; break; case 'C': nm = r[v]; break;
						case 's': nm = (int16_t) ld16(r + v); break; case 'S': nm = ld16(r + v); break;
						case 'i': nm = (int32_t) ld32(r + v); break; default: nm = ld32(r + v); break;
					}
				}
				bool const preserved = 0 != ((T.preserved[key >> 5] >> (key & 31)) & 1u);
				if (!preserved) removed = true;
				else if (('O' << 8 | 'A') == key && 'Z' == ty) { old_oa_at = v; old_oa_len = size - 1; }
				else
				{
					// into the list, in key order; the same key again replaces the earlier tag
					uint32_t at = n_kept;
					while (at > 0 && k_key[at - 1][lane] > key) --at;
					if (at > 0 && k_key[at - 1][lane] == key) { k_off[at - 1][lane] = q; k_len[at - 1][lane] = 3 + size; }
					else if (n_kept >= KMAX) { bad = true; err = PVC3_REC_E_TAGS; }
					else
					{
						for (uint32_t x = n_kept; x > at; --x) { k_key[x][lane] = k_key[x - 1][lane]; k_off[x][lane] = k_off[x - 1][lane]; k_len[x][lane] = k_len[x - 1][lane]; }
						k_key[at][lane] = key; k_off[at][lane] = q; k_len[at][lane] = 3 + size;
						++n_kept;
					}
				}
				q = v + size;
			}
		}
		if (!EMIT)
		{
			// the histogram of removed tags (project_alignments.cc:796-800): one atomic per distinct key and step of the wave
			bool pending = removed;
			for (;;)
			{
				unsigned long long const m = __ballot(pending);
				if (0 == m) break;
				int const first = (int) __ffsll((long long) m) - 1;
				uint32_t const k = (uint32_t) __shfl((int) key, first);
				bool const same = pending && key == k;
				unsigned long long const sm = __ballot(same);
				if ((int) lane == first) atomicAdd(&T.removed_counts[k], (unsigned long long) __popcll(sm));
				pending = pending && !same;
			}
		}
	}

	if (have)
	{
		uint64_t const idx = kept_input[j];
		uint32_t const status = R.rec_flags[j] & 0xffu;
		uint64_t const c0 = R.out_cigar_off[j], c1 = R.out_cigar_off[j + 1];
		uint64_t const r0 = R.run_off[j], r1 = R.run_off[j + 1];
		uint32_t const n_cig = (uint32_t) (c1 - c0), n_runs = (uint32_t) (r1 - r0);
		uint32_t const dst_pos = R.dst_pos[j], mate_dst = R.mate_dst_pos[j];
		if (!bad && status) { bad = true; err = PVC3_REC_E_STATUS; }
		if (!bad && c1 - c0 > 65535) { bad = true; err = PVC3_REC_E_CIGAR_COUNT; }
		if (!bad && next_ref >= 0 && next_pos >= 0 && PVC3_NO_MATE == mate_dst) { bad = true; err = PVC3_REC_E_MATE; }
		if (bad) { note_error(st, idx, err); if (!EMIT) out_size[j] = 0; }
		else
		{
			// :864-869: the same value for the (up to) four records of a chunk
			uint32_t const chunk_begin = j & ~3u, chunk_end = (chunk_begin + 4 < n) ? (chunk_begin + 4) : n;
			uint64_t const xi = kept_input[chunk_end - 1] - (chunk_end - chunk_begin) + 1;
			uint32_t const *cig = R.out_cigar + c0;
			pvc3_run const *runs = R.runs + r0;
			bool const has_qual = l_seq && 0xff != r[32 + l_name + 4 * n_cigar_old + (l_seq + 1) / 2];

			auto body = [&](auto &s) {
				// -- fixed fields (:915-918, mate :880-911)
				int32_t const new_next_ref = (next_ref >= 0) ? T.out_ref_id[next_ref] : -1;
				int32_t const new_next_pos = (next_ref >= 0 && next_pos >= 0) ? (int32_t) mate_dst : -1;
				int64_t span = 0;
				for (uint32_t k = 0; k < n_cig; ++k) { uint32_t const op = cig[k] & 0xf; if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) span += cig[k] >> 4; }
				uint32_t const bin = reg2bin(dst_pos, (int64_t) dst_pos + (span ? span : 1));
				put32(s, (uint32_t) T.out_ref_id[ref_id]);
				put32(s, dst_pos);
				s.byte(l_name); s.byte(mapq); s.byte(bin & 0xff); s.byte(bin >> 8);
				s.byte(n_cig & 0xff); s.byte(n_cig >> 8); s.byte(flag & 0xff); s.byte(flag >> 8);
				put32(s, l_seq);
				put32(s, (uint32_t) new_next_ref);
				put32(s, (uint32_t) new_next_pos);
				s.copy(r + 28, 4 + l_name);		// TLEN and the name
				for (uint32_t k = 0; k < n_cig; ++k) put32(s, cig[k]);
				s.copy_seq(r + 32 + l_name + 4 * n_cigar_old, l_seq);
				s.copy_qual(r + 32 + l_name + 4 * n_cigar_old + (l_seq + 1) / 2, l_seq, has_qual);
				// -- tags: the kept old ones and the new ones, merged by key
				uint32_t a = 0, b = 0;
				for (;;)
				{
					// the next new tag that applies to this record
					uint32_t nk = 0x10000, kind = 0;
					while (b < T.n_new)
					{
						kind = T.new_kind[b]; nk = T.new_key[b];
						bool applies = true;
						if (NEW_XB == kind || NEW_YB == kind) applies = n_runs > 0;
						else if (NEW_XI == kind) applies = xi <= 0x7fffffffull;
						else if (NEW_YR == kind) applies = next_ref >= 0;
						else if (NEW_YP == kind) applies = next_ref >= 0 && next_pos >= 0;
						if (applies) break;
						++b; nk = 0x10000;
					}
					uint32_t const ok = (a < n_kept) ? k_key[a][lane] : 0x10000;
					if (0x10000 == ok && 0x10000 == nk) break;
					if (ok < nk) { s.copy(r + k_off[a][lane], k_len[a][lane]); ++a; continue; }
					if (ok == nk) ++a;		// the new tag replaces the old one of the same name
					++b;
					switch (kind)
					{
						case NEW_OA:
						{
							// :805-844: RNAME,POS (0-based as the reference prints it),strand,CIGAR,MAPQ,NM; behind the record's old OA
							s.byte('O'); s.byte('A'); s.byte('Z');
							s.copy(r + old_oa_at, old_oa_len);
							uint32_t const n0 = T.ref_name_off[ref_id], n1 = T.ref_name_off[ref_id + 1];
							s.copy((uint8_t const *) T.ref_names + n0, n1 - n0);
							s.byte(','); put_dec(s, pos); s.byte(','); s.byte((flag & 0x10) ? '-' : '+'); s.byte(',');
							for (uint32_t k = 0; k < n_cigar_old; ++k)
							{
								uint32_t const cc = ld32(r + 32 + l_name + 4 * k);
								put_dec(s, cc >> 4);
								uint32_t const op = cc & 0xf;
								s.byte("MIDNSHP=X"[op < 9 ? op : 0]);
							}
							s.byte(','); put_dec(s, mapq); s.byte(',');
							if (have_nm) put_dec(s, nm);
							s.byte(';'); s.byte(0);
							break;
						}
						case NEW_XB: case NEW_YB:		// :848-861: B:I [start, end) pairs (the reference range's location: quirk Q7)
						{
							s.byte(nk >> 8); s.byte(nk & 0xff); s.byte('B'); s.byte('I');
							put32(s, 2 * n_runs);
							for (uint32_t k = 0; k < n_runs; ++k)
							{
								uint32_t const lo = (NEW_XB == kind) ? runs[k].query_location : runs[k].ref_position;
								uint32_t const hi = lo + ((NEW_XB == kind) ? runs[k].query_length : runs[k].ref_length);
								put32(s, lo); put32(s, hi);
							}
							break;
						}
						case NEW_XI: put_int_tag(s, nk, xi); break;
						case NEW_XR: put_int_tag(s, nk, (uint64_t) ref_id); break;		// :872-877
						case NEW_XP: put_int_tag(s, nk, (uint64_t) pos); break;
						case NEW_YR: put_int_tag(s, nk, (uint64_t) next_ref); break;
						default: put_int_tag(s, nk, (uint64_t) next_pos); break;		// NEW_YP
					}
				}
			};
			if (!EMIT)
			{
				count_sink s;
				body(s);
				out_size[j] = 4 + s.n;
			}
			else
			{
				uint64_t const o = out_off[j];
				write_sink s(out, o, block_bytes, block_stride, block_pad);
				put32(s, (uint32_t) (out_off[j + 1] - o - 4));
				body(s);
			}
		}
	}
}

__global__ void k_reset_state(rec_state *st)
{
	st->n_records = 0; st->tail_off = 0; st->n_kept = 0; st->n_take = 0; st->n_deferred = 0; st->carry_len = 0; st->frame_error_at = NONE;
	st->err_key = ~0ull;
	for (int k = 0; k < 4; ++k) st->counts[k] = 0;
}

inline uint32_t blocks_for(uint64_t n, uint32_t per) { return (uint32_t) ((n + per - 1) / per); }

}	// anonymous namespace

// ---- launch wrappers (all on `stream`; the return value is a hipError_t) ------------------------------------------------------------
int rec_launch_reset(rec_state *st, void *stream)
{
	hipLaunchKernelGGL(k_reset_state, dim3(1), dim3(1), 0, (hipStream_t) stream, st);
	return (int) hipGetLastError();
}

int rec_launch_frame(uint8_t const *U, uint32_t const *blk_off, uint32_t n_blocks, uint32_t start0, uint32_t end, uint32_t n_refs, rec_frame_work const &w, rec_state *st, void *stream)
{
	hipStream_t const s = (hipStream_t) stream;
	hipLaunchKernelGGL(k_frame_walk, dim3(blocks_for(n_blocks, WAVE)), dim3(WAVE), 0, s, U, blk_off, n_blocks, start0, end, n_refs, w.guess, w.cnt, w.exits, w.tails);
	hipLaunchKernelGGL(k_frame_fix, dim3(1), dim3(WAVE), 0, s, U, blk_off, n_blocks, start0, end, w.guess, w.cnt, w.exits, w.tails, w.rec_base, st);
	hipLaunchKernelGGL(k_frame_emit, dim3(blocks_for(n_blocks, WAVE)), dim3(WAVE), 0, s, U, blk_off, n_blocks, end, w.guess, w.rec_base, w.rec_off);
	return (int) hipGetLastError();
}

int rec_launch_scan(uint32_t const *in, uint32_t const *n_ptr, uint32_t n_max, uint64_t *out, uint32_t *tile_sums, uint64_t *tile_base, void *stream)
{
	hipStream_t const s = (hipStream_t) stream;
	uint32_t const n_tiles = blocks_for(n_max ? n_max : 1, SCAN_TILE);
	hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(SCAN_BLOCK), 0, s, in, n_ptr, n_max, out, tile_sums);
	hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, tile_sums, n_tiles, tile_base);
	hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(SCAN_BLOCK), 0, s, n_ptr, n_max, out, tile_base, n_tiles);
	return (int) hipGetLastError();
}

uint32_t rec_scan_tiles(uint32_t n_max) { return blocks_for(n_max ? n_max : 1, SCAN_TILE); }

int rec_launch_filter(uint8_t const *U, uint32_t n_max, rec_frame_work const &w, rec_state *st, uint32_t n_refs, uint32_t n_pending, uint32_t parity, uint64_t seen, void *stream)
{
	hipLaunchKernelGGL(k_filter, dim3(blocks_for(n_max ? n_max : 1, 256)), dim3(256), 0, (hipStream_t) stream, U, w.rec_off, st, n_refs, n_pending, parity, seen, w.keep);
	return (int) hipGetLastError();
}

int rec_launch_compact(uint32_t n_max, rec_frame_work const &w, rec_state *st, uint32_t n_pending, uint32_t parity, uint64_t seen, int last, void *stream)
{
	hipLaunchKernelGGL(k_compact, dim3(blocks_for(n_max ? n_max : 1, 256)), dim3(256), 0, (hipStream_t) stream, w.rec_off, w.keep, w.kept_pos, st, n_pending, parity, seen, last, w.kept_off, w.kept_input);
	return (int) hipGetLastError();
}

int rec_launch_carry(uint8_t const *U, uint32_t end, rec_state *st, uint8_t *carry, uint32_t carry_cap, void *stream)
{
	hipLaunchKernelGGL(k_carry, dim3(1), dim3(256), 0, (hipStream_t) stream, U, end, st, carry, carry_cap);
	return (int) hipGetLastError();
}

int rec_launch_scatter(uint8_t const *U, uint32_t end, uint32_t n_take, rec_frame_work const &w, rec_state const *st, rec_tables const &T, rec_batch_arrays const &B, void *stream)
{
	hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n_take ? n_take : 1, 256)), dim3(256), 0, (hipStream_t) stream, U, end, w.kept_off, st, T.src_entry, T.dst_entry,
		B.src_pos, B.src_entry, B.dst_entry, B.flags, B.l_seq, B.seq_off, B.qual_off, B.mate_se, B.mate_de, B.mate_pos, w.n_cigar);
	return (int) hipGetLastError();
}

int rec_launch_scatter_cigar(uint8_t const *U, uint32_t n_take, rec_frame_work const &w, rec_state const *st, uint64_t const *cigar_off, uint64_t cigar_cap, uint32_t *cigar, void *stream)
{
	hipLaunchKernelGGL(k_scatter_cigar, dim3(blocks_for(n_take ? n_take : 1, 256)), dim3(256), 0, (hipStream_t) stream, U, w.kept_off, st, cigar_off, cigar_cap, cigar);
	return (int) hipGetLastError();
}

int rec_launch_rewrite(bool emit, uint8_t const *U, uint32_t n_take, rec_frame_work const &w, rec_state *st, rec_tables const &T, rec_results const &R, uint32_t *out_size, uint64_t const *out_off,
	uint8_t *out, uint32_t block_bytes, uint32_t block_stride, uint32_t block_pad, void *stream)
{
	if (0 == n_take) return 0;
	if (emit)
		hipLaunchKernelGGL(k_rewrite <true>, dim3(blocks_for(n_take, WAVE)), dim3(WAVE), 0, (hipStream_t) stream, U, w.kept_off, w.kept_input, st, T, R, out_size, out_off, out, block_bytes, block_stride, block_pad);
	else
		hipLaunchKernelGGL(k_rewrite <false>, dim3(blocks_for(n_take, WAVE)), dim3(WAVE), 0, (hipStream_t) stream, U, w.kept_off, w.kept_input, st, T, R, out_size, out_off, out, block_bytes, block_stride, block_pad);
	return (int) hipGetLastError();
}

}
