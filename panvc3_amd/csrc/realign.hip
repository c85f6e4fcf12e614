// realign.hip — local realignment of indel runs (the compute-heavy leg of the hot path).
//
// The records whose rewritten CIGAR holds an indel run come as per-region work lists (k_project_mid / k_project_general).
// Five steps, all in record order of nothing: every step writes to places that only depend on the record or the run.
//
//   k_run_prefix      prefix sums of the per-region list lengths: the lists become one dense index space
//   k_realign_scan    lane = record.  Replays indel_run_checker (libpanvc3/indel_run_checker.cc:31-121) over the rewritten
//                     CIGAR; every run with both insertions and deletions becomes a pvc3_run (alignment_projector.cc:69-73)
//                     and a work item, binned by the shape of its alignment
//   k_realign_dp      lane = row (or column) of one run's alignment matrix, 4 / 2 / 1 runs per wavefront: global affine
//                     alignment of the run's reference slice against its query slice (include/panvc3/align.hh:59-112;
//                     seqan3::align_pairwise restated, see oracle/pvc3_oracle.c for the recurrence and the traceback
//                     priority), anti-diagonal by anti-diagonal, fp64 adds / compares only (the score table comes from
//                     the host's libm), four trace bits per cell, written to the context's trace buffer
//   k_realign_traceback   lane = run: walks the trace back and writes the run's operations run-length into its slot
//   k_realign_big     the same alignment for runs that do not fit the packed kernel: one run per wavefront, lane = query row
//                     in stripes of 64 rows, trace bytes in LDS or in a global scratch slot
//   k_realign_splice  lane = record.  Copies the untouched operations, splices the realigned runs
//                     (alignment_projector.cc:58-148) and merges adjacent equal operations (collapse_cigar_operations,
//                     libpanvc3/cigar.cc:13-51)
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

constexpr uint32_t REALIGN_LDS_BYTES = 4096;		// k_realign_big, per wave (= per block); larger alignments use the wave's global scratch slot
constexpr uint32_t TR_DIAG = 0, TR_UP = 1, TR_LEFT = 2, TR_DIR_MASK = 3, TR_V_OPEN = 4, TR_H_OPEN = 8, TR_MATCH = 16;
constexpr uint32_t MAX_PHRED = 41;
constexpr uint32_t NO_QUAL = MAX_PHRED + 1;			// row of the score table for records without base qualities: 0 / -1
constexpr uint32_t SCORE_ROWS = NO_QUAL + 1;
constexpr uint32_t SCAN_STAGE = 128;				// work items per class a wave of k_realign_scan collects before it appends them

// seqan3::dna5 ranks (A C G N T); assign_char is case-insensitive, U -> T, anything else -> N.
__device__ __forceinline__ uint32_t dna5_rank_of_char(uint32_t c)
{
	c &= ~0x20u;	// upper case for letters
	return ('A' == c) ? 0 : ('C' == c) ? 1 : ('G' == c) ? 2 : ('T' == c || 'U' == c) ? 4 : 3;
}

__device__ __forceinline__ uint32_t dna5_rank_of_seq4(uint32_t code)
{
	return (1 == code) ? 0 : (2 == code) ? 1 : (4 == code) ? 2 : (8 == code) ? 4 : 3;
}

__device__ __forceinline__ uint32_t seq4_at(uint8_t const *seq4, uint32_t i)
{
	uint32_t const b = seq4[i >> 1];
	return (i & 1) ? (b & 0xf) : (b >> 4);
}

// Lane i receives lane i - 1's value (lane 0 keeps its own): DPP wave_shr:1, no LDS round trip.
__device__ __forceinline__ double shfl_up_f64(double v)
{
	union { double d; int i[2]; } a, r;
	a.d = v;
	r.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], 0x138, 0xf, 0xf, false);
	r.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], 0x138, 0xf, 0xf, false);
	return r.d;
}

// Lane i of every row of 16 receives lane i - 1's value, the first lane of a row receives `first`: DPP row_shr:1.
__device__ __forceinline__ double row_shr1_f64(double v, double first)
{
	union { double d; int i[2]; } a, f, r;
	a.d = v; f.d = first;
	r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x111, 0xf, 0xf, false);
	r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x111, 0xf, 0xf, false);
	return r.d;
}

__device__ __forceinline__ void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- dense index space over the per-region work lists ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_run_prefix(uint32_t n_regions, dev_batch_work const W)
{
	__shared__ uint32_t lds[1024];
	uint32_t const per = (n_regions + 1023) / 1024;
	uint32_t const begin = threadIdx.x * per;
	uint32_t sum = 0;
	for (uint32_t i = begin; i < begin + per && i < n_regions; ++i)
		sum += W.run_count[i];
	lds[threadIdx.x] = sum;
	__syncthreads();
	for (uint32_t off = 1; off < 1024; off <<= 1)		// Hillis-Steele inclusive scan
	{
		uint32_t const t = (threadIdx.x >= off) ? lds[threadIdx.x - off] : 0;
		__syncthreads();
		lds[threadIdx.x] += t;
		__syncthreads();
	}
	uint32_t run = lds[threadIdx.x] - sum;
	for (uint32_t i = begin; i < begin + per && i < n_regions; ++i)
	{
		W.run_prefix[i] = run;
		run += W.run_count[i];
	}
	if (1023 == threadIdx.x)
	{
		W.run_prefix[n_regions] = lds[1023];
		W.counters->run_records = lds[1023];
	}
}

// Record number `i` of the concatenated work lists.
__device__ __forceinline__ uint32_t listed_record(dev_batch_work const &W, uint32_t n_regions, uint64_t n_groups, uint32_t i)
{
	uint32_t lo = 0, hi = n_regions;		// largest r with run_prefix[r] <= i (empty regions repeat a prefix: take the last)
	while (hi - lo > 1)
	{
		uint32_t const mid = (lo + hi) >> 1;
		if (W.run_prefix[mid] <= i) lo = mid; else hi = mid;
	}
	return W.run_list[64 * (n_groups * lo / n_regions) + (i - W.run_prefix[lo])];
}

// QUAL of the listed records from the caller's page-locked host arrays into the batch's compact device buffer (see
// dev_batch_work::host_qual): lane = record for the bookkeeping, the whole wave copies one record at a time.
__global__ __launch_bounds__(256) void k_gather_qual(dev_batch_in const in, dev_batch_work const W, uint32_t n_regions)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	uint64_t const n_groups = (in.n + 63) / 64;
	uint32_t const total = W.run_prefix[n_regions];
	for (uint32_t base = 64 * wave; base < total; base += 64 * n_waves)
	{
		bool const mine = base + lane < total;
		uint32_t rec = 0, len = 0;
		uint64_t src_off = 0;
		if (mine)
		{
			rec = listed_record(W, n_regions, n_groups, base + lane);
			if (in.flags[rec] & PVC3_FLAG_HAS_QUAL)
			{
				len = in.l_seq[rec];
				src_off = W.host_qual_off[rec];		// read over PCIe
			}
		}
		uint32_t const rounded = (len + 3) & ~3u;
		uint32_t const scan = wave_inclusive_scan(rounded, lane);
		uint32_t const sum = readlane(scan, 63);
		if (0 == sum) continue;
		unsigned long long at = 0;
		if (0 == lane) at = atomicAdd(&W.counters->qual_head, (unsigned long long) sum);
		at = rfl64(at);
		if (at + sum + 4 > W.qual_compact_cap)
		{
			W.counters->qual_overflow = 1;		// the host grows the buffer and repeats the batch
			continue;
		}
		uint64_t const dst_off = at + (scan - rounded);
		if (mine) W.qual_off_dev[rec] = dst_off;
		uint64_t todo = ballot(len > 0);
		while (todo)
		{
			uint32_t const r = ctz64(todo);
			todo &= todo - 1;
			uint8_t const *src = W.host_qual + rfl64(readlane((uint32_t) src_off, r) | ((uint64_t) readlane((uint32_t) (src_off >> 32), r) << 32));
			uint8_t *dst = W.qual_compact + rfl64(readlane((uint32_t) dst_off, r) | ((uint64_t) readlane((uint32_t) (dst_off >> 32), r) << 32));
			uint32_t const n = readlane(len, r);
			for (uint32_t x = 4 * lane; x < n; x += 256)
			{
				// whole dwords where the record has them (the destination is 4-byte aligned and padded), single bytes at its end:
				// nothing is read past the caller's array
				if (x + 4 <= n) { uint32_t v; __builtin_memcpy(&v, src + x, 4); *(uint32_t *) (dst + x) = v; }
				else for (uint32_t y = x; y < n; ++y) dst[y] = src[y];
			}
		}
	}
}

// The operations of a lane's record: the first STAGED_OPS are fetched with four 16-byte loads and parked in LDS
// (operation x of lane l at [x][l]: conflict-free), so that the replay loops below do not pay a global round trip per
// operation; longer CIGARs read the rest from memory.
constexpr uint32_t STAGED_OPS = 16;
struct op_source
{
	uint32_t const *ops;		// global
	uint32_t const *staged;		// LDS, this lane's column
	__device__ __forceinline__ uint32_t operator[](uint32_t x) const { return (x < STAGED_OPS) ? staged[64 * x] : ops[x]; }
};

__device__ __forceinline__ op_source stage_ops(uint32_t const *ops, uint32_t n_ops, uint32_t *lds_wave, uint32_t lane)
{
	typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
	u32x4 v[STAGED_OPS / 4];
#pragma unroll
	for (uint32_t k = 0; k < STAGED_OPS / 4; ++k)
	{
		v[k] = u32x4{0, 0, 0, 0};
		if (4 * k < n_ops) __builtin_memcpy(&v[k], ops + 4 * k, 16);		// the arena is padded: reading past the record's last operation is harmless
	}
#pragma unroll
	for (uint32_t k = 0; k < STAGED_OPS / 4; ++k)
	{
		lds_wave[64 * (4 * k + 0) + lane] = v[k].x; lds_wave[64 * (4 * k + 1) + lane] = v[k].y;
		lds_wave[64 * (4 * k + 2) + lane] = v[k].z; lds_wave[64 * (4 * k + 3) + lane] = v[k].w;
	}
	return op_source{ops, lds_wave + lane};
}

// indel_run_checker (indel_run_checker.cc:13-121) over `n_ops` operations; calls on_run(first op, end op, reference start,
// reference length, query start, query length, reference position at report time) for every run with both I and D.
template <typename t_ops, typename t_on_run>
__device__ __forceinline__ void replay_runs(t_ops const &ops, uint32_t n_ops, uint32_t dst_pos, t_on_run &&on_run)
{
	uint32_t ref_pos = dst_pos, query_pos = 0, run_type = 0;
	uint32_t run_first = 0, run_ref_loc = dst_pos, run_query_loc = 0;
	for (uint32_t x = 0; x <= n_ops; ++x)
	{
		bool const at_end = (x == n_ops);
		uint32_t op = 0xf, cnt = 0;
		if (!at_end)
		{
			uint32_t const c = ops[x];
			op = c & 0xf; cnt = c >> 4;
			if (0 == run_type) { run_ref_loc = ref_pos; run_query_loc = query_pos; run_first = x; }	// :45-50
			if (OP_I == op) { query_pos += cnt; run_type |= 1; continue; }							// :62-65
			if (OP_D == op) { ref_pos += cnt; run_type |= 2; continue; }							// :67-70
		}
		// A run-terminating operation (:55-60, :72-94) or the end of the CIGAR (:111-118).
		uint32_t const prev_type = run_type;
		uint32_t const run_ref_end = ref_pos, run_query_end = query_pos;
		run_type = 0;
		if (!at_end)
		{
			if (OP_S == op) query_pos += cnt;
			else if (OP_N == op) ref_pos += cnt;
			else if (OP_M == op || OP_EQ == op || OP_X == op) { ref_pos += cnt; query_pos += cnt; }
		}
		if (3 == prev_type)
			on_run(run_first, x, run_ref_loc, run_ref_end - run_ref_loc, run_query_loc, run_query_end - run_query_loc, ref_pos);
	}
}

__device__ __forceinline__ uint32_t dp_class_of(uint32_t m, uint32_t n, uint32_t packed)
{
	uint32_t const shorter = (m < n) ? m : n, longer = (m < n) ? n : m;
	if (!packed || shorter > 64 || longer + shorter - 1 > DP_MAX_STEPS)
		return DP_CLASS_BIG;
	uint32_t const lanes = (shorter + 3) / 4;
	uint32_t const w = (lanes <= 3) ? 0 : (lanes <= 4) ? 1 : (lanes <= 6) ? 2 : (lanes <= 8) ? 3 : (lanes <= 12) ? 4 : 5;
	return 2 * w + ((m < n) ? 1 : 0);		// orientation 1: lines along the reference (the shorter side)
}

__global__ __launch_bounds__(256) void k_realign_scan(dev_entry const *__restrict__ entries, dev_batch_in const in, dev_batch_work const W, uint32_t n_regions, uint32_t packed)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	uint64_t const n_groups = (in.n + 63) / 64;
	uint32_t const total = W.run_prefix[n_regions];
	arena_alloc arena{0, 0}, runs_alloc{0, 0};
	uint32_t acc_runs = 0, acc_failed = 0;
	uint64_t acc_qual = 0;
	// The work items of a class are collected per wave in LDS and appended to the class list with one atomic per flush:
	// an atomic per class and 64 records (half a million of them on the same cache line in config C5) made this kernel
	// five times slower than the rest of its work.
	__shared__ uint32_t lds_items[4][DP_CLASSES][SCAN_STAGE];
	uint32_t (*stage)[SCAN_STAGE] = lds_items[rfl(threadIdx.x >> 6)];
	__shared__ uint32_t lds_ops[4][STAGED_OPS * 64];
	uint32_t *my_ops = lds_ops[rfl(threadIdx.x >> 6)];
	uint32_t staged[DP_CLASSES];		// wave-uniform
#pragma unroll
	for (uint32_t c = 0; c < DP_CLASSES; ++c) staged[c] = 0;
	auto flush = [&](uint32_t c) {
		uint32_t const cnt = staged[c];
		uint32_t at = 0;
		if (0 == lane) at = atomicAdd(&W.counters->dp_count[c], cnt);
		at = rfl(at);
		for (uint32_t i = lane; i < cnt; i += 64)
			W.dp_list[(uint64_t) c * W.runs_cap + at + i] = stage[c][i];
		staged[c] = 0;
	};

	for (uint32_t base = 64 * wave; base < total; base += 64 * n_waves)
	{
		bool const mine = base + lane < total;
		uint32_t rec = 0, n_ops = 0, dst_pos = 0, text_len = 0, l_seq = 0;
		uint32_t const *ops = W.arena;
		uint8_t const *text = nullptr, *seq4 = nullptr, *qual = nullptr;
		bool has_text = false, has_qual = false;
		if (mine)
		{
			rec = listed_record(W, n_regions, n_groups, base + lane);
			ops = W.arena + W.cig_loc[rec];
			n_ops = W.cig_n[rec];
			dst_pos = W.dst_pos[rec];
			dev_entry const *de = entries + in.dst_entry[rec];
			text = de->text; text_len = de->text_len;
			has_text = nullptr != text;
			l_seq = in.l_seq[rec];
			seq4 = in.seq4 + in.seq_off[rec];
			has_qual = (in.flags[rec] & PVC3_FLAG_HAS_QUAL) && in.qual;
			if (has_qual) qual = in.qual + in.qual_off[rec];
		}

		wave_lds_fence();		// the previous chunk's operations have been read
		op_source const staged_ops = stage_ops(ops, n_ops, my_ops, lane);
		wave_lds_fence();
		// Pass 1: count the runs, the slots they need and check their ranges (alignment_projector.cc:76-137 slices both sequences).
		uint32_t n3 = 0, slot_need = 0, status = PVC3_ST_OK;
		replay_runs(staged_ops, n_ops, dst_pos, [&](uint32_t, uint32_t, uint32_t r_loc, uint32_t m, uint32_t q_loc, uint32_t nq, uint32_t) {
			if (status) return;
			if (!has_text || (uint64_t) r_loc + m > text_len) { status = PVC3_ST_REF_RANGE; return; }
			if ((uint64_t) q_loc + nq > l_seq) { status = PVC3_ST_QUERY_RANGE; return; }
			++n3; slot_need += m + nq;
		});
		if (mine && status)
		{
			W.cig_n[rec] = 0; W.run_n[rec] = 0; W.dst_pos[rec] = 0;		// a rejected record has empty outputs
			W.rec_flags[rec] = status;
			++acc_failed;
			n3 = 0; slot_need = 0;
		}
		// One piece of the arena per record: its final CIGAR has at most one operation per realigned base plus the untouched
		// operations.  The runs' slots (m + n operations each, filled from their end by the alignment kernels) lie inside the same
		// piece, run k behind the untouched operations that precede it and the slots of the runs before it: k_realign_splice writes
		// the piece from its start and can never reach an operation of a run before it has read it.
		uint32_t const need = (mine && !status && n3) ? (n_ops + slot_need + 1) : 0;
		uint32_t const run_scan = wave_inclusive_scan(n3, lane), need_scan = wave_inclusive_scan(need, lane);
		uint32_t const run_total = readlane(run_scan, 63), need_total = readlane(need_scan, 63);
		uint64_t run_base = 0, arena_base = 0;
		bool ok = true;
		if (run_total) ok = runs_take(runs_alloc, run_total, W, lane, run_base);
		if (need_total) ok = arena_take(arena, need_total, W, lane, arena_base) && ok;
		if (!ok)
			continue;		// the overflow flag is set; the host grows the arenas and repeats the batch

		// Pass 2: the pvc3_run entries, the work items and the class lists.
		uint64_t const my_runs = run_base + (run_scan - n3);
		uint64_t const my_arena = arena_base + (need_scan - need);
		uint32_t k = 0, slot = 0, run_ops = 0;
		uint32_t cls[2] = {DP_CLASSES, DP_CLASSES};		// the first two runs of the lane go to the class lists wave-wide
		if (mine && !status)
		{
			replay_runs(staged_ops, n_ops, dst_pos, [&](uint32_t first, uint32_t end, uint32_t r_loc, uint32_t m, uint32_t q_loc, uint32_t nq, uint32_t ref_pos) {
				pvc3_run rr;
				rr.query_location = q_loc; rr.query_length = nq; rr.ref_position = ref_pos; rr.ref_length = m;		// alignment_projector.cc:69-73 (quirk Q7)
				W.runs_arena[my_runs + k] = rr;
				run_work it;
				it.rec = rec; it.ref_loc = r_loc; it.query_loc = q_loc; it.m = m; it.n = nq; it.n_out = 0;
				it.op_first = first; it.op_end = end;
				it.ops_loc = my_arena + (first - run_ops) + slot;
				it.splice_loc = my_arena;
				it.text = text + r_loc; it.seq4 = seq4; it.qual = qual; it.trace_loc = TRACE_NONE;
				W.run_items[my_runs + k] = it;
				uint32_t const c = dp_class_of(m, nq, packed);
				if (k < 2) cls[k] = c;
				else
				{
					uint32_t const at = atomicAdd(&W.counters->dp_count[c], 1u);		// third and later runs of one record: rare
					W.dp_list[(uint64_t) c * W.runs_cap + at] = (uint32_t) (my_runs + k);
				}
				slot += m + nq; run_ops += end - first; ++k;
			});
			W.run_loc[rec] = my_runs; W.run_n[rec] = n3;
			acc_runs += n3;
			acc_qual += has_qual ? l_seq : 0;
		}
#pragma unroll
		for (uint32_t r = 0; r < 2; ++r)
		{
			if (0 == ballot(cls[r] < DP_CLASSES)) continue;
#pragma unroll
			for (uint32_t c = 0; c < DP_CLASSES; ++c)
			{
				uint64_t const mk = ballot(cls[r] == c);
				if (!mk) continue;
				if (staged[c] + 64 > SCAN_STAGE) { wave_lds_fence(); flush(c); wave_lds_fence(); }
				if ((mk >> lane) & 1) stage[c][staged[c] + mbcnt64(mk)] = (uint32_t) (my_runs + r);
				staged[c] += popc64(mk);
			}
		}
	}
	// What is left goes out once per block: the four waves add up their counts, one thread per class reserves the block's
	// piece of the class list (the counters of all classes share a cache line: thousands of atomics on it take longer than the
	// rest of this kernel on a small batch), then every wave copies its items.
	__shared__ uint32_t blk_cnt[4][DP_CLASSES], blk_base[DP_CLASSES];
	__shared__ unsigned long long blk_acc[4][3];
	for (int off = 32; off; off >>= 1)
	{
		acc_runs += __shfl_xor(acc_runs, off);
		acc_failed += __shfl_xor(acc_failed, off);
		acc_qual += __shfl_xor(acc_qual, off);
	}
	uint32_t const wv = rfl(threadIdx.x >> 6);
	if (0 == lane)
	{
#pragma unroll
		for (uint32_t c = 0; c < DP_CLASSES; ++c) blk_cnt[wv][c] = staged[c];
		blk_acc[wv][0] = acc_failed; blk_acc[wv][1] = acc_runs; blk_acc[wv][2] = acc_qual;
	}
	__syncthreads();
	if (threadIdx.x < DP_CLASSES)
	{
		uint32_t const tot = blk_cnt[0][threadIdx.x] + blk_cnt[1][threadIdx.x] + blk_cnt[2][threadIdx.x] + blk_cnt[3][threadIdx.x];
		blk_base[threadIdx.x] = tot ? atomicAdd(&W.counters->dp_count[threadIdx.x], tot) : 0;
	}
	else if (threadIdx.x >= 64 && threadIdx.x < 67)
	{
		uint32_t const k = threadIdx.x - 64;
		unsigned long long const tot = blk_acc[0][k] + blk_acc[1][k] + blk_acc[2][k] + blk_acc[3][k];
		if (tot) atomicAdd((0 == k) ? &W.counters->n_failed : (1 == k) ? &W.counters->n_runs : &W.counters->qual_bytes, tot);
	}
	__syncthreads();
#pragma unroll
	for (uint32_t c = 0; c < DP_CLASSES; ++c)
	{
		uint32_t at = blk_base[c];
		for (uint32_t w2 = 0; w2 < wv; ++w2) at += blk_cnt[w2][c];
		for (uint32_t i = lane; i < staged[c]; i += 64)
			W.dp_list[(uint64_t) c * W.runs_cap + at + i] = stage[c][i];
	}
}


// ---- the class lists in pieces of equal length --------------------------------------------------------------------------------------
// A task of k_realign_dp takes as long as the longest of its 4 - 21 runs, and the runs come in the order of the records: on C5, where the
// longer side is anything from 13 to 60, 54 % of the lane-steps of a task were cells as the lists stood and 67 % are with every 1 024
// consecutive entries of a list sorted by the steps their runs take (a sort of the whole list would give 68 %).  One block per piece:
// key | index as one 64-bit word, a bitonic network in LDS.
constexpr uint32_t SORT_PIECE = 1024, SORT_THREADS = 256;
__global__ __launch_bounds__(SORT_THREADS) void k_realign_sort(dev_batch_work const W)
{
	__shared__ unsigned long long keyed[SORT_PIECE];
	uint32_t const cls = blockIdx.y;
	uint32_t const count = W.counters->dp_count[cls];
	uint32_t *list = W.dp_list + (uint64_t) cls * W.runs_cap;
	bool const ref_lanes = cls & 1;
	for (uint32_t base = blockIdx.x * SORT_PIECE; base < count; base += gridDim.x * SORT_PIECE)
	{
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < SORT_PIECE; i += SORT_THREADS)
		{
			unsigned long long v = ~0ull;		// (behind the list's end: sorts last, is not written back)
			if (base + i < count)
			{
				uint32_t const idx = list[base + i];
				uint32_t const m = W.run_items[idx].m, n = W.run_items[idx].n;
				uint32_t const n_short = ref_lanes ? m : n, n_long = ref_lanes ? n : m;
				v = ((unsigned long long) (n_long + (n_short + DP_LINES_PER_LANE - 1) / DP_LINES_PER_LANE) << 32) | idx;
			}
			keyed[i] = v;
		}
		__syncthreads();
		for (uint32_t k = 2; k <= SORT_PIECE; k <<= 1)
			for (uint32_t j = k >> 1; j; j >>= 1)
			{
				for (uint32_t t = threadIdx.x; t < SORT_PIECE / 2; t += SORT_THREADS)
				{
					uint32_t const lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;		// the t-th pair j apart
					unsigned long long const a = keyed[lo], b = keyed[hi];
					bool const up = 0 == (lo & k);
					if ((a > b) == up) { keyed[lo] = b; keyed[hi] = a; }
				}
				__syncthreads();
			}
		for (uint32_t i = threadIdx.x; i < SORT_PIECE; i += SORT_THREADS)
			if (base + i < count) list[base + i] = (uint32_t) keyed[i];
	}
}

// ---- packed alignment: t_lanes lanes per run with four lines of the matrix each, 64 / t_lanes runs per wavefront ---------------
// Matrix M(i, j), i = 0..n over the query (vertical), j = 0..m over the reference (horizontal); V = best score ending in a
// vertical gap (query base against nothing: 'I'), H = horizontal gap ('D').  The lines of the matrix along the shorter side of
// the run (t_ref_lanes: columns j, otherwise rows i) are dealt out four to a lane, lines 4l + 1 .. 4l + 4 to lane l of the run;
// a lane walks along the longer side, one position — four cells, from its first line down to its fourth — per step, lane l one
// step behind lane l - 1.  What the first of a lane's lines needs from the line before it (its M and its gap score towards
// this line, at the same position) is what lane l - 1 computed for its fourth line one step earlier and arrives by DPP; the
// other three lines take it from the cell computed just before in the same lane; a line's own M and gap score stay in registers.
//
// Until round 4 a lane owned ONE line, so a step computed one anti-diagonal: a run of s x L cells took s + L - 1 steps of
// max(16, 32, 64) lanes, of which 35 - 50 % were cells (C5: s up to 58, L around 45) at 36 vector instructions a step.  With
// four lines per lane the fill and drain of the wavefront is a quarter as long (L + s / 4 - 1 steps) and what a step costs
// beside its cells (the exchange, the stream, the trace) is paid once for four of them.
constexpr uint32_t DP_LINES = DP_LINES_PER_LANE;
struct dp_shared
{
	double score[SCORE_ROWS][2];					// [phred or NO_QUAL][0 match, 1 mismatch]
	uint16_t stream[21][64 + DP_MAX_STEPS + 8];		// per run: the items along the longer side, dna5 rank | phred << 8, behind 64 unused entries
};
constexpr uint32_t TRACE_CHUNK = 16384;				// trace words a wave takes from the trace buffer at a time (4 - 64 tasks)

struct dp_constants { double open, ge, go, bnd_step, neg_inf; };

// One task: 64 / t_lanes runs of class `cls`, from the task-th group of its list.
template <uint32_t t_lanes, bool t_ref_lanes>
__device__ __forceinline__ void realign_dp_task(dev_batch_work const &W, uint32_t cls, uint32_t task, uint32_t count, dp_constants const &K, dp_shared &S, arena_alloc &trace_alloc, uint32_t trace_chunk)
{
	constexpr uint32_t G = 64 / t_lanes;		// (3, 6, 12 lanes: 63, 60, 60 lanes of the wave at work)
	uint32_t const lane = lane_id();
	bool const idle = lane >= G * t_lanes;
	uint32_t const g = idle ? 0 : (lane / t_lanes), l = lane % t_lanes;
	double const open = K.open, ge = K.ge, go = K.go, bnd_step = K.bnd_step, neg_inf = K.neg_inf;
	uint32_t const *list = W.dp_list + (uint64_t) cls * W.runs_cap;
#ifdef PVC3_DP_STAMPS
	unsigned long long stamp_acc[4] = {0, 0, 0, 1};		// set-up, alignment, -, tasks
	unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define PVC3_STAMP(k) do { unsigned long long const now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_t; stamp_t = now_; } while (0)
#else
#define PVC3_STAMP(k) do {} while (0)
#endif
	{
		wave_lds_fence();		// the previous task's LDS reads are done
		uint32_t const slot_idx = task * G + g;
		bool const have = !idle && slot_idx < count;
		uint32_t const item_idx = have ? list[slot_idx] : 0;
		uint32_t it_m = 0, it_n = 0, it_q0 = 0;
		uint8_t const *text = nullptr, *seq4 = nullptr, *qual = nullptr;
		if (have)
		{
			run_work const *it = W.run_items + item_idx;
			it_m = it->m; it_n = it->n; it_q0 = it->query_loc;
			text = it->text; seq4 = it->seq4; qual = it->qual;
		}
		uint32_t const n_short = t_ref_lanes ? it_m : it_n, n_long = t_ref_lanes ? it_n : it_m;
		// the last lane of the run with a line is lane (n_short - 1) / 4: it reaches the last position in step n_long - 1 + that
		uint32_t steps = (have && n_short) ? (n_long + (n_short - 1) / DP_LINES) : 0, longest = have ? n_long : 0;
#pragma unroll
		for (int off = 32; off; off >>= 1)
		{
			uint32_t const o = (uint32_t) __shfl_xor((int) steps, off), o2 = (uint32_t) __shfl_xor((int) longest, off);
			steps = (o > steps) ? o : steps;
			longest = (o2 > longest) ? o2 : longest;
		}
		steps = rfl(steps); longest = rfl(longest);

		// The run's stream (longer side) and the lane's own items (its four lines of the shorter side): rank | phred << 8.  The
		// loads of eight items are issued before any of them is used.
		auto load_raw = [&](bool query_side, uint32_t x, uint32_t &a, uint32_t &b) {
			a = 0; b = 0x100;		// 0x100: no base quality
			if (query_side) { a = seq4[(it_q0 + x) >> 1]; if (qual) b = qual[it_q0 + x]; }
			else a = text[x];
		};
		auto item_of = [&](bool query_side, uint32_t x, uint32_t a, uint32_t b) -> uint32_t {
			if (!query_side) return dna5_rank_of_char(a) | (NO_QUAL << 8);
			uint32_t const code = ((it_q0 + x) & 1) ? (a & 0xf) : (a >> 4);
			uint32_t const ph = (0x100 == b) ? NO_QUAL : ((b > MAX_PHRED) ? MAX_PHRED : b);		// phred42 saturates
			return dna5_rank_of_seq4(code) | (ph << 8);
		};
		for (uint32_t x0 = 0; x0 < longest; x0 += 8 * t_lanes)
		{
			uint32_t ra[8], rb[8];
#pragma unroll
			for (uint32_t u = 0; u < 8; ++u)
			{
				uint32_t const x = x0 + l + u * t_lanes;
				ra[u] = 0; rb[u] = 0x100;
				if (x < n_long) load_raw(t_ref_lanes, x, ra[u], rb[u]);
			}
#pragma unroll
			for (uint32_t u = 0; u < 8; ++u)
			{
				uint32_t const x = x0 + l + u * t_lanes;
				if (x < n_long) S.stream[g][64 + x] = (uint16_t) item_of(t_ref_lanes, x, ra[u], rb[u]);
			}
		}
		uint32_t own[DP_LINES];
		{
			uint32_t ra[DP_LINES], rb[DP_LINES];
#pragma unroll
			for (uint32_t k = 0; k < DP_LINES; ++k)
			{
				ra[k] = 0; rb[k] = 0x100;
				if (DP_LINES * l + k < n_short) load_raw(!t_ref_lanes, DP_LINES * l + k, ra[k], rb[k]);
			}
#pragma unroll
			for (uint32_t k = 0; k < DP_LINES; ++k)
				own[k] = (DP_LINES * l + k < n_short) ? item_of(!t_ref_lanes, DP_LINES * l + k, ra[k], rb[k]) : (3u | (NO_QUAL << 8));
		}

		// Trace of this task: a 64-bit word per lane and four steps (a nibble per cell, 16 bits per step), 64 of them a row.
		uint64_t trace_loc = 0;
		bool const traced = arena_take_from(trace_alloc, ((steps + 3) / 4) * 128, trace_chunk, &W.counters->trace_head, W.trace_cap, &W.counters->trace_overflow, lane, trace_loc);
		if (have && 0 == l) W.run_items[item_idx].trace_loc = traced ? trace_loc : TRACE_NONE;
		if (!traced)
			return;			// the host grows the trace buffer and repeats the batch
		uint64_t *trace = reinterpret_cast <uint64_t *>(W.trace + trace_loc) + lane;
		wave_lds_fence();
		PVC3_STAMP(0);

		// Line 0 of the matrix (the first row, or the first column) feeds lane 0 of the run: M = the boundary gap, gap score -inf.
		// gap(k) = k * (open + ext) when the opening score is positive, open + k * ext otherwise (small integers: exact).
		auto gap = [&](uint32_t k) -> double { return (0 == k) ? 0.0 : ((open > 0.0) ? (double) k * go : open + (double) k * ge); };
		double own_m[DP_LINES], own_b[DP_LINES];		// per line: M and the along-the-line gap score of the cell before the next one
#pragma unroll
		for (uint32_t k = 0; k < DP_LINES; ++k) { own_m[k] = gap(DP_LINES * l + k + 1); own_b[k] = neg_inf; }
		double diag0 = gap(DP_LINES * l);				// M of the line before the lane's first, one position back
		double out_m = 0.0, out_a = neg_inf;			// what the next lane receives: the fourth line's M and its across-the-lines gap score
		double bnd = go - bnd_step;						// the boundary line's M at the position of the step: gap(t + 1) = go + t * bnd_step (small integers: exact)
		uint64_t acc = 0;
		// Lane l is at position t - l + 1 in step t; the 64 entries in front of the stream make the index non-negative (whatever
		// they hold is read by lanes that are not active yet).
		uint16_t const *my_stream = &S.stream[g][64 - l];
		uint32_t item_next = my_stream[0];
		for (uint32_t t = 0; t < steps; ++t)
		{
			bnd += bnd_step;
			double nb_m, nb_a;
			if constexpr (16 == t_lanes) { nb_m = row_shr1_f64(out_m, bnd); nb_a = row_shr1_f64(out_a, neg_inf); }		// (a run = a row of the DPP)
			else
			{
				nb_m = shfl_up_f64(out_m); nb_a = shfl_up_f64(out_a);
				if (0 == l) { nb_m = bnd; nb_a = neg_inf; }
			}
			uint32_t const pos = t - l + 1;				// 1-based position along the longer side handled in this step
			bool const stepping = (t >= l) && (pos <= n_long);
			uint32_t const item = item_next;
			item_next = my_stream[t + 1];				// t + 1 <= steps <= DP_MAX_STEPS: inside the array for every lane
			uint32_t nibs = 0;
			if (stepping)
			{
				double in_m = nb_m, in_a = nb_a, dg = diag0;
#pragma unroll
				for (uint32_t k = 0; k < DP_LINES; ++k)
				{
					if (DP_LINES * l + k < n_short)
					{
						bool const is_match = (item & 7) == (own[k] & 7);
						double const s = S.score[t_ref_lanes ? (item >> 8) : (own[k] >> 8)][is_match ? 0 : 1];
						// The larger value wins and ties keep the extension: max() gives the value (equal operands are the same number,
						// no NaN can arise from finite scores and -inf), the comparison only feeds the trace bit.
						double const a_open = in_m + go, a_ext = in_a + ge;
						double const a = __builtin_fmax(a_ext, a_open); bool const a_opened = a_ext < a_open;
						double const b_open = own_m[k] + go, b_ext = own_b[k] + ge;
						double const b = __builtin_fmax(b_ext, b_open); bool const b_opened = b_ext < b_open;
						// lanes along the query: a = V (from the row above), b = H; lanes along the reference: a = H (from the column to the left), b = V
						double const v = t_ref_lanes ? b : a, h = t_ref_lanes ? a : b;
						bool const v_opened = t_ref_lanes ? b_opened : a_opened, h_opened = t_ref_lanes ? a_opened : b_opened;
						double const d = dg + s;
						uint32_t state = is_match ? 0u : 1u;		// 0 diagonal match, 1 diagonal mismatch, 2 up, 3 left
						double const dv = __builtin_fmax(d, v);
						if (d < v) state = 2;
						double const best = __builtin_fmax(dv, h);
						if (dv < h) state = 3;
						nibs |= (state | (v_opened ? 4u : 0u) | (h_opened ? 8u : 0u)) << (4 * k);
						dg = own_m[k];			// this line's M one position back: the next line's diagonal
						own_m[k] = best; own_b[k] = b;
						in_m = best; in_a = a;
					}
				}
				diag0 = nb_m;
				out_m = in_m; out_a = in_a;
			}
			acc |= (uint64_t) nibs << (16 * (t & 3));
			if (3 == (t & 3) || t + 1 == steps)
			{
				trace[(t >> 2) * 64] = acc;			// 512 contiguous bytes per wave
				acc = 0;
			}
		}
		PVC3_STAMP(1);
	}
#ifdef PVC3_DP_STAMPS
	// diagnostic build only: cycle sums per phase and class into the wave statistics area (nothing else reads it afterwards)
	if (0 == lane)
		for (uint32_t k = 0; k < 4; ++k)
			atomicAdd(&W.wave_stats[8 * cls + k], stamp_acc[k]);
#endif
}

// One launch for the twelve classes: their tasks one range behind the other, dealt out to the waves round robin, so that a class
// with many runs gets as many waves as it has work for (a grid per class, as it was, gave every class the same number of waves:
// the classes with most of the cells were still at it when the others' waves had long gone).
__global__ __launch_bounds__(64, 4) void k_realign_dp(dev_batch_work const W, dev_params const *__restrict__ prm)
{
	__shared__ dp_shared S;
	__shared__ uint32_t first_task[DP_CLASS_BIG + 1], class_count[DP_CLASS_BIG];
	uint32_t const lane = lane_id();
	{
		uint32_t const c = (lane < DP_CLASS_BIG) ? lane : 0;
		uint32_t const cnt = W.counters->dp_count[c], per_task = 64 / dp_lanes_of_class(c);
		uint32_t tasks = (lane < DP_CLASS_BIG) ? ((cnt + per_task - 1) / per_task) : 0, before = tasks;
		for (uint32_t d = 1; d < 16; d <<= 1)
		{
			uint32_t const y = (uint32_t) __shfl_up((int) before, (int) d);
			if (lane >= d) before += y;
		}
		if (lane < DP_CLASS_BIG) { first_task[lane + 1] = before; class_count[lane] = cnt; }
		if (0 == lane) first_task[0] = 0;
	}
	wave_lds_fence();
	uint32_t const total = first_task[DP_CLASS_BIG];
	if (blockIdx.x >= total) return;
	for (uint32_t i = lane; i < SCORE_ROWS; i += 64)
	{
		S.score[i][0] = (i <= MAX_PHRED) ? prm->s_match[i] : 0.0;		// NO_QUAL: nucleotide_scoring_scheme<> default 0 / -1
		S.score[i][1] = (i <= MAX_PHRED) ? prm->s_mismatch[i] : -1.0;
	}
	dp_constants K;
	K.open = (double) prm->gap_open; K.ge = (double) prm->gap_ext;
	K.go = K.open + K.ge;									// seqan3's internal gap-open score = open + extension
	K.bnd_step = (K.open > 0.0) ? K.go : K.ge;				// first row / column: x_1 = go, x_k = x_{k-1} + max(go, ge), ties keep the extension
	K.neg_inf = -__builtin_inf();
	arena_alloc trace_alloc{0, 0};
	// A wave takes trace space for several tasks at a time (one atomic per chunk), but never much more than its own tasks can
	// use: with few runs every wave has one or two tasks, and 8192 waves x 64 KB would be half a gigabyte of reservations.
	constexpr uint32_t TASK_WORDS = ((DP_MAX_STEPS + 16 + 3) / 4) * 128;		// trace words of a task at most
	uint32_t const my_tasks = (total - blockIdx.x + gridDim.x - 1) / gridDim.x;
	uint32_t const trace_chunk = (my_tasks * TASK_WORDS < TRACE_CHUNK) ? (my_tasks * TASK_WORDS) : TRACE_CHUNK;
	// This wave's tasks are blockIdx.x, blockIdx.x + gridDim.x, ...: class by class, so that each class's code is a loop of its own
	// (all twelve bodies inside one loop share one register allocation, and spill).
	auto tasks_of = [&](uint32_t cls, auto &&one_task) {
		uint32_t const lo = first_task[cls], hi = first_task[cls + 1], n = gridDim.x;
		uint32_t const behind = (blockIdx.x + n - lo % n) % n;		// the first of this wave's tasks at or behind lo
		for (uint32_t at = lo + behind; at < hi; at += n)
			one_task(at - lo, class_count[cls]);
	};
	tasks_of(0, [&](uint32_t task, uint32_t count) { realign_dp_task <3, false>(W, 0, task, count, K, S, trace_alloc, trace_chunk); });		// (dp_lanes_of_class)
	tasks_of(1, [&](uint32_t task, uint32_t count) { realign_dp_task <3, true>(W, 1, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(2, [&](uint32_t task, uint32_t count) { realign_dp_task <4, false>(W, 2, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(3, [&](uint32_t task, uint32_t count) { realign_dp_task <4, true>(W, 3, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(4, [&](uint32_t task, uint32_t count) { realign_dp_task <6, false>(W, 4, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(5, [&](uint32_t task, uint32_t count) { realign_dp_task <6, true>(W, 5, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(6, [&](uint32_t task, uint32_t count) { realign_dp_task <8, false>(W, 6, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(7, [&](uint32_t task, uint32_t count) { realign_dp_task <8, true>(W, 7, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(8, [&](uint32_t task, uint32_t count) { realign_dp_task <12, false>(W, 8, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(9, [&](uint32_t task, uint32_t count) { realign_dp_task <12, true>(W, 9, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(10, [&](uint32_t task, uint32_t count) { realign_dp_task <16, false>(W, 10, task, count, K, S, trace_alloc, trace_chunk); });
	tasks_of(11, [&](uint32_t task, uint32_t count) { realign_dp_task <16, true>(W, 11, task, count, K, S, trace_alloc, trace_chunk); });
}

// Traceback (trace_iterator_base) of the packed alignments, lane = run: diagonal > up > left where the direction is decided;
// a gap continues until the cell whose gap score was formed by opening.  Cell (i, j) lies on line x (j for lanes along the
// reference, i otherwise) at position p (the other one): it was computed by lane (x - 1) / 4 of the run in step p - 1 + that
// lane, as the ((x - 1) mod 4)-th nibble of the step's sixteen bits.  The operations come out last to first and are written
// run-length from the end of the run's slot.  One cell per iteration, written as data flow: the nibble of the cell, the
// direction taken, the operation, the next state.
__global__ __launch_bounds__(256) void k_realign_traceback(dev_batch_work const W)
{
	uint32_t const cls = blockIdx.y;
	uint32_t const lanes = dp_lanes_of_class(cls), per_task = 64 / lanes;
	bool const ref_lanes = cls & 1;
	uint32_t const count = W.counters->dp_count[cls];
	uint32_t const *list = W.dp_list + (uint64_t) cls * W.runs_cap;
	for (uint32_t at = blockIdx.x * blockDim.x + threadIdx.x; at < count; at += gridDim.x * blockDim.x)
	{
		uint32_t const item_idx = list[at];
		run_work const *it = W.run_items + item_idx;
		uint64_t const trace_loc = it->trace_loc;
		if (TRACE_NONE == trace_loc)
			continue;		// the trace buffer overflowed: the batch is repeated
		uint32_t const m = it->m, n = it->n;
		uint64_t const ops_loc = it->ops_loc;
		uint64_t const *trace = reinterpret_cast <uint64_t const *>(W.trace + trace_loc) + (at % per_task) * lanes;		// + lane of the run
		uint32_t *slot_end = W.arena + ops_loc + (m + n);
		uint32_t n_out = 0, cur_op = 0xf, cur_cnt = 0;
		uint32_t i = n, j = m, st = 0;		// st: 0 = M, 1 = V (up), 2 = H (left)
		// A load per cell would make the walk a chain of some hundred dependent memory round trips.  The path moves to the same
		// or the previous line and one or two steps back per cell, so a window of 2 lanes (8 lines) x 8 steps (two 16-byte loads)
		// serves the next few cells from registers.
		typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
		u64x2 win_a{0, 0}, win_b{0, 0};		// lanes win_lane - 1, win_lane of row win_row and of row win_row - 1
		uint32_t win_row = 0xFFFFFFF0u, win_lane = 0;
		while (i | j)
		{
			bool const inner = (0 != i) && (0 != j);
			uint32_t const line = (ref_lanes ? j : i) - 1, p = (ref_lanes ? i : j) - 1;		// (meaningless on the matrix' edge, where they are not used)
			uint32_t const ln = line >> 2, t = p + ln, row = t >> 2;
			if (inner && !((row == win_row || row + 1 == win_row) && (ln == win_lane || ln + 1 == win_lane)))
			{
				win_row = row; win_lane = ln;
				uint64_t const *q = trace + (uint64_t) row * 64 + ln;		// the trace buffer has room in front of its first word
				__builtin_memcpy(&win_a, q - 1, 16);
				if (row) __builtin_memcpy(&win_b, q - 65, 16);
			}
			u64x2 const wv = (row == win_row) ? win_a : win_b;
			uint64_t const word = (ln == win_lane) ? wv.y : wv.x;
			uint32_t const edge = (0 == i) ? (3u | ((1 == j) ? 8u : 0u)) : (2u | ((1 == i) ? 4u : 0u));	// first row: left, opened at column 1; first column: up
			uint32_t const nibble = inner ? ((uint32_t) (word >> (16 * (t & 3) + 4 * (line & 3))) & 0xfu) : edge;
			uint32_t const state = nibble & 3;
			uint32_t const dir = (0 == st) ? ((state < 2) ? 0u : (state - 1)) : st;		// 0 diagonal, 1 up, 2 left
			uint32_t const op = (0 == dir) ? ((state & 1) ? OP_X : OP_EQ) : ((1 == dir) ? OP_I : OP_D);		// align.hh:99-106
			bool const opened = 0 != (nibble & ((1 == dir) ? 4u : 8u));
			st = (0 == dir || opened) ? 0u : dir;
			i -= (2 != dir) ? 1u : 0u;
			j -= (1 != dir) ? 1u : 0u;
			bool const same = (op == cur_op);
			if (!same && cur_cnt) { ++n_out; *(slot_end - n_out) = (cur_cnt << 4) | cur_op; }
			cur_cnt = same ? (cur_cnt + 1) : 1u;
			cur_op = op;
		}
		if (cur_cnt) { ++n_out; *(slot_end - n_out) = (cur_cnt << 4) | cur_op; }
		W.run_items[item_idx].n_out = n_out;
		W.run_items[item_idx].ops_loc = ops_loc + (m + n) - n_out;
	}
}


// ---- striped alignment, one run per wavefront (runs that do not fit the packed kernel) --------------------------------------
struct dp_memory
{
	double *col_m;		// M(i, 0), i = 0..n
	double *bnd_m;		// M(i0, j), j = 0..m   (row above the current stripe)
	double *bnd_v;		// V(i0, j)
	uint8_t *trace;		// n * m bytes
	uint8_t *ref_rank;	// m bytes: dna5 ranks of the reference slice
};

__device__ __forceinline__ uint64_t dp_bytes(uint32_t m, uint32_t n)
{
	return 8ull * (n + 1) + 16ull * (m + 1) + (uint64_t) m * n + m + 8;
}

__device__ __forceinline__ dp_memory dp_carve(uint8_t *base, uint32_t m, uint32_t n)
{
	dp_memory d;
	d.col_m = (double *) base;
	d.bnd_m = d.col_m + (n + 1);
	d.bnd_v = d.bnd_m + (m + 1);
	d.trace = (uint8_t *) (d.bnd_v + (m + 1));
	d.ref_rank = d.trace + (uint64_t) m * n;
	return d;
}

// The order of preference among equal candidates of a cell for tie-break policy p (pvc3_params::tiebreak_mode = 2 p + o).
__constant__ uint32_t const TIEBREAK_ORDER[6][3] = {{TR_DIAG, TR_UP, TR_LEFT}, {TR_DIAG, TR_LEFT, TR_UP}, {TR_UP, TR_DIAG, TR_LEFT}, {TR_UP, TR_LEFT, TR_DIAG}, {TR_LEFT, TR_DIAG, TR_UP}, {TR_LEFT, TR_UP, TR_DIAG}};

struct dp_scores
{
	double open, ext;			// pvc3_params gap costs
	double const *match;		// LDS copies of dev_params::s_match / s_mismatch
	double const *mismatch;
	uint32_t tiebreak;		// pvc3_params::tiebreak_mode
};

// Global affine alignment of ref[0, m) (horizontal) against query bases [q0, q0 + n) (vertical); writes the operations
// run-length, last to first, below `slot_end` and returns their number.  All control flow is wave-uniform.
__device__ __forceinline__ uint32_t align_global_striped(
	uint8_t const *ref, uint32_t m, uint8_t const *seq4, uint32_t q0, uint32_t n, uint8_t const *qual /* or nullptr */,
	dp_scores const &sc, dp_memory const mem, uint32_t *slot_end, uint32_t lane
)
{
	double const go = sc.open + sc.ext;	// seqan3's internal gap-open score = open + extension
	double const ge = sc.ext;
	double const neg_inf = -__builtin_inf();

	auto load_row = [&](uint32_t i0, uint32_t &q_rank, double &s_match, double &s_mismatch) {
		q_rank = 3; s_match = 0.0; s_mismatch = -1.0;			// nucleotide_scoring_scheme<> default: 0 / -1
		if (i0 + lane < n)
		{
			q_rank = dna5_rank_of_seq4(seq4_at(seq4, q0 + i0 + lane));
			if (qual)
			{
				uint32_t qq = qual[q0 + i0 + lane];
				qq = (qq > MAX_PHRED) ? MAX_PHRED : qq;			// phred42 saturates
				s_match = sc.match[qq];
				s_mismatch = sc.mismatch[qq];
			}
		}
	};
	uint32_t q_rank0; double s_match0, s_mismatch0;
	load_row(0, q_rank0, s_match0, s_mismatch0);

	for (uint32_t j = lane; j < m; j += 64)
		mem.ref_rank[j] = (uint8_t) dna5_rank_of_char(ref[j]);

	// First column and first row (initialise_first_column_cell / initialise_first_row_cell).  The recurrence
	//     x_1 = go,  x_k = max(x_{k-1} + go [open], x_{k-1} + ge [extension, kept on ties])
	// has the closed forms k * (open + ext) when the opening score is positive and open + k * ext otherwise; all values are
	// small integers, so the products are exactly the iterated sums of the oracle.
	{
		double const open = sc.open;
		for (uint32_t k = lane; k <= (m > n ? m : n); k += 64)
		{
			double const x = (0 == k) ? 0.0 : ((open > 0.0) ? (double) k * go : open + (double) k * ge);
			if (k <= n) mem.col_m[k] = x;
			if (k <= m) { mem.bnd_m[k] = x; mem.bnd_v[k] = neg_inf; }
		}
	}
	wave_lds_fence();

	for (uint32_t i0 = 0; i0 < n; i0 += 64)
	{
		uint32_t const nr = (n - i0 < 64) ? (n - i0) : 64;		// rows in this stripe
		bool const row_ok = lane < nr;
		uint32_t const i = i0 + lane + 1;						// this lane's row (1-based)
		uint32_t q_rank = q_rank0;
		double s_match = s_match0, s_mismatch = s_mismatch0;
		if (i0) load_row(i0, q_rank, s_match, s_mismatch);
		double m_left = 0.0, h_left = neg_inf, diag = 0.0;
		if (row_ok)
		{
			m_left = mem.col_m[i];
			diag = mem.col_m[i - 1];
		}
		double cur_m = 0.0, cur_v = neg_inf;
		uint32_t const steps = m + nr - 1;
		for (uint32_t t = 0; t < steps; ++t)
		{
			double up_m = shfl_up_f64(cur_m);
			double up_v = shfl_up_f64(cur_v);
			uint32_t const j = t - lane + 1;					// column handled by this lane in this step
			bool const active = row_ok && (t >= lane) && (j <= m);
			if (active)
			{
				if (0 == lane) { up_m = mem.bnd_m[j]; up_v = mem.bnd_v[j]; }
				uint32_t const r_rank = mem.ref_rank[j - 1];
				double const s = (r_rank == q_rank) ? s_match : s_mismatch;
				uint32_t tr = 0;
				double const v_open = up_m + go, v_ext = up_v + ge;
				double const h_open = m_left + go, h_ext = h_left + ge;
				double v, h, best = diag + s;
				uint32_t dir = TR_DIAG;
				if (0 == sc.tiebreak)
				{
					if (v_ext < v_open) { v = v_open; tr |= TR_V_OPEN; } else v = v_ext;
					if (h_ext < h_open) { h = h_open; tr |= TR_H_OPEN; } else h = h_ext;
					if (best < v) { best = v; dir = TR_UP; }
					if (best < h) { best = h; dir = TR_LEFT; }
				}
				else
				{
					// the other tie-break policies (uniform over the launch): the same values, another path among the co-optimal ones
					bool const open_on_tie = sc.tiebreak & 1;
					if (open_on_tie ? !(v_open < v_ext) : (v_ext < v_open)) { v = v_open; tr |= TR_V_OPEN; } else v = v_ext;
					if (open_on_tie ? !(h_open < h_ext) : (h_ext < h_open)) { h = h_open; tr |= TR_H_OPEN; } else h = h_ext;
					uint32_t const o0 = TIEBREAK_ORDER[sc.tiebreak >> 1][0], o1 = TIEBREAK_ORDER[sc.tiebreak >> 1][1], o2 = TIEBREAK_ORDER[sc.tiebreak >> 1][2];
					double const dg = best;
					auto cand = [&](uint32_t d) { return (TR_DIAG == d) ? dg : ((TR_UP == d) ? v : h); };
					best = cand(o0); dir = o0;
					if (best < cand(o1)) { best = cand(o1); dir = o1; }
					if (best < cand(o2)) { best = cand(o2); dir = o2; }
				}
				mem.trace[(uint64_t) (i - 1) * m + (j - 1)] = (uint8_t) (tr | dir | ((r_rank == q_rank) ? TR_MATCH : 0));
				m_left = best; h_left = h; cur_m = best; cur_v = v; diag = up_m;
				if (lane == nr - 1) { mem.bnd_m[j] = best; mem.bnd_v[j] = v; }
			}
		}
		wave_lds_fence();
	}

	// Traceback (trace_iterator_base): diagonal > up > left where the direction is decided; a gap continues until the
	// cell whose gap score was formed by opening.  Wave-uniform; the stores are uniform too.
	uint32_t n_out = 0, cur_op = 0xf, cur_cnt = 0;
	auto emit = [&](uint32_t op) {
		if (op == cur_op) { ++cur_cnt; return; }
		if (cur_cnt) { ++n_out; *(slot_end - n_out) = (cur_cnt << 4) | cur_op; }
		cur_op = op; cur_cnt = 1;
	};
	{
		uint32_t i = n, j = m;
		uint32_t st = 0;	// 0 = M, 1 = V (up), 2 = H (left)
		while (i || j)
		{
			uint32_t t;
			if (0 == i) t = TR_LEFT | ((1 == j) ? TR_H_OPEN : 0);		// first row: left, opened at column 1
			else if (0 == j) t = TR_UP | ((1 == i) ? TR_V_OPEN : 0);	// first column
			else t = mem.trace[(uint64_t) (i - 1) * m + (j - 1)];
			t = rfl(t);
			if (0 == st)
			{
				uint32_t const dir = t & TR_DIR_MASK;
				if (TR_DIAG == dir)
				{
					emit((t & TR_MATCH) ? OP_EQ : OP_X);		// align.hh:103-106
					--i; --j;
					continue;
				}
				st = (TR_UP == dir) ? 1 : 2;
			}
			if (1 == st) { emit(OP_I); if (t & TR_V_OPEN) st = 0; --i; }
			else { emit(OP_D); if (t & TR_H_OPEN) st = 0; --j; }
		}
	}
	if (cur_cnt) { ++n_out; *(slot_end - n_out) = (cur_cnt << 4) | cur_op; }
	return n_out;
}

__global__ __launch_bounds__(64, 8) void k_realign_big(dev_batch_work const W, dev_params const *__restrict__ prm, uint32_t lds_budget)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[REALIGN_LDS_BYTES];
	__shared__ double lds_scores[2 * (MAX_PHRED + 1)];
	uint32_t const lane = lane_id();
	if (lane <= MAX_PHRED) { lds_scores[lane] = prm->s_match[lane]; lds_scores[MAX_PHRED + 1 + lane] = prm->s_mismatch[lane]; }
	dp_scores sc;
	sc.open = (double) prm->gap_open; sc.ext = (double) prm->gap_ext;
	sc.match = lds_scores; sc.mismatch = lds_scores + (MAX_PHRED + 1);
	sc.tiebreak = (uint32_t) prm->tiebreak_mode;
	wave_lds_fence();
	uint32_t const count = W.counters->dp_count[DP_CLASS_BIG];
	uint32_t const *list = W.dp_list + (uint64_t) DP_CLASS_BIG * W.runs_cap;
	for (uint32_t task = blockIdx.x; task < count; task += gridDim.x)
	{
		uint32_t const item_idx = rfl(list[task]);
		run_work const it = W.run_items[item_idx];
		uint32_t const m = rfl(it.m), nq = rfl(it.n), q_loc = rfl(it.query_loc);
		uint64_t const ops_loc = rfl64(it.ops_loc);
		uint8_t const *text = rfl_ptr(it.text), *seq4 = rfl_ptr(it.seq4), *qual = rfl_ptr(it.qual);
		uint32_t *slot_end = W.arena + ops_loc + (m + nq);
		uint64_t const need = dp_bytes(m, nq);
		uint32_t n_out;
		wave_lds_fence();
		if (need <= lds_budget)
		{
			// Separate call site: after inlining every access is provably to LDS (ds_read / ds_write instead of flat_*).
			n_out = align_global_striped(text, m, seq4, q_loc, nq, qual, sc, dp_carve(lds, m, nq), slot_end, lane);
		}
		else
		{
			if (need > W.scratch_per_wave)
			{
				atomicMax(&W.counters->scratch_needed, (unsigned int) ((need > 0xFFFFFFFFull) ? 0xFFFFFFFFu : need));	// idempotent
				W.counters->scratch_overflow = 1;		// the host grows the scratch and repeats the batch
				continue;
			}
			n_out = align_global_striped(text, m, seq4, q_loc, nq, qual, sc, dp_carve(W.scratch + (uint64_t) blockIdx.x * W.scratch_per_wave, m, nq), slot_end, lane);
		}
		// wave-uniform stores
		W.run_items[item_idx].n_out = n_out;
		W.run_items[item_idx].ops_loc = ops_loc + (m + nq) - n_out;
	}
}


// ---- splice -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_realign_splice(dev_batch_in const in, dev_batch_work const W, uint32_t n_regions)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	uint64_t const n_groups = (in.n + 63) / 64;
	uint32_t const total = W.run_prefix[n_regions];
	__shared__ uint32_t lds_ops[4][STAGED_OPS * 64];
	uint32_t *my_ops = lds_ops[rfl(threadIdx.x >> 6)];
	for (uint32_t base = 64 * wave; base < total; base += 64 * n_waves)
	{
		bool const mine = base + lane < total;
		uint32_t const rec = mine ? listed_record(W, n_regions, n_groups, base + lane) : 0;
		uint32_t const n_runs = mine ? W.run_n[rec] : 0;
		bool const work = mine && n_runs && !(W.rec_flags[rec] & REC_STATUS_MASK);		// not rejected by k_realign_scan, not left out by an arena overflow
		uint32_t const *ops_global = W.arena + (work ? W.cig_loc[rec] : 0);
		uint32_t const n_ops = work ? W.cig_n[rec] : 0;
		wave_lds_fence();
		op_source const ops = stage_ops(ops_global, n_ops, my_ops, lane);
		wave_lds_fence();
		if (!work)
			continue;
		run_work const *items = W.run_items + W.run_loc[rec];
		uint64_t const out_loc = items[0].splice_loc;
		uint32_t *out = W.arena + out_loc;
		// cigar_buffer (cigar.cc:54-95); the rewritten CIGAR has no zero counts and neither have the alignments.
		uint32_t n_out = 0, cur_op = 0xf, cur_cnt = 0;
		auto push = [&](uint32_t c) {
			uint32_t const op = c & 0xf, cnt = c >> 4;
			if (op == cur_op) { cur_cnt += cnt; return; }
			if (cur_cnt) out[n_out++] = (cur_cnt << 4) | cur_op;
			cur_op = op; cur_cnt = cnt;
		};
		uint32_t k = 0;
		uint32_t next_first = items[0].op_first;
		for (uint32_t x = 0; x < n_ops; )
		{
			if (x == next_first)		// alignment_projector.cc:140-141: the run's operations are replaced by its alignment
			{
				run_work const it = items[k];
				uint32_t const *src = W.arena + it.ops_loc;
				typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
				for (uint32_t y = 0; y < it.n_out; y += 4)		// four operations per load (the arena is padded)
				{
					u32x4 v;
					__builtin_memcpy(&v, src + y, 16);
					push(v.x);
					if (y + 1 < it.n_out) push(v.y);
					if (y + 2 < it.n_out) push(v.z);
					if (y + 3 < it.n_out) push(v.w);
				}
				x = it.op_end;
				++k;
				next_first = (k < n_runs) ? items[k].op_first : 0xFFFFFFFFu;
				continue;
			}
			push(ops[x]);				// :61-65, :145-148
			++x;
		}
		if (cur_cnt) out[n_out++] = (cur_cnt << 4) | cur_op;
		W.cig_loc[rec] = out_loc;
		W.cig_n[rec] = n_out;
	}
}

}	// anonymous namespace

uint32_t realign_lds_bytes() { return REALIGN_LDS_BYTES; }

int launch_realign(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, dev_params const *params, uint32_t n_waves, uint32_t big_waves, uint32_t lds_budget, uint32_t packed, void *stream)
{
	if (0 == in.n) return 0;
	(void) n_entries;
	if (lds_budget > REALIGN_LDS_BYTES) lds_budget = REALIGN_LDS_BYTES;
	uint32_t const n_regions = project_wave_count(in.n);		// one work list per wave of the projection kernels
	hipStream_t const s = (hipStream_t) stream;
	uint32_t const list_blocks = (n_regions + 3) / 4 < 2048 ? (n_regions + 3) / 4 : 2048;
	hipLaunchKernelGGL(k_run_prefix, dim3(1), dim3(1024), 0, s, n_regions, work);
	if (work.host_qual)
		hipLaunchKernelGGL(k_gather_qual, dim3(list_blocks), dim3(256), 0, s, in, work, n_regions);
	hipLaunchKernelGGL(k_realign_scan, dim3(list_blocks), dim3(256), 0, s, entries, in, work, n_regions, packed);
	if (n_waves < 1) n_waves = 1;
	hipLaunchKernelGGL(k_realign_sort, dim3(512, DP_CLASS_BIG), dim3(SORT_THREADS), 0, s, work);
	hipLaunchKernelGGL(k_realign_dp, dim3(n_waves), dim3(64), 0, s, work, params);
	hipLaunchKernelGGL(k_realign_traceback, dim3(1024, DP_CLASS_BIG), dim3(256), 0, s, work);
	if (big_waves < 1) big_waves = 1;
	hipLaunchKernelGGL(k_realign_big, dim3(big_waves), dim3(64), 0, s, work, params, lds_budget);
	hipLaunchKernelGGL(k_realign_splice, dim3(list_blocks), dim3(256), 0, s, in, work, n_regions);
	return (int) hipGetLastError();
}

}
