// project_probes.hip — k_project_fast_probes: lane = record.  First kernel of the hot path for an index WITHOUT position
// windows (dev_entry::pwin; pvc3_context_upload_index leaves them out when they do not fit the device or are switched off);
// with them k_project_fast (project_fast.hip) runs instead.  Same results, same work lists.
//
// For 64 records at a time per wavefront (coalesced SoA descriptor loads):
//   1. validates the CIGAR and the positions (per-record status, same rules as the oracle's batch driver);
//   2. aligned_position + project_aligned_position (include/panvc3/rewrite_cigar.hh:51-56): one select sample, then four
//      probes of 24 bytes (wave.hpp: word 0 + one pair of a block) into the paired copy of the two rows give the start
//      column, the projected position and the ranks of both rows at the end of the span;
//   3. the mate's project_position (include/panvc3/msa_index.hh:253-258, project_alignments.cc:897-901);
//   4. FAST PATH.  A record is "simple" when its CIGAR is [S|H] (M|=|X)+ [S|H] with non-zero counts, it consumes at most
//      256 reference bases and neither row has a gap among the columns it spans (the ranks of both rows advance by exactly
//      that many bases).  Then rewrite_cigar (rewrite_cigar.hh:80-118) degenerates to a byte compare of the query against the
//      destination text (4-bit codes; from the windowed copy when the read fits one of its lines): n_words lanes per record
//      build the mismatch words, then the lane emits the = / X runs itself (cigar_buffer semantics, libpanvc3/cigar.cc:54-95).
//      Such a record has no indel run, so it is finished.
//   5. Every other valid record goes to the wave's work list for k_project_mid (REC_SLOW; start column, projected position
//      and reference-consuming length are left in the work arrays).
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"
#include "compare.hpp"

namespace pvc3 {

namespace {

constexpr uint32_t FAST_MAX_BASES = 256;
constexpr uint32_t FAST_WORDS = FAST_MAX_BASES / 32;
constexpr uint32_t LDS_ENTRIES = 256;
constexpr uint32_t STAGE_TOKENS = 64 * FAST_WORDS;		// uint32 in a wave's LDS area (mismatch words, then output operations)
#ifndef PVC3_FAST_PASSES
#define PVC3_FAST_PASSES 4
#endif
constexpr uint32_t FAST_PASSES = PVC3_FAST_PASSES;	// passes of the cooperative compare whose loads are issued together
#ifndef PVC3_FAST_MIN_WAVES
#define PVC3_FAST_MIN_WAVES 4
#endif

// Mismatch bits of 32 consecutive query nibbles (16 bytes of SEQ) against 32 text nibbles (t, nibble-linear).
__device__ __forceinline__ uint32_t mismatch32(u64x2 const &qq, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3)
{
	uint64_t const q0 = nibble_linear(qq.x), q1 = nibble_linear(qq.y);
	return mismatch8((uint32_t) q0, t0) | (mismatch8((uint32_t) (q0 >> 32), t1) << 8)
		| (mismatch8((uint32_t) q1, t2) << 16) | (mismatch8((uint32_t) (q1 >> 32), t3) << 24);
}

// Orders this wave's LDS accesses before its later ones.  The LDS queue of a wave is in-order, so only the compiler has to
// be stopped: the staging area is accessed through several types (16-byte pieces, mismatch words, operations), type-based
// alias analysis would let the machine scheduler move a load past a store of another type, and a wavefront-scope fence
// emits no instruction that could stop it.  The asm statement is a barrier for both the optimiser and the scheduler.
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
	for (int off = 32; off; off >>= 1)
	{
		uint32_t const o = (uint32_t) __shfl_xor((int) v, off);
		v = (o > v) ? o : v;
	}
	return rfl(v);
}

// Diagnostic build (tools/fast_phase_cycles.py): s_memtime stamps at the phase boundaries of a group, every stamp behind a
// full wait so that a phase is charged with the latency of what it loads.  Sums per wave at the end of the trace buffer.
#ifdef PVC3_FAST_STAMPS
#define FAST_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); uint64_t const t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
	stamp_acc[k] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define FAST_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256, PVC3_FAST_MIN_WAVES) void k_project_fast_probes(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W, uint32_t use_fast_path)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	// Contiguous group range per wave: sequential descriptor reads and a sequentially filled arena chunk.
	uint64_t const n_groups = (in.n + 63) / 64;
	uint64_t const g_begin = n_groups * wave / n_waves;
	uint64_t const g_end = n_groups * (wave + 1) / n_waves;
	arena_alloc arena{0, 0};
	uint32_t acc_failed = 0, acc_fast = 0;		// per lane
	uint32_t n_slow = 0;						// wave-uniform: entries in this wave's work list for k_project_mid
	uint32_t *slow_list = W.slow_list + 64 * g_begin;
	slow_item *slow_items = W.slow_items + 64 * g_begin;
	bool items_mode = true;					// wave-uniform: decided by the wave's first group (see slow_item)

	// The entry table is read by every lane of every group: keep its head in LDS.
	__shared__ dev_entry lds_entries[LDS_ENTRIES];
	for (uint32_t i = threadIdx.x; i < LDS_ENTRIES && i < n_entries; i += blockDim.x)
		lds_entries[i] = entries[i];
	__syncthreads();
	// Per-wave staging for the cooperative compare: record descriptors out, mismatch words back.
	__shared__ u64x2 lds_desc[4][64];
	__shared__ uint32_t lds_mask[4][STAGE_TOKENS];
	uint32_t const wave_in_block = rfl(threadIdx.x >> 6);
	u64x2 *my_desc = lds_desc[wave_in_block];
	uint32_t *my_mask = lds_mask[wave_in_block];
	auto get_entry = [&](uint32_t id) -> dev_entry { return (id < LDS_ENTRIES) ? lds_entries[id] : entries[id]; };
	// A batch whose offsets are malformed (k_validate_batch; a pipelined batch is checked while its kernels are already
	// queued) is not read at all: every record is rejected and the host reports PVC3_E_INVALID.
	bool const batch_bad = 0xFFFFFFFFu != rfl(*W.first_bad);
	uint64_t acc_bitvec = 0, acc_ref = 0, acc_bytes = 0;

#ifdef PVC3_FAST_STAMPS
	uint64_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint64_t stamp_prev = __builtin_amdgcn_s_memtime();
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
	for (uint64_t g = g_begin; g < g_end; ++g)
	{
		uint64_t const rec = g * 64 + lane;
		bool const valid = rec < in.n;
		uint64_t const rl = valid ? rec : (in.n - 1);		// clamped index: every lane issues the same loads

		// ---- descriptors (coalesced) -------------------------------------------------------------------------------------------
		uint32_t const src_pos = in.src_pos[rl];
		uint32_t const src_e = in.src_entry[rl];
		uint32_t const dst_e = in.dst_entry[rl];
		uint32_t const fl = in.flags[rl];
		uint64_t const cig_begin = batch_bad ? 0 : in.cigar_off[rl];
		uint32_t const n_cigar = batch_bad ? 0 : (uint32_t) (in.cigar_off[rl + 1] - cig_begin);
		uint64_t const seq_begin = in.seq_off[rl];
		uint32_t const l_seq = in.l_seq[rl];
		bool const has_mate = valid && in.mate_pos && (fl & PVC3_FLAG_HAS_MATE);
		uint32_t const mate_se = has_mate ? in.mate_src_entry[rl] : 0;
		uint32_t const mate_de = has_mate ? in.mate_dst_entry[rl] : 0;
		uint32_t const mate_pos = has_mate ? in.mate_pos[rl] : 0;

		FAST_STAMP(0);		// descriptors
		bool const e_ok = (src_e < n_entries) && (dst_e < n_entries);
		dev_entry const S = get_entry(e_ok ? src_e : 0);
		dev_entry const D = get_entry(e_ok ? dst_e : 0);
		row_pair const rows = rows_of(S, D, dst_e);
		uint32_t const *s_sel = S.sel;
		uint32_t const s_zeros = S.n_zeros, s_cols = S.n_cols;
		uint8_t const *d_text4 = D.text4, *d_text4w = D.text4w;
		uint32_t const d_text_len = D.text_len, d_cols = D.n_cols;

		// ---- CIGAR: validation + shape automaton of the fast path (first four operations from one 16-byte load) ----------------
		uint32_t const *ops = in.cigar + cig_begin;
		typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
		u32x4 pre4;
		__builtin_memcpy(&pre4, ops, 16);					// the CIGAR buffer is padded by the host side
		uint32_t const pre[4] = {pre4.x, pre4.y, pre4.z, pre4.w};
		uint32_t ref_consumed = 0, query_consumed = 0, shape = 0, lead = 0, trail = 0;
		bool bad_op = false;
		// (Plain selects: see k_project_fast — an `if` that assigns `lead` in one arm and `trail` in another puts both into scratch.)
		auto step = [&](uint32_t c) {
			uint32_t const op = c & 0xf, n = c >> 4;
			bool const m_type = (OP_M == op || OP_EQ == op || OP_X == op);
			bool const clip = (OP_S == op || OP_H == op);
			bad_op = bad_op || (op > OP_X);					// any bad operation rejects the record, the counts are then unused
			ref_consumed += (m_type || OP_D == op || OP_N == op) ? n : 0;
			query_consumed += (m_type || OP_I == op || OP_S == op) ? n : 0;
			// 0 start, 1 after the leading clip, 2 inside the M-type span, 3 after the trailing clip, 4 not simple
			bool const is_m = (0 != n) && m_type && shape <= 2;
			bool const is_lead = (0 != n) && !is_m && clip && 0 == shape;
			bool const is_trail = (0 != n) && !is_m && clip && 2 == shape;
			lead = is_lead ? c : lead;
			trail = is_trail ? c : trail;
			shape = is_m ? 2u : is_lead ? 1u : is_trail ? 3u : 4u;
		};
#pragma unroll
		for (uint32_t i = 0; i < 4; ++i)
			if (i < n_cigar) step(pre[i]);
		for (uint32_t i = 4; i < n_cigar; ++i)
			step(ops[i]);

		FAST_STAMP(1);		// entries, CIGAR
		uint32_t status = PVC3_ST_OK;
		if (batch_bad) status = PVC3_ST_QUERY_RANGE;
		else if (!e_ok) status = PVC3_ST_SRC_RANGE;
		else if (bad_op) status = PVC3_ST_CIGAR_OP;
		else if (s_cols != d_cols || (uint64_t) src_pos + (ref_consumed ? ref_consumed : 1u) > s_zeros) status = PVC3_ST_SRC_RANGE;
		else if (query_consumed > l_seq) status = PVC3_ST_QUERY_RANGE;
		bool const ok = valid && (PVC3_ST_OK == status);

		// ---- start column and projected position (rewrite_cigar.hh:51-56) --------------------------------------------------------
		uint32_t col = 0, r0 = 0;
		bool fast = false;
		uint32_t const lead_q = (OP_S == (lead & 0xf)) ? (lead >> 4) : 0;	// query bases before the M-type span
		uint32_t const phase = lead_q & 1;
		if (ok)
		{
			// One select sample, then four probes (word 0 + one 16-byte pair of a block, wave.hpp) at the sample's guess for the
			// start column and for the end column: both rows, all issued together.  A probe is fetched again when the guess
			// did not hold (a gap column between the sample and the record).
			select_guess_t const sg = select_guess(s_sel, src_pos, ref_consumed);
			uint32_t const guess = sg.col;
			FAST_STAMP(2);		// select sample
			uint32_t const step = rows.src.step;
			col_pos const g0 = locate(step, guess), g1 = locate_in(step, (sg.end < s_cols) ? sg.end : s_cols, g0.blk);		// (an interpolated guess may pass the last column)
			dir_probe const ps0 = dir_load(rows.src, g0.blk, pair_of(g0));
			dir_probe const pd0 = dir_load(rows.dst, g0.blk, pair_of(g0));
			dir_probe const ps1 = dir_load(rows.src, g1.blk, pair_of(g1));
			dir_probe const pd1 = dir_load(rows.dst, g1.blk, pair_of(g1));
			col = dir_select0(rows.src, src_pos, ps0);
			// (If the paired copy's blocks overlap, PAIRED_STEP < 448, the end of the span is looked up in the block of its start.)
			col_pos const p0 = locate_in(step, col, g0.blk);
			col_pos const p1 = locate_in(step, col + ref_consumed, p0.blk);
			r0 = dir_rank(rows.dst, pd0, p0);
			// No gap column in either row over the span <=> both ranks advance by exactly ref_consumed.
			bool const gapless = (dir_rank(rows.src, ps1, p1) == src_pos + ref_consumed) && (dir_rank(rows.dst, pd1, p1) == r0 + ref_consumed);
			bool const simple = (2 == shape || 3 == shape);
			fast = use_fast_path && simple && gapless && (ref_consumed + phase < FAST_MAX_BASES) && d_text4 && ((uint64_t) r0 + ref_consumed <= d_text_len);
		}

		FAST_STAMP(3);		// probes, select, ranks
		// ---- mate (project_alignments.cc:897-901) -----------------------------------------------------------------------------------
		// project_position is a pure function of (entries, position).  When the mate is the neighbouring record of the batch
		// (pairs stored back to back) its projected position has just been computed by the neighbouring lane.
		uint32_t mate = PVC3_NO_MATE;
		{
			uint32_t const n_se = (uint32_t) __shfl_xor((int) src_e, 1), n_de = (uint32_t) __shfl_xor((int) dst_e, 1);
			uint32_t const n_pos = (uint32_t) __shfl_xor((int) src_pos, 1), n_r0 = (uint32_t) __shfl_xor((int) r0, 1);
			bool const n_ok = 0 != __shfl_xor((int) ok, 1);
			bool const from_neighbour = has_mate && n_ok && (mate_se == n_se) && (mate_de == n_de) && (mate_pos == n_pos);
			if (from_neighbour)
				mate = n_r0;
			else if (has_mate && mate_se < n_entries && mate_de < n_entries)
			{
				mate = thread_project_position(get_entry(mate_se), get_entry(mate_de), mate_de, mate_pos);
				acc_bytes += 4 + 64 + 64;		// select sample, one source block, one destination block
			}
		}

		FAST_STAMP(4);		// mate
		// ---- fast path ------------------------------------------------------------------------------------------------------------------
		uint32_t f_n_out = 0;
		uint64_t f_loc = 0;
#ifdef PVC3_ABL_NO_COMPARE
		if (false)
#else
		if (ballot(fast))
#endif
		{
			uint32_t const R = fast ? ref_consumed : 0;
			// Work in query-nibble positions u = t + phase counted from a byte boundary of SEQ, so that 32 positions are exactly
			// 16 bytes of SEQ and 32 bytes of text; the mask is shifted back by `phase` afterwards.
			uint32_t const U = fast ? (R + phase) : 0;
			uint8_t const *qp = in.seq4 + seq_begin + (lead_q >> 1);
			// Text nibble that lines up with query position u = 0 (base r0 - phase; the packed text has TEXT4_FRONT nibbles in front).
			uint64_t const tn = (uint64_t) r0 + TEXT4_FRONT - phase;
			// One more position than needed, so that the lane after a record's last full word exists (see below).
			uint32_t const n_words = (wave_max(U) + 32) / 32;		// 1 .. FAST_WORDS
			// The compare is done by n_words lanes per record, one 32-position word each: a wave instruction then touches the
			// lines of 64 / n_words records instead of one line per lane, which the memory pipeline serves 2-3x faster
			// (tools/calib/random_access.hip).  Descriptors go to the lanes through LDS, the masks come back the same way.
			{
				u64x2 d;
				d.x = (uint64_t) qp | ((uint64_t) U << 48);
				// From the windowed copy when the run fits a line of it (reads of up to 150 bases): one 128-byte line of text.
				uint32_t const tn32 = (uint32_t) tn;				// text_len + TEXT4_FRONT < 2^32 (pvc3_index_set_reference)
				uint32_t const line = tn32 / TEXT_WINDOW_STEP, in_line = tn32 - line * TEXT_WINDOW_STEP;
				uint8_t const *tp = (d_text4w && U <= TEXT_WINDOW_MAX) ? (d_text4w + 128 * (uint64_t) line + (in_line >> 1)) : (d_text4 + (tn >> 1));
				d.y = (uint64_t) tp | ((tn & 1) << 63);
				my_desc[lane] = d;
			}
			wave_lds_sync();
			{
				uint32_t const inv = (65536 + n_words - 1) / n_words;
				uint32_t const per = (64 * inv) >> 16;			// records per pass
				uint32_t const sr = (lane * inv) >> 16;				// lane / n_words
				uint32_t const w = lane - sr * n_words;
				typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
				auto fetch = [&](uint32_t rr, bool act, u64x2 &qq, u32x4 &tt, uint32_t &left, uint32_t &odd4) {
					u64x2 const dd = my_desc[act ? rr : lane];
					uint32_t const Ur = (uint32_t) (dd.x >> 48);
					left = (act && Ur > 32 * w) ? (Ur - 32 * w) : 0;
					odd4 = (uint32_t) (dd.y >> 63) * 4;
					// When the text starts at an odd nibble a word needs the first nibble of the next 16 bytes: the lane at
					// position 32 w == Ur loads them although it has nothing to compare.
					if (act && Ur && 32 * w <= Ur)		// SEQ loads stay inside the record (+ at most 31 positions of padding)
					{
						uint8_t const *q = (uint8_t const *) (dd.x & UINT64_C(0xFFFFFFFFFFFF)) + 16 * w;
						uint8_t const *t = (uint8_t const *) (dd.y & UINT64_C(0xFFFFFFFFFFFF)) + 16 * w;
						if (left) qq = load_u128_unaligned(q);
						__builtin_memcpy(&tt, t, 16);
					}
				};
				auto finish = [&](uint32_t rr, bool act, u64x2 const &qq, u32x4 const &tt, uint32_t left, uint32_t odd4) {
					// Shift the 128-bit text window right by one nibble if needed; the top nibble comes from the next lane.
					uint32_t const next = (uint32_t) __shfl_down((int) tt.x, 1);
					uint32_t m = 0;
					if (left)
					{
						uint32_t const t0 = __builtin_amdgcn_alignbit(tt.y, tt.x, odd4), t1 = __builtin_amdgcn_alignbit(tt.z, tt.y, odd4);
						uint32_t const t2 = __builtin_amdgcn_alignbit(tt.w, tt.z, odd4), t3 = __builtin_amdgcn_alignbit(next, tt.w, odd4);
						m = mismatch32(qq, t0, t1, t2, t3) & (uint32_t) low_mask(left < 32 ? left : 32);
					}
					if (act) my_mask[rr * FAST_WORDS + w] = m;
				};
				// FAST_PASSES passes at a time: all their loads are in flight before the first word is compared (a pass per round
				// trip made the compare three dependent round trips for 150-base reads, a quarter of the kernel's time).
				for (uint32_t base = 0; base < 64; base += FAST_PASSES * per)
				{
					u64x2 qv[FAST_PASSES];
					u32x4 tv[FAST_PASSES];
					uint32_t left_v[FAST_PASSES], odd_v[FAST_PASSES];
#pragma unroll
					for (uint32_t k = 0; k < FAST_PASSES; ++k)
					{
						uint32_t const rr = base + k * per + sr;
						qv[k] = u64x2{0, 0}; tv[k] = u32x4{0, 0, 0, 0};
						fetch(rr, (sr < per) & (rr < 64), qv[k], tv[k], left_v[k], odd_v[k]);
					}
#pragma unroll
					for (uint32_t k = 0; k < FAST_PASSES; ++k)
					{
						uint32_t const rr = base + k * per + sr;
						finish(rr, (sr < per) & (rr < 64), qv[k], tv[k], left_v[k], odd_v[k]);
					}
				}
			}
			wave_lds_sync();
			FAST_STAMP(5);		// compare
			uint32_t mu[FAST_WORDS + 1];
			{
				typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
				u32x4 const *mp = (u32x4 const *) (my_mask + lane * FAST_WORDS);
				u32x4 const a = mp[0], b = mp[1];
				uint32_t const all[FAST_WORDS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
				for (uint32_t w = 0; w < FAST_WORDS; ++w)
					mu[w] = (fast && w < n_words) ? all[w] : 0;
			}
			mu[FAST_WORDS] = 0;
			// Back to base positions: drop the `phase` leading positions.
			uint32_t mm[FAST_WORDS];
#pragma unroll
			for (uint32_t w = 0; w < FAST_WORDS; ++w)
				mm[w] = __builtin_amdgcn_alignbit(mu[w + 1], mu[w], phase);
			// Number of runs = 1 + number of positions whose bit differs from the previous one.
			uint32_t runs = fast ? 1 : 0;
			uint32_t prev_top = mm[0] & 1;		// makes position 0 a non-transition
#pragma unroll
			for (uint32_t w = 0; w < FAST_WORDS; ++w)
			{
				if (w < n_words)
				{
					uint32_t const left = (R > 32 * w) ? (R - 32 * w) : 0;
					uint32_t const vmask = (uint32_t) low_mask(left < 32 ? left : 32);
					uint32_t const tr = (mm[w] ^ ((mm[w] << 1) | prev_top)) & vmask;
					runs += (uint32_t) __builtin_popcount(tr);
					prev_top = mm[w] >> 31;
				}
			}
			f_n_out = fast ? (runs + (lead ? 1 : 0) + (trail ? 1 : 0)) : 0;
			uint32_t const scan = wave_inclusive_scan(f_n_out, lane);
			uint32_t const total = readlane(scan, 63);
			uint64_t base_loc = 0;
			bool const have = arena_take(arena, total, W, lane, base_loc);
			if (!have) f_n_out = 0;		// arena overflow: the host grows it and repeats the batch
			f_loc = base_loc + (scan - f_n_out);
			// The group's operations are contiguous in the arena.  Each lane stores its own few; through LDS (then one coalesced
			// copy by the whole wave) when the group's operations fit there, directly otherwise.
			bool const staged = total <= STAGE_TOKENS;
			wave_lds_sync();		// the mismatch words in this LDS area have been read
			FAST_STAMP(6);		// run count, scan, arena
			auto emit = [&](auto put) {
				uint32_t k = 0;
				if (lead) put(k++, lead);
				uint32_t start = 0, type = mm[0] & 1;
				prev_top = type;
#pragma unroll
				for (uint32_t w = 0; w < FAST_WORDS; ++w)
				{
					if (w < n_words)
					{
						uint32_t const left = (R > 32 * w) ? (R - 32 * w) : 0;
						uint32_t const vmask = (uint32_t) low_mask(left < 32 ? left : 32);
						uint32_t tr = (mm[w] ^ ((mm[w] << 1) | prev_top)) & vmask;
						prev_top = mm[w] >> 31;
						while (tr)
						{
							uint32_t const pos = 32 * w + (uint32_t) __builtin_ctz(tr);
							put(k++, ((pos - start) << 4) | (type ? OP_X : OP_EQ));
							start = pos; type ^= 1;
							tr &= tr - 1;
						}
					}
				}
				put(k++, ((R - start) << 4) | (type ? OP_X : OP_EQ));
				if (trail) put(k++, trail);
			};
			if (staged)
			{
				uint32_t *dst = my_mask + (scan - f_n_out);
				if (have && fast) emit([&](uint32_t k, uint32_t v) { dst[k] = v; });
				wave_lds_sync();
				if (have)
					for (uint32_t j = lane; j < total; j += 64)
						W.arena[base_loc + j] = my_mask[j];
			}
			else if (have && fast)
			{
				uint32_t *dst = W.arena + f_loc;
				emit([&](uint32_t k, uint32_t v) { dst[k] = v; });
			}
			FAST_STAMP(7);		// emission
			if (have && fast)
			{
				acc_bitvec += 2 * ((R + 8) / 8);
				acc_ref += (R + 1) / 2;
				acc_fast += 1;
				// what the fast path itself must read and write for this record: SEQ, gap-bit windows, text, CIGAR out
				acc_bytes += (l_seq + 1) / 2 + 2 * ((R + 8) / 8) + (R + 1) / 2 + 4 * f_n_out;
			}
		}

		// ---- work list of the records left for k_project_mid --------------------------------------------------------------------------
		{
			uint64_t const m = ballot(valid && ok && !fast);
			if (g == g_begin) items_mode = popc64(m) <= 32;
			if ((m >> lane) & 1)
			{
				uint32_t const at = n_slow + mbcnt64(m);
				slow_list[at] = (uint32_t) rec;
			}
			if (items_mode && ((m >> lane) & 1))
			{
				uint32_t const at = n_slow + mbcnt64(m);
				slow_item it;
				it.rec = (uint32_t) rec; it.col0 = col; it.r0 = r0; it.src_e = src_e;
				it.dst_e = dst_e; it.n_cigar = n_cigar; it.l_seq = l_seq; it.skip = 0;
				it.cig_begin = cig_begin; it.seq_begin = seq_begin;
				slow_items[at] = it;
			}
			n_slow += popc64(m);
		}

		// ---- per-record outputs (coalesced) ----------------------------------------------------------------------------------------------
		if (valid)
		{
			bool const slow = ok && !fast;
			W.mate_dst_pos[rec] = mate;
			W.dst_pos[rec] = ok ? r0 : 0;		// a rejected record has empty outputs
			W.rec_flags[rec] = slow ? REC_SLOW : status;
			W.cig_n[rec] = f_n_out;
			W.cig_loc[rec] = slow ? ((uint64_t) col | ((uint64_t) ref_consumed << 32)) : f_loc;
			W.run_n[rec] = 0;
			acc_failed += (PVC3_ST_OK != status);
			// every record: 7 descriptor fields (28 + 4 flags), input CIGAR, select sample, destination rank word; 24 bytes of results
			acc_bytes += 32 + 4 * n_cigar + (ok ? 12 : 0) + 24 + (has_mate ? 12 : 0);
		}
		FAST_STAMP(8);		// work list, outputs
	}
#ifdef PVC3_FAST_STAMPS
	if (0 == lane)
	{
		unsigned long long *dst = (unsigned long long *) (W.trace + W.trace_cap) - 16 * (uint64_t) (wave + 1);
		for (uint32_t k = 0; k < 12; ++k) dst[k] = stamp_acc[k];
		dst[12] = g_end - g_begin;
	}
#endif

	W.slow_count[wave] = n_slow | (items_mode ? SLOW_ITEMS : 0u);		// wave-uniform stores
	W.run_count[wave] = 0;				// k_project_mid / k_project_general append the records with indel runs
	// Per-lane accumulators -> wave totals -> one atomic each.
	for (int off = 32; off; off >>= 1)
	{
		acc_bitvec += __shfl_xor(acc_bitvec, off);
		acc_ref += __shfl_xor(acc_ref, off);
		acc_failed += __shfl_xor(acc_failed, off);
		acc_fast += __shfl_xor(acc_fast, off);
		acc_bytes += __shfl_xor(acc_bytes, off);
	}
	if (0 == lane)
	{
		unsigned long long *ws = W.wave_stats + ((uint64_t) 0 * n_waves + wave) * WAVE_STATS;
		ws[WS_BITVEC] = acc_bitvec;
		ws[WS_REF] = acc_ref;
		ws[WS_FAILED] = acc_failed;
		ws[WS_DONE] = acc_fast;
		ws[WS_BYTES] = acc_bytes;
	}
}

}	// anonymous namespace

int launch_project_probes(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, uint32_t use_fast_path, void *stream)
{
	if (0 == in.n) return 0;
	// The grid is that of k_project_fast: the work lists are per wave.
	hipLaunchKernelGGL(k_project_fast_probes, dim3(project_wave_count(in.n) / 4), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work, use_fast_path);
	return (int) hipGetLastError();
}

}
