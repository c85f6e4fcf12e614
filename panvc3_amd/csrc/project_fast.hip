// project_fast.hip — k_project_fast: lane = record.  First kernel of the hot path, over an index with position windows.
//
// A position window (dev_entry::pwin, pvc3_internal.hpp) holds, in the ONE 128- or 256-byte line that a record's source
// position addresses by itself, everything rewrite_cigar needs of the index for an alignment without gap columns: the
// start column and the projected position (aligned_position + project_aligned_position, include/panvc3/rewrite_cigar.hh:
// 51-56), the number of columns from there on that have no gap in either row, and the destination text of those columns.
// So for 64 records at a time per wavefront (coalesced SoA descriptor loads):
//   1. validates the CIGAR and the positions (per-record status, same rules as the oracle's batch driver);
//   2. requests the window's 16-byte header — and, at once, the compare's loads: SEQ and the window's text, whose addresses
//      depend on the descriptors only.  One memory round trip after the descriptors instead of the three of
//      k_project_fast_probes (select sample -> gap blocks -> text), one random line per record instead of three;
//   3. FAST PATH.  A record is "simple" when its CIGAR is [S|H] (M|=|X)+ [S|H] with non-zero counts and it consumes fewer
//      than 256 reference bases; it is finished here when the window is clean from its first column to the record's last
//      (no gap column in either row) and its text lies in the window's line.  rewrite_cigar (rewrite_cigar.hh:80-118) is
//      then a compare of the query against the destination text (4-bit codes): n_words lanes per record build the mismatch
//      words, the lane emits the = / X runs itself (cigar_buffer semantics, libpanvc3/cigar.cc:54-95).  Such a record has no
//      indel run, so it is finished;
//   4. the mate's project_position (include/panvc3/msa_index.hh:253-258, project_alignments.cc:897-901): the neighbouring
//      lane's result when the mate is the neighbouring record, a window of its own otherwise;
//   5. every other valid record goes to the wave's work list for k_project_mid with what the window told about its start:
//      the start column and projected position themselves, or the window's last clean column and the number of source
//      positions still to go (resolve_start, wave.hpp).
#include <hip/hip_runtime.h>

#include <cstdlib>

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
#define PVC3_FAST_PASSES 3
#endif
constexpr uint32_t FAST_PASSES = PVC3_FAST_PASSES;	// passes of the cooperative compare whose loads are issued together
#ifndef PVC3_FAST_MIN_WAVES
#define PVC3_FAST_MIN_WAVES 4
#endif

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;

// 16 bytes from a dword-aligned address in global memory.  Pointers that went through LDS or an integer have lost their
// address space; a flat load would also count against the LDS counter (lgkmcnt) and wait with the LDS traffic around it.
__device__ __forceinline__ u32x4 global_load16(void const *p)
{
	typedef uint32_t __attribute__((address_space(1))) const *gptr;
	gptr const g = (gptr) (uintptr_t) p;
	u32x4 v;
	v.x = g[0]; v.y = g[1]; v.z = g[2]; v.w = g[3];
	return v;
}

// What this kernel needs of an entry, 32 bytes: the head of the table lives in LDS in this form.
struct fast_entry
{
	uint8_t const *pwin;
	uint32_t n_zeros, n_cols;
	uint32_t partner;			// destination the windows were made for
	uint32_t text_len;			// of that destination
	uint32_t shift;
	uint32_t pad;
};
static_assert(32 == sizeof(fast_entry), "fast_entry layout");

__device__ __forceinline__ fast_entry make_fast_entry(dev_entry const *__restrict__ entries, uint32_t id)
{
	dev_entry const &e = entries[id];
	fast_entry f;
	f.pwin = e.pwin; f.n_zeros = e.n_zeros; f.n_cols = e.n_cols; f.partner = e.pwin_partner; f.shift = e.pwin_shift; f.pad = 0;
	f.text_len = e.pwin ? entries[e.pwin_partner].text_len : 0;
	return f;
}

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

// t_wide: 256-byte windows, for reads of up to 255 bases.  Only there can a record need a ninth piece of text, and the code
// that fetches it costs the narrow kernel registers it does not have.
template <bool t_wide, bool t_wary = false>
__global__ __launch_bounds__(256, PVC3_FAST_MIN_WAVES) void k_project_fast(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W, uint32_t use_fast_path)
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

	// The entry table is read by every lane of every group: keep its head in LDS, in the form this kernel needs.
	__shared__ fast_entry lds_entries[LDS_ENTRIES];
	for (uint32_t i = threadIdx.x; i < LDS_ENTRIES && i < n_entries; i += blockDim.x)
		lds_entries[i] = make_fast_entry(entries, i);
	__syncthreads();
	// Per-wave staging for the cooperative compare: record descriptors out, mismatch words back.
	__shared__ u64x2 lds_desc[4][64];
	__shared__ uint32_t lds_mask[4][STAGE_TOKENS];
	// What only the few records that go on to k_project_mid need again (their work-list items): parked in LDS instead of
	// eight registers that would be live across the compare.
	__shared__ u32x4 lds_park[4][2][64];
	uint32_t const wave_in_block = rfl(threadIdx.x >> 6);
	u64x2 *my_desc = lds_desc[wave_in_block];
	uint32_t *my_mask = lds_mask[wave_in_block];
	u32x4 (*my_park)[64] = lds_park[wave_in_block];
	// Entries beyond the table ("far") are handled in branches of their own that leave their results in LDS: a value that
	// is loaded from global memory on a rare path and merged with the common path's LDS read in one register makes the
	// compiler wait for ALL outstanding vector memory operations (s_waitcnt vmcnt(0)) before that LDS read on the common
	// path too — which serialised the loads requested ahead with the group's own.
	auto near_entry = [&](uint32_t id) -> fast_entry { return lds_entries[(id < LDS_ENTRIES) ? id : (LDS_ENTRIES - 1)]; };
	// A batch whose offsets are malformed (k_validate_batch; a pipelined batch is checked while its kernels are already
	// queued) is not read at all: every record is rejected and the host reports PVC3_E_INVALID.
	bool const batch_bad = 0xFFFFFFFFu != rfl(*W.first_bad);
	uint64_t acc_bitvec = 0, acc_ref = 0, acc_bytes = 0;

#ifdef PVC3_FAST_STAMPS
	uint64_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint64_t stamp_prev = __builtin_amdgcn_s_memtime();
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
	// ---- the descriptors of a group, requested one group ahead ---------------------------------------------------------------------
	// Nothing a group loads may wait for another load of the same group if it can be helped: at four waves per SIMD every
	// dependent memory round trip of a group is time nobody fills (occupancy experiment).  The descriptors of
	// group g + 1 and the 256 CIGAR operations that follow the last operation of group g — where group g + 1's begin, which
	// is known without its descriptors — are requested at the top of group g; group g + 1 then starts with everything in
	// registers and what it loads itself (window header, SEQ, window text) is one round trip.
	struct group_desc
	{
		uint32_t src_pos, src_e, dst_e, fl, l_seq, mate_se, mate_de, mate_pos;
		uint64_t cig_begin, cig_end, seq_begin;
	};
	bool const mates = nullptr != in.mate_pos;		// wave-uniform
	auto load_desc = [&](uint64_t g) -> group_desc {
		uint64_t const rec = g * 64 + lane;
		uint64_t const rl = (rec < in.n) ? rec : (in.n - 1);		// clamped index: every lane issues the same loads
		group_desc d;
		d.src_pos = in.src_pos[rl]; d.src_e = in.src_entry[rl]; d.dst_e = in.dst_entry[rl]; d.fl = in.flags[rl];
		d.cig_begin = in.cigar_off[rl]; d.cig_end = in.cigar_off[rl + 1];
		d.seq_begin = in.seq_off[rl]; d.l_seq = in.l_seq[rl];
		// (The mate fields are loaded whether or not the flags announce a mate: a load that waits for the flags is a memory
		// round trip of its own.)
		// (Without mate arrays the loads go to another array and are ignored: no branch, so that the compiler can count the
		// operations in flight instead of waiting for all of them.)
		d.mate_se = (mates ? in.mate_src_entry : in.src_pos)[rl]; d.mate_de = (mates ? in.mate_dst_entry : in.src_pos)[rl]; d.mate_pos = (mates ? in.mate_pos : in.src_pos)[rl];
		return d;
	};
	// The CIGAR operations [base + 4 lane, base + 4 lane + 4): 1 KB per wave, coalesced.  The buffer is padded by 64 operations.
	uint64_t const cigar_total = batch_bad ? 0 : rfl64(in.cigar_off[in.n]);
	auto load_chunk = [&](uint64_t base) -> u32x4 {
		// Lanes past the end of the (padded) buffer read its last 16 bytes: what they get is never used.
		uint64_t const at = base + 4 * lane, last = cigar_total + 60;
		u32x4 v;
		__builtin_memcpy(&v, in.cigar + ((at < last) ? at : last), 16);
		return v;
	};
	group_desc next_desc = load_desc(g_begin);
	uint64_t next_chunk_base = (g_begin < g_end && !batch_bad) ? rfl64(in.cigar_off[64 * g_begin]) : 0;
	u32x4 next_chunk = load_chunk(next_chunk_base);

	// The compare is requested before the window's header says whether it will be needed (one memory round trip instead of two).
	// Where hardly any window is clean — rows with a gap every few dozen columns — nearly all of that is thrown away.  The wary
	// form of the kernel (t_wary: launched for an index with such rows, pvc3_context_upload_index decides) keeps a credit per wave,
	// 0 .. 3: up for a group of which one candidate in sixteen was finished here, down otherwise; a wave without credit asks for the
	// compare only in every eighth group (whose outcome can earn the credit back) and leaves the other groups' candidates to
	// k_project_mid, which takes any record.  C3G: 2.5 -> 1.8 ms.  The choice depends on the wave's own history alone, so a run
	// repeats itself.  (A form of its own, because the plain kernel has no register to spare: with the credit in it — even in LDS —
	// it spills, and C3 is 3 % slower.)
	__shared__ uint32_t lds_credit[t_wary ? 4 : 1];
	uint32_t *const credit = &lds_credit[t_wary ? rfl(threadIdx.x >> 6) : 0];
	if (t_wary && 0 == lane) *credit = 3;
	for (uint64_t g = g_begin; g < g_end; ++g)
	{
		uint64_t const rec = g * 64 + lane;
		bool const valid = rec < in.n;
		uint64_t const rl = valid ? rec : (in.n - 1);

		// ---- descriptors: those requested a group ago; the next group's are requested now -----------------------------------------
		group_desc const D = next_desc;
		u32x4 const chunk = next_chunk;
		uint64_t const chunk_base = next_chunk_base;
		uint32_t const src_pos = D.src_pos, src_e = D.src_e, dst_e = D.dst_e, fl = D.fl, l_seq = D.l_seq;
		uint32_t const mate_se = D.mate_se, mate_de = D.mate_de, mate_pos = D.mate_pos;
		uint64_t const cig_begin = batch_bad ? 0 : D.cig_begin;
		uint32_t const n_cigar = batch_bad ? 0 : (uint32_t) (D.cig_end - D.cig_begin);
		uint64_t const seq_begin = D.seq_begin;
		bool const has_mate = valid && mates && (fl & PVC3_FLAG_HAS_MATE);
		// The group's operations go through LDS to the lanes that own them (the area is free here: the parked fields of the
		// previous group have been read).
		uint32_t *stage = (uint32_t *) &my_park[0][0];
		*(u32x4 *) (stage + 4 * lane) = chunk;
		{
			// (After the wave's last group this requests that group's descriptors once more: harmless, and no branch.)
			next_desc = load_desc((g + 1 < g_end) ? (g + 1) : g);
			// The last record's end is the next group's begin.
			next_chunk_base = batch_bad ? 0 : (((uint64_t) readlane((uint32_t) (D.cig_end >> 32), 63) << 32) | readlane((uint32_t) D.cig_end, 63));
			next_chunk = load_chunk(next_chunk_base);
		}
		wave_lds_sync();

		FAST_STAMP(0);		// descriptors
		bool const e_ok = (src_e < n_entries) && (dst_e < n_entries);
		fast_entry S = near_entry(e_ok ? src_e : 0);
		uint32_t d_cols = near_entry(e_ok ? dst_e : 0).n_cols;
		{
			// A far entry: what the status needs of it, through LDS; the record is then left to k_project_mid.
			bool const far = e_ok && (src_e >= LDS_ENTRIES || dst_e >= LDS_ENTRIES);
			if (ballot(far))
			{
				uint32_t *slot = (uint32_t *) &my_desc[lane];
				if (far)
				{
					slot[0] = entries[src_e].n_zeros; slot[1] = entries[src_e].n_cols; slot[2] = entries[dst_e].n_cols;
				}
				wave_lds_sync();
				if (far)
				{
					S.n_zeros = slot[0]; S.n_cols = slot[1]; d_cols = slot[2];
					S.pwin = nullptr;
				}
				wave_lds_sync();
			}
		}

		// ---- CIGAR: validation + shape automaton of the fast path (first four operations from one 16-byte load) ----------------
		uint32_t const *ops = in.cigar + cig_begin;
		u32x4 pre4;
		{
			uint64_t const at = cig_begin - chunk_base;		// (wraps to a large number if the offsets are not what the chunk assumed)
			bool const staged_ops = at + 4 <= 256;
			uint32_t const *sp = staged_ops ? (stage + (uint32_t) at) : (uint32_t const *) &my_desc[lane];
			if (ballot(!staged_ops))
			{
				// A group with more than 252 operations before this record's: fetched here, handed over through LDS like the
				// others (see the note on far entries).
				if (!staged_ops)
				{
					u32x4 v{0, 0, 0, 0};
					if (n_cigar) __builtin_memcpy(&v, ops, 16);			// the CIGAR buffer is padded by the host side
					*(u32x4 *) &my_desc[lane] = v;
				}
				wave_lds_sync();
			}
			pre4 = u32x4{sp[0], sp[1], sp[2], sp[3]};
		}
		wave_lds_sync();		// the parked fields go to the same area
		uint32_t const pre[4] = {pre4.x, pre4.y, pre4.z, pre4.w};
		uint32_t ref_consumed = 0, query_consumed = 0, shape = 0, lead = 0, trail = 0;
		bool bad_op = false;
		// (Plain selects: an `if` that assigns `lead` in one arm and `trail` in another is turned into one store through a
		// selected address, which puts both variables into scratch memory — and every scratch access is a vector memory
		// operation that the wave then waits for together with everything else in flight.)
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
		else if (S.n_cols != d_cols || (uint64_t) src_pos + (ref_consumed ? ref_consumed : 1u) > S.n_zeros) status = PVC3_ST_SRC_RANGE;
		else if (query_consumed > l_seq) status = PVC3_ST_QUERY_RANGE;
		bool const ok = valid && (PVC3_ST_OK == status);
		my_park[0][lane] = u32x4{src_e, dst_e, n_cigar, l_seq};
		my_park[1][lane] = u32x4{(uint32_t) cig_begin, (uint32_t) (cig_begin >> 32), (uint32_t) seq_begin, (uint32_t) (seq_begin >> 32)};
		if (valid)
		{
			acc_failed += (PVC3_ST_OK != status);
			// every record: 7 descriptor fields (28 + 4 flags), input CIGAR, select sample, destination rank word; 24 bytes of results
			acc_bytes += 32 + 4 * n_cigar + (ok ? 12 : 0) + 24 + (has_mate ? 12 : 0);
		}
		// what the fast path itself must read and write for a record it finishes: SEQ, gap-bit windows, text (+ CIGAR out)
		uint32_t const fast_bytes = (l_seq + 1) / 2 + 2 * ((ref_consumed + 8) / 8) + (ref_consumed + 1) / 2;

		// ---- the record's window: header requested now, read after the compare ---------------------------------------------------------
		bool const has_win = ok && S.pwin && S.partner == dst_e;
		uint32_t const text_len = S.text_len;
		uint32_t const sh = S.shift;
		uint32_t const o = src_pos & ((1u << sh) - 1);		// the record starts o source positions into its window
		uint8_t const *line = S.pwin + ((uint64_t) (src_pos >> sh) << (sh + 1));
		u32x4 hdr{0, 0, 0, 0};
		if (has_win) hdr = global_load16(line);

		// The mate's window as well, unless the neighbouring lane is about to look the same position up (pairs stored back
		// to back): project_position is a pure function of (entries, position).
		bool const m_in = has_mate && mate_se < n_entries && mate_de < n_entries;
		bool nb_cand;
		{
			uint32_t const n_se = (uint32_t) __shfl_xor((int) src_e, 1), n_de = (uint32_t) __shfl_xor((int) dst_e, 1);
			uint32_t const n_pos = (uint32_t) __shfl_xor((int) src_pos, 1);
			nb_cand = has_mate && (mate_se == n_se) && (mate_de == n_de) && (mate_pos == n_pos);
		}
		bool m_win = false, m_valid = false;
		uint32_t mo = 0;
		u32x4 mh{0, 0, 0, 0};
		if (m_in)
		{
			fast_entry const M = near_entry(mate_se);
			bool const m_far = mate_se >= LDS_ENTRIES || mate_de >= LDS_ENTRIES;		// left to thread_project_position below
			m_valid = m_far || ((mate_pos < M.n_zeros) && (M.n_cols == near_entry(mate_de).n_cols));		// msa_index.hh:253-258 needs the position to exist
			m_win = m_valid && !m_far && M.pwin && M.partner == mate_de;
			mo = mate_pos & ((1u << M.shift) - 1);
			if (m_win && !nb_cand) mh = global_load16(M.pwin + ((uint64_t) (mate_pos >> M.shift) << (M.shift + 1)));
		}
		bool const count_mate = m_in && !nb_cand;
		// The per-lane conditions that are needed again after the compare, as bits of one register: as booleans each of them
		// is a lane mask in a pair of SGPRs, and the kernel has more such pairs alive than SGPRs.
		enum : uint32_t { B_VALID = 1, B_OK = 2, B_HAS_WIN = 4, B_NB = 8, B_M_WIN = 16, B_M_VALID = 32, B_COUNT_MATE = 64, B_CAND = 128 };
		uint32_t bits = (valid ? B_VALID : 0u) | (ok ? B_OK : 0u) | (has_win ? B_HAS_WIN : 0u) | (nb_cand ? B_NB : 0u) | (m_win ? B_M_WIN : 0u)
			| (m_valid ? B_M_VALID : 0u) | (count_mate ? B_COUNT_MATE : 0u);
		FAST_STAMP(2);		// window headers (the diagnostic build waits for them here; the kernel reads them after the compare)

		// ---- fast path: the compare, on every record that can still turn out to be one --------------------------------------------------
		uint32_t const lead_q = (OP_S == (lead & 0xf)) ? (lead >> 4) : 0;	// query bases before the M-type span
		uint32_t const phase = lead_q & 1;
		// Work in query-nibble positions u = t + phase counted from a byte boundary of SEQ, so that 32 positions are exactly
		// 16 bytes of SEQ and 32 nibbles of text; the mask is shifted back by `phase` afterwards.
		// Both sides are loaded in 16-byte pieces from DWORD-aligned addresses and shifted into place in registers (SEQ by
		// whole bytes, the text by nibbles; the dword a shift pulls in comes from the lane of the next piece): a 16-byte load
		// from an address that is not a multiple of four costs the texture addresser three times as much
		// (tools/calib/ta_cost.hip: 65 against 22 cycles for this lane pattern), and these loads were 60 % of its time.
		uint8_t const *qb = in.seq4 + seq_begin + (lead_q >> 1);		// SEQ byte of query position u = 0
		uint32_t const nib = o + PWIN_TEXT_BACK - phase;			// nibble of the window's text that lines up with u = 0
		uint32_t const sb = (uint32_t) (uintptr_t) qb & 3, st = nib & 7;	// SEQ shift in bytes, text shift in nibbles
		uint32_t pieces;											// 16-byte pieces of either side the compare has to load
		bool cand;
		{
			bool const simple = (2 == shape || 3 == shape);
			uint32_t const U = ref_consumed + phase;
			uint32_t const pq = (2 * sb + U + 31) / 32, pt = (st + U + 31) / 32;
			pieces = (pq > pt) ? pq : pt;
			// (A ninth piece — 250 bases and an unlucky shift — is never more than the one dword that the shift pulls in: the
			// lane of the eighth piece loads it itself.)
			uint32_t const bytes = (pieces <= FAST_WORDS) ? 16 * pieces : 16 * FAST_WORDS + 4;
			bool const fits = (pieces <= FAST_WORDS + (t_wide ? 1 : 0)) && (PWIN_HEADER + 4 * (nib >> 3) + bytes <= (2u << sh));
			bool speculate = true;
			if constexpr (t_wary) speculate = (0 != *credit) || (0 == ((g - g_begin) & 7));
			cand = speculate && use_fast_path && has_win && simple && (U < FAST_MAX_BASES) && fits;
			bits |= cand ? B_CAND : 0u;
		}
		bool fast = false;
		uint32_t col = 0, r0 = 0, skip = PWIN_NO_ANCHOR;
		uint32_t f_n_out = 0;
		uint64_t f_loc = 0;
		uint32_t n_words = 0;		// wave-uniform: lanes per record in the compare, mask words per record afterwards
#ifdef PVC3_ABL_NO_COMPARE
		if (false)
#else
		if (ballot(cand))
#endif
		{
			uint32_t const U = cand ? (ref_consumed + phase) : 0;
			n_words = wave_max(cand ? ((pieces < FAST_WORDS) ? pieces : FAST_WORDS) : 0);		// 1 .. FAST_WORDS
			// The compare is done by n_words lanes per record, one 32-position word each: a wave instruction then touches the
			// lines of 64 / n_words records instead of one line per lane, which the memory pipeline serves 2-3x faster
			// (tools/calib/random_access.hip).  Descriptors go to the lanes through LDS, the masks come back the same way.
			{
				u64x2 d;
				d.x = ((uint64_t) (qb - sb) & UINT64_C(0xFFFFFFFFFFFF)) | ((uint64_t) U << 48);
				// (pieces = 0 for a record that is no candidate: nothing is loaded for it, its addresses may be anything)
				d.y = ((uint64_t) (line + PWIN_HEADER + 4 * (nib >> 3)) & UINT64_C(0xFFFFFFFFFFFF)) | ((uint64_t) (sb | (st << 2) | ((cand ? pieces : 0) << 5)) << 48);
				my_desc[lane] = d;
			}
			wave_lds_sync();
			{
				uint32_t const inv = (65536 + n_words - 1) / n_words;
				uint32_t const per = (64 * inv) >> 16;			// records per pass
				uint32_t const sr = (lane * inv) >> 16;				// lane / n_words
				uint32_t const w = lane - sr * n_words;
				// (`how`: bits 0-7 positions left to compare from this word on, 8-12 the two shifts, 13 "no lane holds the ninth piece")
				auto fetch = [&](uint32_t rr, bool act, u32x4 &qq, u32x4 &tt, uint32_t &how) {
					u64x2 const dd = my_desc[act ? rr : lane];
					uint32_t const Ur = (uint32_t) (dd.x >> 48), meta = (uint32_t) (dd.y >> 48);
					uint32_t const left = (act && Ur > 32 * w) ? (Ur - 32 * w) : 0;
					uint32_t shifts = meta & 31;
					if (act && w < (meta >> 5))		// both loads stay inside the record's SEQ (+ padding) and the window's line
					{
						qq = global_load16((uint32_t const *) (dd.x & UINT64_C(0xFFFFFFFFFFFF)) + 4 * w);
						tt = global_load16((uint32_t const *) (dd.y & UINT64_C(0xFFFFFFFFFFFF)) + 4 * w);
					}
					if (t_wide) shifts |= (act && FAST_WORDS - 1 == w && FAST_WORDS < (meta >> 5)) ? 0x100u : 0u;		// no lane holds the ninth piece
					how = left | (shifts << 8);
				};
				auto finish = [&](uint32_t rr, bool act, u32x4 const &qq, u32x4 const &tt, uint32_t how) {
					uint32_t const left = how & 0xffu, shifts = how >> 8;
					// The dword after a lane's 16 bytes is the first dword of the next lane's piece (a record's last piece
					// never needs one: `pieces` counts what the shifts pull in).
					uint32_t nq = (uint32_t) __shfl_down((int) qq.x, 1), nt = (uint32_t) __shfl_down((int) tt.x, 1);
					if (t_wide && (shifts & 0x100u))
					{
						// The ninth piece's first dword, from lines this lane has just read (a dependent load, but rare:
						// 250 bases and a shift of seven nibbles; kept out of `fetch` for the registers it would hold there).
						typedef uint32_t __attribute__((address_space(1))) const *gptr;
						u64x2 const dd = my_desc[rr];
						nq = ((gptr) (uintptr_t) (dd.x & UINT64_C(0xFFFFFFFFFFFF)))[4 * FAST_WORDS];
						nt = ((gptr) (uintptr_t) (dd.y & UINT64_C(0xFFFFFFFFFFFF)))[4 * FAST_WORDS];
					}
					uint32_t m = 0;
					if (left)
					{
						uint32_t const bq = 8 * (shifts & 3), bt = 4 * ((shifts >> 2) & 7);
						uint32_t const q0 = __builtin_amdgcn_alignbit(qq.y, qq.x, bq), q1 = __builtin_amdgcn_alignbit(qq.z, qq.y, bq);
						uint32_t const q2 = __builtin_amdgcn_alignbit(qq.w, qq.z, bq), q3 = __builtin_amdgcn_alignbit(nq, qq.w, bq);
						uint32_t const t0 = __builtin_amdgcn_alignbit(tt.y, tt.x, bt), t1 = __builtin_amdgcn_alignbit(tt.z, tt.y, bt);
						uint32_t const t2 = __builtin_amdgcn_alignbit(tt.w, tt.z, bt), t3 = __builtin_amdgcn_alignbit(nt, tt.w, bt);
						u64x2 q64;
						q64.x = (uint64_t) q0 | ((uint64_t) q1 << 32); q64.y = (uint64_t) q2 | ((uint64_t) q3 << 32);
#ifdef PVC3_ABL_NO_MISMATCH
						m = ((uint32_t) q64.x ^ (uint32_t) q64.y ^ (uint32_t) (q64.x >> 32) ^ (uint32_t) (q64.y >> 32) ^ t0 ^ t1 ^ t2 ^ t3) & (uint32_t) low_mask(left < 32 ? left : 32) & 0x10101u;
#else
						m = mismatch32(q64, t0, t1, t2, t3) & (uint32_t) low_mask(left < 32 ? left : 32);
#endif
					}
					if (act && w < FAST_WORDS) my_mask[rr * FAST_WORDS + w] = m;
				};
				// FAST_PASSES passes at a time: all their loads are in flight before the first word is compared (a pass per round
				// trip made the compare three dependent round trips for 150-base reads, a quarter of the kernel's time).
				for (uint32_t base = 0; base < 64; base += FAST_PASSES * per)
				{
					u32x4 qv[FAST_PASSES], tv[FAST_PASSES];
					uint32_t how_v[FAST_PASSES];
#pragma unroll
					for (uint32_t k = 0; k < FAST_PASSES; ++k)
					{
						uint32_t const rr = base + k * per + sr;
						qv[k] = u32x4{0, 0, 0, 0}; tv[k] = u32x4{0, 0, 0, 0};
						fetch(rr, (sr < per) & (rr < 64), qv[k], tv[k], how_v[k]);
					}
#pragma unroll
					for (uint32_t k = 0; k < FAST_PASSES; ++k)
					{
						uint32_t const rr = base + k * per + sr;
						finish(rr, (sr < per) & (rr < 64), qv[k], tv[k], how_v[k]);
					}
				}
			}
			wave_lds_sync();
			FAST_STAMP(5);		// compare
		}

		// ---- what the window says (rewrite_cigar.hh:51-56) ---------------------------------------------------------------------------------
		if (bits & B_HAS_WIN)
		{
			uint32_t const clean = hdr.z;
			// Clean columns hold one source position and one destination position each: inside the clean stretch the start
			// column and the projected position are sums.  Past it they are left to resolve_start (wave.hpp), from the
			// stretch's end.
			uint32_t const adv = (o < clean) ? o : clean;
			col = hdr.x + adv;
			r0 = hdr.y + adv;
			skip = (o < clean) ? 0 : (o - clean + 1);
			// No gap column in either row from the window's first column to the record's last.
			fast = (bits & B_CAND) && (o + ref_consumed <= clean) && ((uint64_t) r0 + ref_consumed <= text_len);
		}
		bool const exact = (bits & B_HAS_WIN) && (0 == skip);
		FAST_STAMP(3);		// window header

		// ---- mate (project_alignments.cc:897-901) -----------------------------------------------------------------------------------
		uint32_t mate = PVC3_NO_MATE;
		{
			uint32_t const n_r0 = (uint32_t) __shfl_xor((int) r0, 1);
			bool const n_exact = 0 != __shfl_xor((int) exact, 1), n_has_win = 0 != (__shfl_xor((int) bits, 1) & B_HAS_WIN);
			bool const nb = 0 != (bits & B_NB), mw = 0 != (bits & B_M_WIN);
			if (nb && n_exact)
				mate = n_r0;
			else if (nb && n_has_win)
				mate = MATE_FROM_NEIGHBOUR;		// the neighbour goes to k_project_mid, which resolves its start: it fills this in
			else if (mw && !nb && mo < mh.z)
				mate = mh.y + mo;
			else if (bits & B_M_VALID)
			{
				// The rare cases: the mate's window is not clean up to the mate, the neighbour's own look-up is left to
				// k_project_mid, or the pair of entries has no windows.  (The mate's descriptor is read again here rather than
				// kept in registers across the compare.)
				uint32_t const se = in.mate_src_entry[rl], de = in.mate_dst_entry[rl], pos = in.mate_pos[rl];
				if (mw)
				{
					dev_entry const &MS = entries[se];
					if (nb) mh = global_load16(MS.pwin + ((uint64_t) (pos >> MS.pwin_shift) << (MS.pwin_shift + 1)));
					if (mo < mh.z)
						mate = mh.y + mo;
					else
					{
						uint32_t c = mh.x + mh.z, r = mh.y + mh.z;
						resolve_start(MS, entries[de], de, pos, c, r, mo - mh.z + 1);
						mate = r;
					}
				}
				else
					mate = thread_project_position(entries[se], entries[de], de, pos);
			}
			if (bits & B_COUNT_MATE)
				acc_bytes += 4 + 64 + 64;		// select sample, one source block, one destination block
		}
		FAST_STAMP(4);		// mate

		// ---- fast path: run counts, arena space, emission ----------------------------------------------------------------------------------
#ifdef PVC3_ABL_NO_COMPARE
		fast = false;
#endif
		if constexpr (t_wary)
		{
			uint32_t const tried = (uint32_t) popc64(ballot(0 != (bits & B_CAND))), won = (uint32_t) popc64(ballot(fast));
			uint32_t const have = *credit;
			if (tried && 0 == lane) *credit = (16 * won >= tried) ? ((have < 3) ? (have + 1) : 3u) : ((have > 0) ? (have - 1) : 0u);
		}
		if (ballot(fast))
		{
			uint32_t const R = fast ? ref_consumed : 0;
			uint32_t mu[FAST_WORDS + 1];
			{
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
				acc_bytes += fast_bytes + 4 * f_n_out;
			}
		}

		// ---- work list of the records left for k_project_mid --------------------------------------------------------------------------
		{
			uint64_t const m = ballot((bits & B_OK) && !fast);
			if (g == g_begin) items_mode = popc64(m) <= 32;
			if ((m >> lane) & 1)
			{
				uint32_t const at = n_slow + mbcnt64(m);
				slow_list[at] = (uint32_t) rec;
			}
			if (items_mode && ((m >> lane) & 1))
			{
				uint32_t const at = n_slow + mbcnt64(m);
				u32x4 const a = my_park[0][lane], b = my_park[1][lane];
				slow_item it;
				it.rec = (uint32_t) rec; it.col0 = col; it.r0 = r0; it.src_e = a.x;
				it.dst_e = a.y; it.n_cigar = a.z; it.l_seq = a.w; it.skip = skip;
				it.cig_begin = (uint64_t) b.x | ((uint64_t) b.y << 32); it.seq_begin = (uint64_t) b.z | ((uint64_t) b.w << 32);
				slow_items[at] = it;
			}
			n_slow += popc64(m);
		}

		// ---- per-record outputs (coalesced) ----------------------------------------------------------------------------------------------
		if (bits & B_VALID)
		{
			bool const slow = (bits & B_OK) && !fast;
			W.mate_dst_pos[rec] = mate;
			W.dst_pos[rec] = (bits & B_OK) ? r0 : 0;		// a rejected record has empty outputs; a record with skip != 0 gets its position from the walk kernels
			W.rec_flags[rec] = slow ? REC_SLOW : status;
			W.cig_n[rec] = slow ? skip : f_n_out;	// (slow: see resolve_start; the walk kernels store the number of operations)
			W.cig_loc[rec] = slow ? ((uint64_t) col | ((uint64_t) ref_consumed << 32)) : f_loc;
			W.run_n[rec] = 0;
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

uint32_t grid_blocks(uint64_t n)
{
	// Persistent waves: 4 waves per block, at most 8 blocks per CU; at least one group of 64 records per wave.
	uint64_t const groups = (n + 63) / 64;
	uint64_t const blocks = (groups + 3) / 4;
	static uint32_t const max_blocks = [] { char const *e = getenv("PVC3_GRID_BLOCKS"); return e ? (uint32_t) atoi(e) : 256u * 8; }();	// (experiments)
	return (uint32_t) (blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks));
}

}	// anonymous namespace

uint32_t project_wave_count(uint64_t n) { return 4 * grid_blocks(n); }

int launch_project_fast(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, uint32_t use_fast_path, bool wide_windows, bool wary, void *stream)
{
	if (0 == in.n) return 0;
	// PVC3_FAST_DYN_LDS: bytes of unused dynamic LDS per block — an occupancy experiment (fewer blocks per CU), not an option.
	static uint32_t const dyn_lds = [] { char const *e = getenv("PVC3_FAST_DYN_LDS"); return e ? (uint32_t) atoi(e) : 0u; }();
	dim3 const grid(grid_blocks(in.n)), block(256);
	hipStream_t const s = (hipStream_t) stream;
	if (wide_windows && wary) hipLaunchKernelGGL((k_project_fast <true, true>), grid, block, dyn_lds, s, entries, n_entries, in, work, use_fast_path);
	else if (wide_windows) hipLaunchKernelGGL((k_project_fast <true, false>), grid, block, dyn_lds, s, entries, n_entries, in, work, use_fast_path);
	else if (wary) hipLaunchKernelGGL((k_project_fast <false, true>), grid, block, dyn_lds, s, entries, n_entries, in, work, use_fast_path);
	else hipLaunchKernelGGL((k_project_fast <false, false>), grid, block, dyn_lds, s, entries, n_entries, in, work, use_fast_path);
	return (int) hipGetLastError();
}

}
