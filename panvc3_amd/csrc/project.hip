// project.hip — k_project: the projection walk (rewrite_cigar + on-the-fly indel-run detection + mate projection).
//
// Restates rewrite_cigar (include/panvc3/rewrite_cigar.hh:18-157) through the column-walk equivalence,
// cigar_buffer (libpanvc3/cigar.cc:54-95), the run detection of indel_run_checker (libpanvc3/indel_run_checker.cc:31-121)
// and msa_index::sequence_entry::project_position (include/panvc3/msa_index.hh:253-258, call site
// project-alignments/project_alignments.cc:897-901).
//
// Each wavefront owns a contiguous range of records and handles them 64 at a time in two phases:
//   A  lane = record.  Coalesced SoA descriptor loads, CIGAR validation, one select sample + block reads per lane give the
//      start column and the projected position of all 64 records (and of their mates) in a handful of gathers, so the
//      dependent memory round trips are paid once per 64 records instead of once per record.
//   B  lane = MSA column.  Record by record, the wave walks 64 columns per step over the src / dst gap words held in a
//      two-block register window, builds the = X I D token masks with ballots and run-length merges them with scalar bit
//      arithmetic.  The loads of record i + 1 (window, SEQ, text, CIGAR) are issued before record i is walked.
// Integer / bit work; HBM-latency- and issue-bound, no MFMA.
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

constexpr uint32_t SINK_LANES = 192;			// runs buffered in three VGPRs before spilling to the arena
constexpr uint32_t SINK_SPILL_AT = SINK_LANES - 66;
constexpr uint32_t SEQ_WINDOW_BASES = 512;		// 64 lanes x 8 four-bit bases
constexpr uint32_t REF_WINDOW_BYTES = 256;

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

// BAM 4-bit code -> seqan3::dna5::to_char() ("=ACMGRSVTWYHKDBN" read through dna5::assign_char).
__device__ __forceinline__ uint32_t seq4_to_dna5_char(uint32_t code)
{
	return (1 == code) ? 'A' : (2 == code) ? 'C' : (4 == code) ? 'G' : (8 == code) ? 'T' : 'N';
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t lane)
{
	return ((uint64_t) readlane((uint32_t) (v >> 32), lane) << 32) | readlane((uint32_t) v, lane);
}

// Phase A results; lane = record of the current group of 64.
struct lane_state
{
	uint32_t status, col, r0, ref_consumed, n_cigar, l_seq, src_e, dst_e, src_pos;
	uint64_t cig_begin, seq_begin;
};

// Prefetched per-lane registers of one record.
struct rec_loads
{
	uint32_t win_lo, win_hi, seqw, refw, cigv;
};

// Wave-uniform facts of one record.
struct rec_ctx
{
	uint32_t status, col, r0, ref_consumed, n_cigar, l_seq, src_e, src_pos, d_text_len, b0, ref_base;
	uint64_t const *s_blocks, *d_blocks;
	uint8_t const *d_text;
	uint32_t const *cigar;
	uint8_t const *seq4;
};

struct rec_out
{
	uint32_t status, n_out, has_run, cols, ref_span;
	uint64_t loc;
};

__device__ __forceinline__ void stage_load(
	dev_entry const *__restrict__ entries, dev_batch_in const &in, lane_state const &A, uint32_t i, uint32_t lane, rec_ctx &C, rec_loads &L
)
{
	C.status = readlane(A.status, i);
	L.win_lo = 0; L.win_hi = 0; L.seqw = 0; L.refw = 0; L.cigv = 0;
	if (C.status)
		return;
	C.src_e = readlane(A.src_e, i);
	uint32_t const dst_e = readlane(A.dst_e, i);
	C.s_blocks = entries[C.src_e].blocks;
	C.d_blocks = entries[dst_e].blocks;
	C.d_text = entries[dst_e].text;
	C.d_text_len = entries[dst_e].text_len;
	C.col = readlane(A.col, i);
	C.r0 = readlane(A.r0, i);
	C.ref_consumed = readlane(A.ref_consumed, i);
	C.n_cigar = readlane(A.n_cigar, i);
	C.l_seq = readlane(A.l_seq, i);
	C.src_pos = readlane(A.src_pos, i);
	C.cigar = in.cigar + readlane64(A.cig_begin, i);
	C.seq4 = in.seq4 + readlane64(A.seq_begin, i);
	C.b0 = C.col / BLOCK_COLS;
	C.ref_base = C.r0 & ~3u;

	window w;
	window_load(w, C.s_blocks, C.d_blocks, C.b0, lane);
	L.win_lo = w.lo; L.win_hi = w.hi;
	if (8 * lane < C.l_seq)
		L.seqw = load_u32_unaligned(C.seq4 + 4 * lane);				// the SEQ buffer is padded by the host side
	if (C.d_text)
		L.refw = *(uint32_t const *) (C.d_text + C.ref_base + 4 * lane);	// device texts are padded by TEXT_PAD bytes
	if (lane < C.n_cigar)
		L.cigv = C.cigar[lane];
}

// Phase B for one record: the column walk.  All control flow is wave-uniform.
__device__ __forceinline__ void walk_record(
	dev_entry const *__restrict__ entries, rec_ctx const &C, rec_loads const &L, dev_batch_work const &W, arena_alloc &arena, uint32_t lane, rec_out &res
)
{
	res.status = C.status; res.n_out = 0; res.has_run = 0; res.loc = 0; res.cols = 0; res.ref_span = 0;
	if (C.status)
		return;

	window win;
	win.lo = L.win_lo; win.hi = L.win_hi; win.b0 = C.b0;
	uint32_t refw = L.refw, ref_base = C.ref_base;
	uint32_t cigv = L.cigv, cig_base = 0;
	uint32_t col = C.col, r = C.r0, q = 0;
	uint64_t spill_loc = 0;
	sink out;
	out.init();

	for (uint32_t ci = 0; ci < C.n_cigar; ++ci)
	{
		if (ci - cig_base >= 64)	// more than 64 input operations: next slice
		{
			cig_base = ci;
			cigv = (cig_base + lane < C.n_cigar) ? C.cigar[cig_base + lane] : 0;
		}
		uint32_t const c = readlane(cigv, ci - cig_base);
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
				dev_entry const S = entries[C.src_e];
				window tmp;
				uint32_t const col_end = wave_select0(S, C.d_blocks, C.src_pos + C.ref_consumed - 1, tmp, lane);
				uint32_t const bound = out.n_out + (col_end - col + 1) + (C.n_cigar - ci) + 2;
				uint64_t loc;
				if (!arena_take(arena, bound, W, lane, loc)) { res.n_out = 0; return; }
				uint32_t *dst = W.arena + loc;
				if (lane < out.n_out) dst[lane] = out.buf0;
				if (64 + lane < out.n_out) dst[64 + lane] = out.buf1;
				if (128 + lane < out.n_out) dst[128 + lane] = out.buf2;
				out.spill = dst;
				spill_loc = loc;
			}

			col_pos const p = split_col(col);
			if (p.blk - win.b0 >= 2)
				window_load(win, C.s_blocks, C.d_blocks, p.blk, lane);
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
					if (!C.d_text || max_ri >= C.d_text_len) { res.status = PVC3_ST_REF_RANGE; res.n_out = 0; return; }
					if (r + nd > ref_base + REF_WINDOW_BYTES)
					{
						ref_base = r & ~3u;
						refw = *(uint32_t const *) (C.d_text + ref_base + 4 * lane);
					}
					bool const in_c = (mC >> lane) & 1;
					uint32_t const qi = q + mbcnt64(Z);
					uint32_t const ri = r + mbcnt64(dz) - ref_base;
					uint32_t code;
					if (q + 64 <= SEQ_WINDOW_BASES)
					{
						uint32_t const wv = (uint32_t) __shfl((int) L.seqw, (int) (qi >> 3));
						uint32_t const byte = (wv >> (((qi >> 1) & 3) * 8)) & 0xff;
						code = (qi & 1) ? (byte & 0xf) : (byte >> 4);
					}
					else
					{
						uint32_t byte = 0;
						if (in_c) byte = C.seq4[qi >> 1];
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
	res.cols = col - C.col;
	res.ref_span = r - C.r0;
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


// ---- lane-parallel fast path ---------------------------------------------------------------------------------------
// A record is "simple" when its CIGAR is [S|H] (M|=|X)+ [S|H] with non-zero counts, at most FAST_MAX_BASES reference bases,
// and neither the source nor the destination row has a gap among the columns it spans.  Then column walk = string compare:
// lane = record, the mismatch bit-mask of all bases is built with plain per-lane loads and turned into = / X runs.
constexpr uint32_t FAST_MAX_BASES = 256;
constexpr uint32_t FAST_WORDS = FAST_MAX_BASES / 32;

struct simple_cigar
{
	uint32_t ok;			// shape matches
	uint32_t lead, trail;	// encoded clip operations or 0
	uint32_t n_bases;		// reference (= query) bases of the M-type span
};

// Nibble-linear view of 8 packed BAM bases starting at nibble `phase` of p: base k at bits [4k, 4k + 4).
__device__ __forceinline__ uint64_t load_bases8(uint8_t const *p, uint32_t phase)
{
	uint64_t w;
	__builtin_memcpy(&w, p, 8);
	w = ((w & UINT64_C(0x0F0F0F0F0F0F0F0F)) << 4) | ((w >> 4) & UINT64_C(0x0F0F0F0F0F0F0F0F));	// high nibble first -> linear
	return w >> (4 * phase);
}

__device__ __forceinline__ uint64_t load_u64_unaligned(uint8_t const *p)
{
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

// Bit k set iff base k of q (BAM code, via dna5::to_char) differs from the raw text byte k of r (quirk Q3).
__device__ __forceinline__ uint32_t mismatch8(uint64_t q, uint64_t r)
{
	// code -> index into "ACGTN" (1 2 4 8 -> 0 1 2 3, everything else -> 4), then the character itself.
	constexpr uint64_t IDX = UINT64_C(0x4444444344424104);
	constexpr uint64_t CHARS = UINT64_C(0x4E54474341);
	uint32_t m = 0;
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k)
	{
		uint32_t const code = (uint32_t) (q >> (4 * k)) & 0xf;
		uint32_t const rb = (uint32_t) (r >> (8 * k)) & 0xff;
		uint32_t const idx = (uint32_t) (IDX >> (4 * code)) & 0xf;
		uint32_t const ch = (uint32_t) (CHARS >> (8 * idx)) & 0xff;
		m |= (uint32_t) (ch != rb) << k;
	}
	return m;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
	for (uint32_t off = 1; off < 64; off <<= 1)
	{
		uint32_t const t = (uint32_t) __shfl_up((int) v, (int) off);
		if (lane >= off) v += t;
	}
	return v;
}

// Per-group staging of the per-record outputs (lane = record).
struct group_out
{
	uint32_t flags, cig_n, loc_lo, loc_hi;
	uint32_t n_failed, n_with_runs;		// wave-uniform accumulators
	uint64_t bitvec_bytes, ref_bytes;

	__device__ __forceinline__ void put(uint32_t i, rec_out const &r)
	{
		bool const ok = (0 == r.status);
		flags = writelane(flags, r.status | ((ok && r.has_run) ? REC_HAS_RUN : 0), i);
		cig_n = writelane(cig_n, ok ? r.n_out : 0, i);
		loc_lo = writelane(loc_lo, (uint32_t) r.loc, i);
		loc_hi = writelane(loc_hi, (uint32_t) (r.loc >> 32), i);
		n_failed += !ok;
		n_with_runs += (ok && r.has_run);
		bitvec_bytes += ok ? 2 * ((r.cols + 8) / 8) : 0;		// src + dst windows, at least the first column's byte
		ref_bytes += ok ? r.ref_span : 0;
	}
};

__global__ __launch_bounds__(256) void k_project(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W, uint32_t use_fast_path)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	// Contiguous group range per wave: sequential descriptor reads and a sequentially filled arena chunk.
	uint64_t const n_groups = (in.n + 63) / 64;
	uint64_t const g_begin = n_groups * wave / n_waves;
	uint64_t const g_end = n_groups * (wave + 1) / n_waves;
	arena_alloc arena{0, 0};
	group_out G;
	G.n_failed = 0; G.n_with_runs = 0; G.bitvec_bytes = 0; G.ref_bytes = 0;
	uint64_t acc_bitvec = 0, acc_ref = 0;		// per lane

	for (uint64_t g = g_begin; g < g_end; ++g)
	{
		uint64_t const base = g * 64;
		uint32_t const cnt = (in.n - base < 64) ? (uint32_t) (in.n - base) : 64;
		uint64_t const rec = base + lane;
		bool const valid = lane < cnt;

		// ---- phase A: lane = record ------------------------------------------------------------------------------------
		lane_state A;
		A.status = PVC3_ST_OK; A.col = 0; A.r0 = 0; A.ref_consumed = 0; A.n_cigar = 0; A.l_seq = 0; A.src_e = 0; A.dst_e = 0; A.src_pos = 0;
		A.cig_begin = 0; A.seq_begin = 0;
		uint32_t mate = PVC3_NO_MATE;
		simple_cigar sc;
		sc.ok = 0; sc.lead = 0; sc.trail = 0; sc.n_bases = 0;
		bool fast = false;
		if (valid)
		{
			A.src_pos = in.src_pos[rec];
			A.src_e = in.src_entry[rec];
			A.dst_e = in.dst_entry[rec];
			A.cig_begin = in.cigar_off[rec];
			A.n_cigar = (uint32_t) (in.cigar_off[rec + 1] - A.cig_begin);
			A.seq_begin = in.seq_off[rec];
			A.l_seq = in.l_seq[rec];
			uint32_t const fl = in.flags[rec];

			if (A.src_e >= n_entries || A.dst_e >= n_entries)
				A.status = PVC3_ST_SRC_RANGE;
			else
			{
				// Validation (defines the per-record status; same order as the oracle's batch driver).
				uint32_t query_consumed = 0;
				bool bad_op = false;
				uint32_t const *ops = in.cigar + A.cig_begin;
				uint32_t shape = 0;		// 0 start, 1 after the leading clip, 2 inside the M-type span, 3 after the trailing clip, 4 not simple
				for (uint32_t i = 0; i < A.n_cigar; ++i)
				{
					uint32_t const c = ops[i];
					uint32_t const op = c & 0xf, n = c >> 4;
					if (op > OP_X) { bad_op = true; break; }
					bool const m_type = (OP_M == op || OP_EQ == op || OP_X == op);
					bool const clip = (OP_S == op || OP_H == op);
					if (m_type) { A.ref_consumed += n; query_consumed += n; }
					else if (OP_D == op || OP_N == op) A.ref_consumed += n;
					else if (OP_I == op || OP_S == op) query_consumed += n;
					// shape automaton of the fast path
					if (0 == n) shape = 4;
					else if (m_type && shape <= 2) shape = 2;
					else if (clip && 0 == shape) { shape = 1; sc.lead = c; }
					else if (clip && 2 == shape) { shape = 3; sc.trail = c; }
					else shape = 4;
				}
				sc.ok = (2 == shape || 3 == shape);
				sc.n_bases = A.ref_consumed;
				dev_entry const *S = entries + A.src_e;
				dev_entry const *D = entries + A.dst_e;
				if (bad_op) A.status = PVC3_ST_CIGAR_OP;
				else if (S->n_cols != D->n_cols || (uint64_t) A.src_pos + (A.ref_consumed ? A.ref_consumed : 1u) > S->n_zeros) A.status = PVC3_ST_SRC_RANGE;
				else if (query_consumed > A.l_seq) A.status = PVC3_ST_QUERY_RANGE;
				else
				{
					// aligned_position + project_aligned_position (rewrite_cigar.hh:51-56)
					A.col = thread_select0(S->blocks, S->sel, A.src_pos);
					A.r0 = thread_rank0(D->blocks, A.col);
					if (use_fast_path && sc.ok && sc.n_bases <= FAST_MAX_BASES && D->text && (uint64_t) A.r0 + sc.n_bases <= D->text_len)
					{
						// No gap column in either row over the span: the ranks advance by exactly n_bases.
						uint32_t const col_end = A.col + sc.n_bases;
						fast = (thread_rank0(S->blocks, col_end) == A.src_pos + sc.n_bases) && (thread_rank0(D->blocks, col_end) == A.r0 + sc.n_bases);
					}
				}
			}
			// Mate: project_position (project_alignments.cc:897-901).
			if (in.mate_pos && (fl & PVC3_FLAG_HAS_MATE))
			{
				uint32_t const ms = in.mate_src_entry[rec], md = in.mate_dst_entry[rec];
				if (ms < n_entries && md < n_entries)
					mate = thread_project_position(entries[ms], entries[md], in.mate_pos[rec]);
			}
			W.mate_dst_pos[rec] = mate;
		}

		// ---- fast path: lane = record -------------------------------------------------------------------------------------
		uint32_t f_n_out = 0;
		uint64_t f_loc = 0;
		uint64_t const fast_mask = ballot(fast);
		if (fast_mask)
		{
			uint32_t const R = fast ? sc.n_bases : 0;
			uint32_t const lead_q = (fast && OP_S == (sc.lead & 0xf)) ? (sc.lead >> 4) : 0;	// query bases before the span
			uint8_t const *qp = in.seq4 + A.seq_begin + (lead_q >> 1);
			uint32_t const phase = lead_q & 1;
			uint8_t const *tp = entries[fast ? A.dst_e : 0].text + A.r0;
			// Mismatch mask, 32 bases per word.  The trip count is the wave maximum; lanes past their span contribute nothing.
			uint32_t mm[FAST_WORDS];
			uint32_t max_r = R;
#pragma unroll
			for (int off = 32; off; off >>= 1)
			{
				uint32_t const o = (uint32_t) __shfl_xor((int) max_r, off);
				max_r = (o > max_r) ? o : max_r;
			}
			uint32_t const n_words = (rfl(max_r) + 31) / 32;
#pragma unroll
			for (uint32_t w = 0; w < FAST_WORDS; ++w)
			{
				mm[w] = 0;
				if (w < n_words)
				{
					uint32_t m = 0;
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j)
					{
						uint32_t const t = 32 * w + 8 * j;
						if (t < R)		// per lane: loads stay inside the record's own bases (+ 7 nibbles / bytes of padding)
							m |= mismatch8(load_bases8(qp + (t >> 1), phase), load_u64_unaligned(tp + t)) << (8 * j);
					}
					uint32_t const left = (R > 32 * w) ? (R - 32 * w) : 0;
					mm[w] = m & (uint32_t) low_mask(left < 32 ? left : 32);
				}
			}
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
			f_n_out = fast ? (runs + (sc.lead ? 1 : 0) + (sc.trail ? 1 : 0)) : 0;
			uint32_t const scan = wave_inclusive_scan(f_n_out, lane);
			uint32_t const total = readlane(scan, 63);
			uint64_t base_loc = 0;
			bool const have = arena_take(arena, total, W, lane, base_loc);
			if (!have) f_n_out = 0;
			f_loc = base_loc + (scan - f_n_out);
			if (have && fast)
			{
				uint32_t *dst = W.arena + f_loc;
				uint32_t k = 0;
				if (sc.lead) dst[k++] = sc.lead;
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
							dst[k++] = ((pos - start) << 4) | (type ? OP_X : OP_EQ);
							start = pos; type ^= 1;
							tr &= tr - 1;
						}
					}
				}
				dst[k++] = ((R - start) << 4) | (type ? OP_X : OP_EQ);
				if (sc.trail) dst[k++] = sc.trail;
				acc_bitvec += 2 * ((R + 8) / 8);
				acc_ref += R;
			}
		}

		// ---- general path: lane = column, record by record, next record's loads in flight -------------------------------
		G.flags = 0; G.cig_n = 0; G.loc_lo = 0; G.loc_hi = 0;
		{
			uint64_t todo = ballot(valid && !fast);
			rec_ctx C0, C1;
			rec_loads L0, L1;
			rec_out R;
			if (todo)
			{
				uint32_t i0 = ctz64(todo), i1 = 0;
				todo &= todo - 1;
				stage_load(entries, in, A, i0, lane, C0, L0);
				for (;;)
				{
					bool const has1 = (0 != todo);
					if (has1) { i1 = ctz64(todo); todo &= todo - 1; stage_load(entries, in, A, i1, lane, C1, L1); }
					walk_record(entries, C0, L0, W, arena, lane, R);
					G.put(i0, R);
					if (!has1) break;
					bool const has0 = (0 != todo);
					if (has0) { i0 = ctz64(todo); todo &= todo - 1; stage_load(entries, in, A, i0, lane, C0, L0); }
					walk_record(entries, C1, L1, W, arena, lane, R);
					G.put(i1, R);
					if (!has0) break;
				}
			}
		}

		if (valid)
		{
			bool const ok = fast || (0 == (G.flags & REC_STATUS_MASK));
			W.dst_pos[rec] = ok ? A.r0 : 0;		// a rejected record has empty outputs
			W.rec_flags[rec] = fast ? 0 : G.flags;
			W.cig_n[rec] = fast ? f_n_out : G.cig_n;
			W.cig_loc[rec] = fast ? f_loc : (((uint64_t) G.loc_hi << 32) | G.loc_lo);
			W.run_n[rec] = 0;
		}
	}
	// Per-lane accumulators of the fast path -> wave totals.
	for (int off = 32; off; off >>= 1)
	{
		acc_bitvec += __shfl_xor(acc_bitvec, off);
		acc_ref += __shfl_xor(acc_ref, off);
	}
	G.bitvec_bytes += rfl64(acc_bitvec);
	G.ref_bytes += rfl64(acc_ref);
	if (0 == lane)
	{
		atomicAdd(&W.counters->bitvec_bytes, (unsigned long long) G.bitvec_bytes);
		atomicAdd(&W.counters->ref_bytes, (unsigned long long) G.ref_bytes);
		if (G.n_failed) atomicAdd(&W.counters->n_failed, (unsigned long long) G.n_failed);
		if (G.n_with_runs) atomicAdd(&W.counters->n_realigned_records, (unsigned long long) G.n_with_runs);
	}
}

uint32_t grid_blocks(uint64_t n)
{
	// Persistent waves: 4 waves per block, at most 8 blocks per CU; at least one group of 64 records per wave.
	uint64_t const groups = (n + 63) / 64;
	uint64_t const blocks = (groups + 3) / 4;
	return (uint32_t) (blocks < 1 ? 1 : (blocks > 256 * 8 ? 256 * 8 : blocks));
}

}	// anonymous namespace

uint32_t project_wave_count(uint64_t n) { return 4 * grid_blocks(n); }

int launch_project(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, uint32_t use_fast_path, void *stream)
{
	if (0 == in.n) return 0;
	hipLaunchKernelGGL(k_project, dim3(grid_blocks(in.n)), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work, use_fast_path);
	return (int) hipGetLastError();
}

}
