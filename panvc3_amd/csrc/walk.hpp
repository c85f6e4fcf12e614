// walk.hpp — the general projection walk: lane = MSA column, one record at a time per wavefront.
//
// Restates rewrite_cigar (include/panvc3/rewrite_cigar.hh:18-157) through the column-walk equivalence,
// with cigar_buffer semantics (libpanvc3/cigar.cc:54-95) and the run detection of indel_run_checker
// (libpanvc3/indel_run_checker.cc:31-121).  The wave walks 64 columns per step over the src / dst gap words held in a
// two-block register window, builds the = X I D token masks with ballots and run-length merges them with scalar bit
// arithmetic.  Used for every record that the lane-parallel fast path (project_fast.hip) cannot take.
#pragma once

#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {


// Emits the tokens of one 64-column step in column order (bit i of m? = column i of the step).
__device__ __forceinline__ void emit_tokens(sink &s, uint64_t mE, uint64_t mX, uint64_t mI, uint64_t mD, uint32_t lane)
{
	mE = rfl64(mE); mX = rfl64(mX); mI = rfl64(mI); mD = rfl64(mD);
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
	// Two table lookups in registers (a chain of selects is turned into a branch tree by the compiler):
	// code -> index into "ACGTN" (1 2 4 8 -> 0 1 2 3, everything else -> 4), then the character.
	uint32_t const idx = (uint32_t) (UINT64_C(0x4444444344424104) >> (4 * code)) & 0xf;
	return (uint32_t) (UINT64_C(0x4E54474341) >> (8 * idx)) & 0xff;
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
	uint32_t status, n_out, has_run, cols, ref_span, l_seq;
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
	// Unconditional 256-byte reads: the SEQ and CIGAR buffers are padded by 256 bytes, device texts by TEXT_PAD bytes.
	// (No lane-dependent branches here: they would be merged into the wave-uniform control flow around them.)
	L.seqw = load_u32_unaligned(C.seq4 + 4 * lane);
	if (C.d_text)
		L.refw = *(uint32_t const *) (C.d_text + C.ref_base + 4 * lane);
	L.cigv = C.cigar[lane];
}

// Phase B for one record: the column walk.  All control flow is wave-uniform.
__device__ __forceinline__ void walk_record(
	dev_entry const *__restrict__ entries, rec_ctx const &C_in, rec_loads const &L, dev_batch_work const &W, arena_alloc &arena, uint32_t lane, rec_out &res
)
{
	res.status = rfl(C_in.status); res.n_out = 0; res.has_run = 0; res.loc = 0; res.cols = 0; res.ref_span = 0; res.l_seq = 0;
	if (res.status)
		return;

	// Everything in `C` is wave-uniform by construction (readlane / scalar loads).  Re-asserting it here cuts the cycles
	// through which a single conservative "divergent" verdict of the compiler would spread to the whole walk.
	rec_ctx C;
	C.status = 0;
	C.col = rfl(C_in.col); C.r0 = rfl(C_in.r0); C.ref_consumed = rfl(C_in.ref_consumed); C.n_cigar = rfl(C_in.n_cigar);
	C.l_seq = rfl(C_in.l_seq); C.src_e = rfl(C_in.src_e); C.src_pos = rfl(C_in.src_pos); C.d_text_len = rfl(C_in.d_text_len);
	C.b0 = rfl(C_in.b0); C.ref_base = rfl(C_in.ref_base);
	C.s_blocks = rfl_ptr(C_in.s_blocks); C.d_blocks = rfl_ptr(C_in.d_blocks); C.d_text = rfl_ptr(C_in.d_text);
	C.cigar = rfl_ptr(C_in.cigar); C.seq4 = rfl_ptr(C_in.seq4);

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
			cigv = C.cigar[cig_base + lane];		// padded buffer: unconditional
		}
		uint32_t const c = readlane(cigv, rfl(ci - cig_base));
		uint32_t const op = c & 0xf;
		uint32_t n_rem = c >> 4;
		q = rfl(q);

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
			// Every value below is wave-uniform by construction; saying so keeps the loop-carried state in SGPRs.
			col = rfl(col); r = rfl(r); q = rfl(q); n_rem = rfl(n_rem); ref_base = rfl(ref_base); win.b0 = rfl(win.b0);
			out.make_uniform();
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
			{
				window_load(win, C.s_blocks, C.d_blocks, p.blk, lane);
				win.b0 = rfl(win.b0);
			}
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
					mE = ballot(in_c & (seq4_to_dna5_char(code) == rb));		// raw byte compare (quirk Q3)
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
	res.l_seq = C.l_seq;
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


// Per-group staging of the per-record outputs (lane = record).
struct group_out
{
	uint32_t flags, cig_n, loc_lo, loc_hi;
	uint32_t n_failed, n_with_runs;		// wave-uniform accumulators
	uint64_t bitvec_bytes, ref_bytes, bytes;

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
		bytes += ok ? (2 * ((r.cols + 8) / 8) + r.ref_span + 4 * r.n_out + (r.l_seq + 1) / 2) : 0;	// gap bits, text, CIGAR out, SEQ
	}
};


}
