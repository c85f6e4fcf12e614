// project_mid.hip — k_project_mid: lane = record for the records the fast path of k_project_fast could not take.
//
// k_project_fast leaves such records (CIGAR with insertions / deletions, or a gap column of either row inside the window)
// in a per-wave work list.  Here 64 of them are walked at a time, one per lane, with exactly the steps of the cooperative
// column walk (walk.hpp: rewrite_cigar, include/panvc3/rewrite_cigar.hh:18-157; cigar_buffer, libpanvc3/cigar.cc:54-95;
// run detection, libpanvc3/indel_run_checker.cc:31-121) but on the lane's own registers: the gap words come from plain
// loads, the = / X split from the packed text (compare.hpp) over the runs of compared columns, the output operations go to
// a lane-interleaved LDS buffer and are copied to the arena once their number is known.  A record that needs more than
// MID_TOKENS output operations or more than MID_MAX_STEPS steps stays marked REC_SLOW for k_project_general.
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"
#include "compare.hpp"

namespace pvc3 {

namespace {

constexpr uint32_t MID_TOKENS = 32;			// output operations buffered per lane
constexpr uint32_t MID_MAX_STEPS = 40;		// 64-column steps per record

// cigar_buffer + indel_run_checker state of one lane (device_common.hpp `sink`, without the wave-uniform plumbing).
struct lane_sink
{
	uint32_t cur_op, cur_cnt, n_out, run_type, has_run;
	uint32_t *tok;		// LDS, operation k at tok[64 * k]

	__device__ __forceinline__ void init(uint32_t *t) { cur_op = 0xf; cur_cnt = 0; n_out = 0; run_type = 0; has_run = 0; tok = t; }

	__device__ __forceinline__ void flush()
	{
		if (n_out < MID_TOKENS) tok[64 * n_out] = (cur_cnt << 4) | cur_op;
		++n_out;		// n_out > MID_TOKENS tells the caller that the record does not fit
		if (OP_I == cur_op) run_type |= 1;
		else if (OP_D == cur_op) run_type |= 2;
		else { has_run |= (3 == run_type); run_type = 0; }
	}

	__device__ __forceinline__ void push(uint32_t op, uint32_t cnt)
	{
		if (0 == cnt) return;
		if (cur_op == op) cur_cnt += cnt;
		else
		{
			if (cur_cnt) flush();
			cur_op = op; cur_cnt = cnt;
		}
	}

	__device__ __forceinline__ void finish()
	{
		if (cur_cnt) flush();
		has_run |= (3 == run_type);
	}
};

// Tokens of one step in column order (walk.hpp emit_tokens).
__device__ __forceinline__ void emit_tokens_lane(lane_sink &s, uint64_t mE, uint64_t mX, uint64_t mI, uint64_t mD)
{
	uint64_t valid = mE | mX | mI | mD;
	while (valid)
	{
		uint64_t const pb = valid & (~valid + 1);
		uint32_t op; uint64_t mt;
		if (mE & pb) { op = OP_EQ; mt = mE; }
		else if (mX & pb) { op = OP_X; mt = mX; }
		else if (mI & pb) { op = OP_I; mt = mI; }
		else { op = OP_D; mt = mD; }
		uint64_t const others = valid & ~mt;
		uint64_t const below = others ? ((others & (~others + 1)) - 1) : ~UINT64_C(0);
		s.push(op, popc64(mt & valid & below));
		valid &= ~below;
	}
}

// 64 + 16 consecutive 4-bit codes starting at nibble `first` of a byte string, as five independent 8-byte loads.
struct nibbles80
{
	uint64_t w[5];
};

template <bool t_query>
__device__ __forceinline__ nibbles80 load_nibbles80(uint8_t const *bytes, uint64_t first)
{
	// (From the byte address itself: on the configurations that come here — one long run per step — the aligned form of
	// load_win below measured 3 % slower, for its ten more shifts.)
	uint8_t const *p = bytes + (first >> 1);
	u64x2 const a = load_u128_unaligned(p), b = load_u128_unaligned(p + 16);
	uint64_t c;
	__builtin_memcpy(&c, p + 32, 8);
	nibbles80 n;
	n.w[0] = a.x; n.w[1] = a.y; n.w[2] = b.x; n.w[3] = b.y; n.w[4] = c;
	if (t_query)		// SEQ stores the first base of a byte in the high nibble
	{
#pragma unroll
		for (uint32_t i = 0; i < 5; ++i) n.w[i] = nibble_linear(n.w[i]);
	}
	if (first & 1)
	{
#pragma unroll
		for (uint32_t i = 0; i < 4; ++i) n.w[i] = (n.w[i] >> 4) | (n.w[i + 1] << 60);
	}
	return n;
}

// Mismatch bits of the compared columns mC of one step.  Inside a run of consecutive compared columns the query and the
// text both advance by one base per column, so a run is a straight comparison of two nibble strings; all the bytes a run
// can need (64 bases of either side) are requested at once, so that a run costs one memory round trip.
__device__ __forceinline__ uint64_t compare_columns(uint64_t mC, uint64_t Z, uint64_t dz, uint32_t q, uint32_t r, uint8_t const *seq, uint8_t const *text4)
{
	uint64_t mX = 0, left = mC;
	while (left)
	{
		uint32_t const c0 = ctz64(left);
		uint64_t const ones = ~(left >> c0);
		uint32_t const L = ones ? ctz64(ones) : 64;
		uint32_t const qi = q + popc64(Z & low_mask(c0)), ri = r + popc64(dz & low_mask(c0));
		nibbles80 const qn = load_nibbles80 <true>(seq, qi);
		nibbles80 const tn = load_nibbles80 <false>(text4, (uint64_t) ri + TEXT4_FRONT);
		uint64_t mm = 0;
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k)
			mm |= (uint64_t) mismatch16(qn.w[k], tn.w[k]) << (16 * k);
		mX |= (mm & low_mask(L)) << c0;
		left &= ~(low_mask(L) << c0);
	}
	return mX;
}

// ---- the compare from register windows --------------------------------------------------------------------------------------
// A 64-column step consumes at most 64 query bases from q on and 64 text bases from r on, whatever its runs of compared
// columns look like.  So the 80 nibbles of either side from the step's start are fetched ONE STEP AHEAD (the addresses
// depend only on q and r, known when the previous step ends), and every run takes its 16-nibble pieces out of the two
// register windows: no load depends on the step's own masks.
struct win80
{
	uint64_t w[5];
	uint32_t first;		// nibble number (in the byte string) of the window's first nibble: even
};

template <bool t_query>
__device__ __forceinline__ win80 load_win(uint8_t const *bytes, uint64_t first)
{
	win80 n;
	load_bytes40(bytes + (first >> 1), n.w);
	if (t_query)		// SEQ stores the first base of a byte in the high nibble
	{
#pragma unroll
		for (uint32_t i = 0; i < 5; ++i) n.w[i] = nibble_linear(n.w[i]);
	}
	n.first = (uint32_t) first & ~1u;
	return n;
}

// 32 nibbles from nibble offset o (0 .. 79) of the window as two words of 16; what lies past the window reads as 0.
__device__ __forceinline__ void win_extract32(win80 const &W, uint32_t o, uint64_t &first, uint64_t &second)
{
	uint32_t const wi = o >> 4, sh = (o & 15) * 4;
	uint64_t a = W.w[0], b = W.w[1], c = W.w[2];
	a = (1 == wi) ? W.w[1] : a; b = (1 == wi) ? W.w[2] : b; c = (1 == wi) ? W.w[3] : c;
	a = (2 == wi) ? W.w[2] : a; b = (2 == wi) ? W.w[3] : b; c = (2 == wi) ? W.w[4] : c;
	a = (3 == wi) ? W.w[3] : a; b = (3 == wi) ? W.w[4] : b; c = (3 <= wi) ? UINT64_C(0) : c;
	a = (4 == wi) ? W.w[4] : a; b = (4 <= wi) ? UINT64_C(0) : b;
	first = sh ? ((a >> sh) | (b << (64 - sh))) : a;
	second = sh ? ((b >> sh) | (c << (64 - sh))) : b;
}

// Mismatch bits of the compared columns mC of one step, up to 32 columns of a run at a time from the windows (the runs of a
// gap-dense MSA are some twenty columns long: one extraction, two 16-base compares).  q_off / r_off: nibble offsets in the
// windows of the step's first query base / first text base.  (Moving a whole window into place once per run — four
// conditional dword stages and eight funnel shifts — and comparing 64 bases from it costs 129 VGPRs, three waves per SIMD:
// slower on every configuration.)
__device__ __forceinline__ uint64_t compare_columns_windows(uint64_t mC, uint64_t Z, uint64_t dz, uint32_t q_off, uint32_t r_off, win80 const &Q, win80 const &T)
{
	uint64_t mX = 0, left = mC;
	while (left)
	{
		uint32_t const c0 = ctz64(left);
		uint64_t const ones = ~(left >> c0);
		uint32_t const run = ones ? ctz64(ones) : 64;
		uint32_t const L = (run < 32) ? run : 32;
		uint32_t const qo = q_off + popc64(Z & low_mask(c0)), ro = r_off + popc64(dz & low_mask(c0));
		uint64_t q1, q2, t1, t2;
		win_extract32(Q, qo, q1, q2);
		win_extract32(T, ro, t1, t2);
		uint32_t mm = mismatch16(q1, t1);
		if (L > 16) mm |= mismatch16(q2, t2) << 16;
		mX |= ((uint64_t) mm & low_mask(L)) << c0;
		left &= ~(low_mask(L) << c0);
	}
	return mX;
}

__global__ __launch_bounds__(256, 4) void k_project_mid(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	// Same partition as k_project_fast (same grid): wave w wrote its list at the start of its own record range.
	uint64_t const n_groups = (in.n + 63) / 64;
	slow_item const *list = W.slow_items + 64 * (n_groups * wave / n_waves);
	uint32_t const *rec_list = W.slow_list + 64 * (n_groups * wave / n_waves);
	uint32_t const count = W.slow_count[wave] & ~SLOW_ITEMS;
	bool const have_items = 0 != (W.slow_count[wave] & SLOW_ITEMS);
	uint32_t *run_list = W.run_list + 64 * (n_groups * wave / n_waves);
	uint32_t n_run = 0;		// wave-uniform
	// The records this kernel leaves to k_project_general are written back to the front of the same list (never ahead of the
	// entries still to be read: at most as many as have been read).
	uint32_t *punt_list = W.slow_list + 64 * (n_groups * wave / n_waves);
	uint32_t n_punt = 0;	// wave-uniform
	__shared__ uint32_t lds_tok[4][64 * MID_TOKENS];
	uint32_t *my_tok = lds_tok[rfl(threadIdx.x >> 6)] + lane;
	arena_alloc arena{0, 0};
	uint32_t acc_failed = 0, acc_runs = 0, acc_done = 0;
	uint64_t acc_bitvec = 0, acc_ref = 0, acc_bytes = 0;

	for (uint32_t base = 0; base < count; base += 64)
	{
		bool const mine = base + lane < count;
		slow_item item{};
		if (mine && have_items) item = list[base + lane];
		else if (mine)
		{
			// a dense list: the record numbers are nearly consecutive and the arrays are read as streams
			uint32_t const rr = rec_list[base + lane];
			item.rec = rr; item.col0 = (uint32_t) W.cig_loc[rr]; item.r0 = W.dst_pos[rr]; item.skip = W.cig_n[rr];
			item.src_e = in.src_entry[rr]; item.dst_e = in.dst_entry[rr];
			item.cig_begin = in.cigar_off[rr]; item.n_cigar = (uint32_t) (in.cigar_off[rr + 1] - item.cig_begin);
			item.seq_begin = in.seq_off[rr]; item.l_seq = in.l_seq[rr];
		}
		uint32_t const rec = item.rec;
		uint32_t status = PVC3_ST_OK;
		bool punt = false;
		lane_sink out;
		out.init(my_tok);
		uint32_t cols = 0, ref_span = 0, l_seq = 0, col_start = 0;
		bool start_moved = false;

		// Which form of the compare: by the gap density of the rows these records lie on (more than one gap column in 16 in
		// either row means runs of compared columns of some twenty bases), decided for the wave.
		bool dense = false;
		if (mine)
		{
			dev_entry const *sp = entries + item.src_e, *dp = entries + item.dst_e;
			dense = (16 * (sp->n_cols - sp->n_zeros) > sp->n_cols) || (16 * (dp->n_cols - dp->n_zeros) > dp->n_cols);
		}
		bool const short_runs = 2 * popc64(ballot(dense)) > popc64(ballot(mine));

		if (mine)
		{
			dev_entry const S = entries[item.src_e];
			dev_entry const D = entries[item.dst_e];
			uint32_t const n_cigar = item.n_cigar;
			uint32_t const *ops = in.cigar + item.cig_begin;
			uint8_t const *seq = in.seq4 + item.seq_begin;
			l_seq = item.l_seq;
			// Start column and projected position: from k_project_fast, or finished here from what its position window knew.
			uint32_t col0 = item.col0, r0 = item.r0;
			if (resolve_start(S, D, item.dst_e, item.skip ? in.src_pos[rec] : 0, col0, r0, item.skip))
			{
				W.dst_pos[rec] = r0;
				start_moved = true;
				// The neighbouring record may be this record's mate, waiting for this position (k_project_fast).
				uint64_t const nb = (uint64_t) rec ^ 1;
				if (nb < in.n && MATE_FROM_NEIGHBOUR == W.mate_dst_pos[nb]) W.mate_dst_pos[nb] = r0;
			}
			uint32_t col = col0, r = r0, q = 0, steps = 0;
			// The two rows' gap words, from the paired copy (one 128-byte line for both) when the destination is the source's
			// partner.  The words of the NEXT 64 columns are requested a step ahead: a step ends at a word boundary or inside
			// the same word, so the walk never waits for a gap word of its own.
			row_pair const rows = rows_of(S, D, item.dst_e);
			auto gap_words = [&](uint32_t first_col, uint64_t &sw_out, uint64_t &dw_out) {
				col_pos const pp = locate(rows.src.step, first_col);
				uint64_t const wi = (uint64_t) pp.blk * rows.src.stride + 1 + pp.w;
				sw_out = global_load(rows.src.blocks + wi); dw_out = global_load(rows.dst.blocks + wi);
			};
			uint32_t have_col = col0 & ~63u;			// first column of the words in (sw, dw); those of have_col + 64 are in (sw_next, dw_next)
			uint64_t sw = 0, dw = 0, sw_next = 0, dw_next = 0;
			gap_words(have_col, sw, dw);
			gap_words(have_col + 64, sw_next, dw_next);

			typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
			u32x4 pre;
			__builtin_memcpy(&pre, ops, 16);			// the CIGAR buffer is padded by the host side
			// One loop over 64-column steps, whatever operation a lane is in: a lane that finishes an operation takes its next
			// one in the same iteration.  (A loop over the operations with a loop over the steps inside makes the wave run the
			// inner loop once per operation INDEX: a record with a leading soft clip has its M steps one index later than the
			// others, and every lane then waits through both rounds.)
			uint32_t ci = 0, n_rem = 0;
			bool m_type = false;
			// Gap-dense rows (several short runs of compared columns per step): the compare works from register windows that
			// are fetched a step ahead.  Otherwise a step has one long run and a load of its own for it is cheaper.
			bool const windows = short_runs && nullptr != D.text4;
			win80 qwin{}, twin{};
			if (windows)
			{
				qwin = load_win <true>(seq, 0);
				twin = load_win <false>(D.text4, (uint64_t) r0 + TEXT4_FRONT);
			}
			for (;;)
			{
				while (0 == n_rem && ci < n_cigar)
				{
					uint32_t const c = (0 == ci) ? pre.x : (1 == ci) ? pre.y : (2 == ci) ? pre.z : (3 == ci) ? pre.w : ops[ci];
					++ci;
					uint32_t const op = c & 0xf, cnt = c >> 4;
					if (OP_I == op || OP_S == op) { out.push(op, cnt); q += cnt; }		// rewrite_cigar.hh:69-73
					else if (OP_H == op || OP_P == op) out.push(op, cnt);				// :75-78
					else { m_type = !(OP_D == op || OP_N == op); n_rem = cnt; }			// M = X (:80-118) vs D N (:121-147)
				}
				if (0 == n_rem) break;
				if (++steps > MID_MAX_STEPS) { punt = true; break; }
				if ((col & ~63u) != have_col)
				{
					have_col += 64;
					sw = sw_next; dw = dw_next;
					gap_words(have_col + 64, sw_next, dw_next);
				}
				col_pos p;
				p.bit = col & 63;
				uint64_t const ge = ~UINT64_C(0) << p.bit;
				uint64_t const z = ~sw & ge;				// source non-gap columns at or after col
				uint32_t const nz = popc64(z);
				uint64_t act;								// columns handled in this step
				uint32_t consumed;
				uint32_t const word_col = col - p.bit;
				if (nz >= n_rem)
				{
					uint32_t const e = thread_select_in_word(z, n_rem - 1);	// last base of this operation
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
						uint32_t const hi = 63 - (uint32_t) __builtin_clzll(mC);
						uint32_t const max_ri = r + popc64(dz & low_mask(hi));
						if (!D.text4 || max_ri >= D.text_len) { status = PVC3_ST_REF_RANGE; break; }
#if defined(PVC3_ABL_MID_NO_COMPARE)
						mX = 0;
#else
						if (windows)
						{
							// Insertions / soft clips since the window was fetched move q inside it; past 14 bases it is fetched anew.
							if (q - qwin.first > 14 + 1) qwin = load_win <true>(seq, q);
							mX = compare_columns_windows(mC, Z, dz, q - qwin.first, r + TEXT4_FRONT - twin.first, qwin, twin);
						}
						else
							mX = compare_columns(mC, Z, dz, q, r, seq, D.text4);
#endif
						mE = mC & ~mX;
					}
					q += consumed;
				}
				else
				{
					mI = 0;
					mD = dz;								// (:133-137) kept where dst has a character, dropped otherwise
				}
#ifndef PVC3_ABL_MID_NO_EMIT
				emit_tokens_lane(out, mE, mX, mI, mD);
#endif
				r += nd;
				n_rem -= consumed;
				if (windows)		// the windows of the next step
				{
					qwin = load_win <true>(seq, q);
					twin = load_win <false>(D.text4, (uint64_t) r + TEXT4_FRONT);
				}
			}
			out.finish();
			punt = punt || (out.n_out > MID_TOKENS);
			cols = col - col0;
			ref_span = r - r0;
			col_start = col0;
		}

		// ---- wave level: arena space for the finished records, copy, per-record outputs ------------------------------------------
		bool const done = mine && !punt;
		bool const ok = done && (PVC3_ST_OK == status);
		uint32_t n_tok = ok ? out.n_out : 0;
		uint32_t const scan = wave_inclusive_scan(n_tok, lane);
		uint32_t const total = readlane(scan, 63);
		uint64_t base_loc = 0;
		if (total && !arena_take(arena, total, W, lane, base_loc))
			n_tok = 0;		// arena overflow: the host grows it and repeats the batch
		uint64_t const loc = base_loc + (scan - (ok ? out.n_out : 0));
		for (uint32_t k = 0; k < n_tok; ++k)
			W.arena[loc + k] = my_tok[64 * k];
		{
			// What is left for k_project_general.  (The list is read 64 entries ahead of where it is written.)
			uint64_t const m = ballot(mine && punt);
			if ((m >> lane) & 1) punt_list[n_punt + mbcnt64(m)] = rec;
			n_punt += popc64(m);
			// k_project_general reads the start of a record from the work arrays: leave what was found here.
			if (mine && punt && start_moved)
			{
				*(uint32_t *) &W.cig_loc[rec] = col_start;		// low half: the column (the high half keeps the reference-consuming length)
				W.cig_n[rec] = 0;
			}
		}
		{
			// Work list for k_realign.
			uint64_t const m = ballot(done && ok && out.has_run);
			if ((m >> lane) & 1) run_list[n_run + mbcnt64(m)] = rec;
			n_run += popc64(m);
		}
		if (done)
		{
			W.rec_flags[rec] = status;		// the records with an indel run are in the work list for k_realign
			W.cig_n[rec] = n_tok;
			W.cig_loc[rec] = loc;
			if (!ok) W.dst_pos[rec] = 0;		// a rejected record has empty outputs
			acc_done += 1;
			acc_failed += !ok;
			acc_runs += (ok && out.has_run);
			if (ok)
			{
				acc_bitvec += 2 * ((cols + 8) / 8);
				acc_ref += (ref_span + 1) / 2;
				acc_bytes += 2 * ((cols + 8) / 8) + (ref_span + 1) / 2 + 4 * n_tok + (l_seq + 1) / 2;	// gap bits, text, CIGAR out, SEQ
			}
		}
	}

	W.run_count[wave] = n_run;		// wave-uniform stores
	W.slow_count[wave] = n_punt;
	for (int off = 32; off; off >>= 1)
	{
		acc_bitvec += __shfl_xor(acc_bitvec, off);
		acc_ref += __shfl_xor(acc_ref, off);
		acc_bytes += __shfl_xor(acc_bytes, off);
		acc_failed += __shfl_xor(acc_failed, off);
		acc_runs += __shfl_xor(acc_runs, off);
		acc_done += __shfl_xor(acc_done, off);
	}
	if (0 == lane)
	{
		unsigned long long *ws = W.wave_stats + ((uint64_t) 1 * n_waves + wave) * WAVE_STATS;
		ws[WS_BITVEC] = acc_bitvec;
		ws[WS_REF] = acc_ref;
		ws[WS_FAILED] = acc_failed;
		ws[WS_DONE] = acc_done;
		ws[WS_BYTES] = acc_bytes;
		ws[WS_WITH_RUNS] = acc_runs;
	}
}

}	// anonymous namespace

int launch_project_mid(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream)
{
	if (0 == in.n) return 0;
	// The grid must be that of k_project_fast: the work lists are per wave.
	hipLaunchKernelGGL(k_project_mid, dim3(project_wave_count(in.n) / 4), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work);
	return (int) hipGetLastError();
}

}
