// wave.hpp — wavefront (64-lane) primitives for gfx950 and the gap bit-vector window.
#pragma once

#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"

namespace pvc3 {

__device__ __forceinline__ uint32_t lane_id()
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Number of set bits of m below the calling lane (v_mbcnt_lo/hi): the wavefront prefix popcount.
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}

__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
	return ((uint64_t) rfl((uint32_t) (v >> 32)) << 32) | rfl((uint32_t) v);
}

template <typename t_ptr>
__device__ __forceinline__ t_ptr *rfl_ptr(t_ptr *p) { return (t_ptr *) rfl64((uint64_t) p); }

__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t lane)
{
	return (uint32_t) __builtin_amdgcn_readlane((int) v, (int) lane);
}

// Replaces lane `lane` (wave-uniform) of a VGPR: v_cmp + v_cndmask (clang has no writelane builtin and v_writelane_b32
// with two SGPR operands violates the gfx9 constant-bus limit).
__device__ __forceinline__ uint32_t writelane(uint32_t old, uint32_t v, uint32_t lane)
{
	return (lane_id() == lane) ? v : old;
}

__device__ __forceinline__ uint32_t popc64(uint64_t v) { return (uint32_t) __builtin_popcountll(v); }
__device__ __forceinline__ uint32_t ctz64(uint64_t v) { return (uint32_t) __builtin_ctzll(v); }

// Bits [0, n) set; n in [0, 64].
__device__ __forceinline__ uint64_t low_mask(uint32_t n) { return (n >= 64) ? ~UINT64_C(0) : ((UINT64_C(1) << n) - 1); }

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

// Position of the t-th (0-based) set bit of the wave-uniform word z; all 64 lanes cooperate.
__device__ __forceinline__ uint32_t wave_select_in_word(uint64_t z, uint32_t t, uint32_t lane)
{
	bool const mine = (0 != ((z >> lane) & 1)) & (mbcnt64(z) == t);		// '&', not '&&': no lane-dependent branch
	return ctz64(ballot(mine));
}

// Same, one thread on its own word (6 halving steps).
__device__ __forceinline__ uint32_t thread_select_in_word(uint64_t z, uint32_t t)
{
	uint32_t x = (uint32_t) z, base = 0;
	uint32_t c = (uint32_t) __builtin_popcount(x);
	if (t >= c) { t -= c; x = (uint32_t) (z >> 32); base = 32; }
	c = (uint32_t) __builtin_popcount(x & 0xffffu); if (t >= c) { t -= c; x >>= 16; base += 16; }
	c = (uint32_t) __builtin_popcount(x & 0xffu); if (t >= c) { t -= c; x >>= 8; base += 8; }
	c = (uint32_t) __builtin_popcount(x & 0xfu); if (t >= c) { t -= c; x >>= 4; base += 4; }
	c = (uint32_t) __builtin_popcount(x & 0x3u); if (t >= c) { t -= c; x >>= 2; base += 2; }
	c = x & 1u; if (t >= c) { base += 1; }
	return base;
}

// ---- two-block window over the src and dst bit-vectors of one record ---------------------------------------------
// Lane l < 16 holds src physical word 8 b0 + l, lane 16 + l holds dst physical word 8 b0 + l (l < 16): one vector
// load instruction fetches four 64-byte blocks {rank, 7 x 64 bits}; words are then read as wave-uniform scalars.
struct window
{
	uint32_t lo, hi;
	uint32_t b0;	// wave-uniform
};

__device__ __forceinline__ void window_load(window &w, uint64_t const *src_blocks, uint64_t const *dst_blocks, uint32_t b0, uint32_t lane)
{
	// Branch-free: lanes 32..63 repeat the addresses of lanes 0..31 (same cache lines, no extra traffic).  A lane-dependent
	// branch around the load would be merged with the surrounding wave-uniform control flow and make it look divergent.
	uint64_t const *p = (lane & 16) ? dst_blocks : src_blocks;
	uint64_t const v = p[(uint64_t) b0 * BLOCK_WORDS + (lane & 15)];
	w.lo = (uint32_t) v;
	w.hi = (uint32_t) (v >> 32);
	w.b0 = b0;
}

__device__ __forceinline__ uint64_t window_word(window const &w, uint32_t idx)
{
	return ((uint64_t) readlane(w.hi, idx) << 32) | readlane(w.lo, idx);
}

struct col_pos
{
	uint32_t blk, w, bit;
};

__device__ __forceinline__ col_pos split_col(uint32_t col)
{
	col_pos p;
	p.blk = col / BLOCK_COLS;
	uint32_t const r = col - p.blk * BLOCK_COLS;
	p.w = r >> 6;
	p.bit = r & 63;
	return p;
}

// rank0 of the dst vector at column (blk, w, bit); the window must cover blk.
__device__ __forceinline__ uint32_t window_rank_dst(window const &win, col_pos const p)
{
	uint32_t const base = 16 + (p.blk - win.b0) * BLOCK_WORDS;
	uint32_t r = readlane(win.lo, base);
	for (uint32_t i = 0; i < p.w; ++i)
		r += popc64(~window_word(win, base + 1 + i));
	r += popc64(~window_word(win, base + 1 + p.w) & low_mask(p.bit));
	return r;
}

// Column of non-gap number k (0-based) of the src vector, i.e. select0(k + 1) in sdsl terms (msa_index.hh:239-243).
// Requires k < n_zeros.  Leaves `win` covering the block of the result.  (Cooperative form, used by the general walk only
// on its rare spill path; the per-record start column comes from the lane-parallel thread_select0.)
__device__ __forceinline__ uint32_t wave_select0(dev_entry const &S, uint64_t const *dst_blocks, uint32_t k, window &win, uint32_t lane)
{
	uint32_t const col0 = S.sel[k >> SELECT_SAMPLE_SHIFT];
	// The target lies at least (k mod SELECT_SAMPLE) columns right of the sample: its block cannot be left of this one.
	col_pos p;
	p.blk = (col0 + (k & (SELECT_SAMPLE - 1))) / BLOCK_COLS; p.w = 0; p.bit = 0;
	window_load(win, S.blocks, dst_blocks, p.blk, lane);
	uint32_t rem = k - readlane(win.lo, 0);
	for (;;)
	{
		if (p.blk - win.b0 >= 2)
		{
			// Entering a block outside the window: gallop over all-gap stretches with the rank words.
			uint32_t blk = p.blk;
			for (;;)
			{
				uint32_t probe = blk + 1 + lane;
				if (probe > S.n_blocks) probe = S.n_blocks;		// padding block: rank = n_zeros > k
				uint64_t const m = ballot((uint32_t) S.blocks[(uint64_t) probe * BLOCK_WORDS] <= k);
				uint32_t const skip = popc64(m);
				blk += skip;
				if (skip < 64) break;
			}
			p.blk = blk; p.w = 0; p.bit = 0;
			window_load(win, S.blocks, dst_blocks, blk, lane);
			rem = k - readlane(win.lo, 0);
		}
		uint64_t const sw = window_word(win, (p.blk - win.b0) * BLOCK_WORDS + 1 + p.w);
		uint64_t const z = ~sw & (~UINT64_C(0) << p.bit);
		uint32_t const cnt = popc64(z);
		if (rem < cnt)
			return p.blk * BLOCK_COLS + p.w * 64 + wave_select_in_word(z, rem, lane);
		rem -= cnt;
		p.bit = 0;
		if (++p.w == BLOCK_DATA_WORDS) { p.w = 0; ++p.blk; }
	}
}

// ---- one lane, directory form: word 0 of a block + ONE of its 16-byte pairs ----------------------------------------------
// Word 0 carries the zeros before the block and before its pairs 1, 2, 3 (pvc3_internal.hpp); pair p holds data words
// 2 p - 1 and 2 p (pair 0: word 0 itself and data word 0).  A rank or a select inside the block therefore needs 24 bytes in
// 6 registers and two popcounts instead of the whole block: k_project_fast keeps four probes in flight per record.
// A row is seen through a row_view: the plain array (blocks 64 bytes apart, block b = columns [448 b, 448 b + 448)) or the
// row's half of a paired copy (dev_entry::paired: 128 bytes apart, block b = columns [PAIRED_STEP b, PAIRED_STEP b + 448)).
typedef unsigned long long __attribute__((ext_vector_type(2))) dir_pair_t;

struct row_view
{
	uint64_t const *blocks;
	uint32_t stride;		// 64-bit words from one block to the next
	uint32_t step;			// columns from one block's first column to the next block's
};

// The last block that starts at or before `col`.
__device__ __forceinline__ col_pos locate(uint32_t step, uint32_t col)
{
	col_pos p;
	p.blk = (PAIRED_STEP == step) ? (col / PAIRED_STEP) : (col / BLOCK_COLS);
	uint32_t const r = col - p.blk * step;
	p.w = r >> 6; p.bit = r & 63;
	return p;
}

// `col` inside block `blk` if that block covers it (overlapping blocks: the end of a span in the block of its start).
__device__ __forceinline__ col_pos locate_in(uint32_t step, uint32_t col, uint32_t blk)
{
	uint32_t const r = col - blk * step;		// wraps to a large number when col lies before the block
	if (r >= BLOCK_COLS)
		return locate(step, col);
	col_pos p;
	p.blk = blk; p.w = r >> 6; p.bit = r & 63;
	return p;
}

struct dir_probe
{
	uint64_t head, first, second;
	uint32_t blk, pair;
};

__device__ __forceinline__ uint32_t pair_of(col_pos const q) { return (q.w + 1) >> 1; }

__device__ __forceinline__ void dir_load_pair(dir_probe &p, row_view const &v, uint32_t pair)
{
	dir_pair_t const x = *(dir_pair_t const *) (v.blocks + (uint64_t) p.blk * v.stride + 2 * pair);
	p.first = x.x; p.second = x.y; p.pair = pair;
}

__device__ __forceinline__ dir_probe dir_load(row_view const &v, uint32_t blk, uint32_t pair)
{
	dir_probe p;
	p.blk = blk;
	p.head = v.blocks[(uint64_t) blk * v.stride];
	dir_load_pair(p, v, pair);
	return p;
}

// Zeros of the block before pair `pair`.
__device__ __forceinline__ uint32_t dir_before(uint64_t head, uint32_t pair)
{
	uint32_t const hi = (uint32_t) (head >> DIR_SHIFT);
	return pair ? ((hi >> (DIR_BITS * (pair - 1))) & ((1u << DIR_BITS) - 1)) : 0;
}

// rank0 at column q; the probe is reloaded when it was fetched for another place (a guess that did not hold).
__device__ __forceinline__ uint32_t dir_rank(row_view const &v, dir_probe p, col_pos const q)
{
	uint32_t const pair = pair_of(q);
	if (p.blk != q.blk || p.pair != pair)
		p = dir_load(v, q.blk, pair);
	bool const is_second = 0 == (q.w & 1);		// data word w is the second word of its pair iff w is even
	uint64_t const z_first = pair ? ~p.first : UINT64_C(0);		// the first word of pair 0 is word 0, not data
	uint64_t const word = is_second ? p.second : p.first;
	uint32_t r = (uint32_t) p.head + dir_before(p.head, pair);
	r += is_second ? popc64(z_first) : 0;
	r += popc64(~word & low_mask(q.bit));
	return r;
}

// Column of non-gap number pos (0-based, < n_zeros).  `p` is a probe of any block, normally the block and pair of the select
// samples' guess.
__device__ __forceinline__ uint32_t dir_select0(row_view const &v, uint32_t pos, dir_probe p)
{
	for (;;)
	{
		if ((uint32_t) p.head > pos)
		{
			// The probe lies past the target (an interpolated guess can): head - pos non-gap columns, so at least that many
			// columns, separate the target from this block's first column.
			uint32_t const back = (uint32_t) p.head - pos, c = p.blk * v.step;
			p = dir_load(v, locate(v.step, (c > back) ? (c - back) : 0).blk, 3);
			continue;
		}
		uint32_t const rem = pos - (uint32_t) p.head;
		uint32_t const hi = (uint32_t) (p.head >> DIR_SHIFT), fm = (1u << DIR_BITS) - 1;
		uint32_t const c1 = hi & fm, c3 = (hi >> DIR_BITS) & fm, c5 = (hi >> (2 * DIR_BITS)) & fm;
		// the last pair whose zeros-before do not exceed rem holds the target, unless it lies beyond the block
		uint32_t const pair = (rem >= c5) ? 3 : (rem >= c3) ? 2 : (rem >= c1) ? 1 : 0;
		uint32_t const before = (rem >= c5) ? c5 : (rem >= c3) ? c3 : (rem >= c1) ? c1 : 0;
		if (pair != p.pair)
			dir_load_pair(p, v, pair);
		uint32_t const r2 = rem - before;
		uint64_t const z1 = pair ? ~p.first : UINT64_C(0), z2 = ~p.second;
		uint32_t const n1 = popc64(z1), n2 = popc64(z2);
		if (r2 < n1 + n2)
		{
			bool const in_first = r2 < n1;
			return p.blk * v.step + 64 * (2 * pair - (in_first ? 1 : 0)) + thread_select_in_word(in_first ? z1 : z2, in_first ? r2 : (r2 - n1));
		}
		// Only pair 3 runs out: the target is r2 - n1 - n2 non-gap columns past the end of this block, at least that many columns.
		p = dir_load(v, locate(v.step, p.blk * v.step + BLOCK_COLS + (r2 - n1 - n2)).blk, 0);
	}
}

// The select samples' guess for non-gap number pos.  Sample j is the column of non-gap number SELECT_SAMPLE * j; the target
// lies pos % SELECT_SAMPLE non-gap columns further.  Where the stretch up to the next sample has few gap columns the guess is
// that many columns further (exact with no gap in between, never past the target); where it has many (more than one column
// in 16: a gap-dense MSA) the gaps are taken to be spread evenly, which lands within tens of columns instead of hundreds.
// `span` columns' worth of non-gap columns after the target are guessed to end at `end`.
struct select_guess_t
{
	uint32_t col, end;
};

__device__ __forceinline__ select_guess_t select_guess(uint32_t const *__restrict__ sel, uint32_t pos, uint32_t span)
{
	uint32_t const j = pos >> SELECT_SAMPLE_SHIFT, rem = pos & (SELECT_SAMPLE - 1);
	uint32_t const s0 = sel[j], s1 = sel[j + 1];		// the table ends with a sentinel (the number of columns)
	uint32_t const width = s1 - s0;
	bool const dense = width > SELECT_SAMPLE + SELECT_SAMPLE / 16;
	select_guess_t g;
	g.col = s0 + (dense ? (uint32_t) (((uint64_t) rem * width) >> SELECT_SAMPLE_SHIFT) : rem);
	g.end = g.col + (dense ? (uint32_t) (((uint64_t) span * width) >> SELECT_SAMPLE_SHIFT) : span);
	return g;
}

// The rows of a projection from entry S to entry number dst_id (= D): the paired copy when D is S's partner.
struct row_pair
{
	row_view src, dst;
};

__device__ __forceinline__ row_pair rows_of(dev_entry const &S, dev_entry const &D, uint32_t dst_id)
{
	bool const paired = S.paired && S.partner == dst_id;
	uint32_t const stride = paired ? 2 * BLOCK_WORDS : BLOCK_WORDS, step = paired ? PAIRED_STEP : BLOCK_COLS;
	return row_pair{row_view{paired ? S.paired : S.blocks, stride, step}, row_view{paired ? S.paired + BLOCK_WORDS : D.blocks, stride, step}};
}

// Start column of source position pos and the destination's rank there, from the select samples (the look-up of
// thread_project_position, both results).  pos < S.n_zeros.
__device__ __forceinline__ void thread_locate(dev_entry const &S, dev_entry const &D, uint32_t dst_id, uint32_t pos, uint32_t &col, uint32_t &r)
{
	row_pair const R = rows_of(S, D, dst_id);
	col_pos const g = locate(R.src.step, select_guess(S.sel, pos, 0).col);
	dir_probe const ps = dir_load(R.src, g.blk, pair_of(g));
	dir_probe const pd = dir_load(R.dst, g.blk, pair_of(g));
	col = dir_select0(R.src, pos, ps);
	r = dir_rank(R.dst, pd, locate_in(R.dst.step, col, g.blk));
}

// From an anchor — a column `col` and the destination's rank `r` there — to the source's non-gap column number `skip`
// counted from col on (0: the first one at or after col): its column and the destination's rank there.  One pair of gap
// words per 64 columns; false when the target lies more than ADVANCE_WORDS words ahead (a long gap run: the caller looks
// it up from the select samples instead).  This is how a position window serves a record whose window is not clean up to
// its start (k_project_fast): the walk kernels finish the look-up from the window's last clean column.
constexpr uint32_t ADVANCE_WORDS = 6;
__device__ __forceinline__ bool advance_anchor(row_pair const &R, uint32_t &col, uint32_t &r, uint32_t skip)
{
	for (uint32_t it = 0; it < ADVANCE_WORDS; ++it)
	{
		col_pos const p = locate(R.src.step, col);
		uint64_t const wi = (uint64_t) p.blk * R.src.stride + 1 + p.w;
		uint64_t const sw = R.src.blocks[wi], dw = R.dst.blocks[wi];
		uint64_t const ge = ~UINT64_C(0) << p.bit;
		uint64_t const z = ~sw & ge;
		uint32_t const nz = popc64(z);
		if (skip < nz)
		{
			uint32_t const e = thread_select_in_word(z, skip);
			r += popc64(~dw & ge & low_mask(e));
			col = col - p.bit + e;
			return true;
		}
		skip -= nz;
		r += popc64(~dw & ge);
		col = col - p.bit + 64;
	}
	return false;
}

// What k_project_fast leaves of a record it does not finish: (col, r, skip).  skip == 0: col is the record's start column
// and r its projected position.  skip == PWIN_NO_ANCHOR: nothing is known (no position window for this pair of entries).
// Otherwise (col, r) is an anchor — the first column after the clean stretch of the record's window — and the start
// column is the source's non-gap column number skip - 1 from there on.  Returns true when (col, r) changed.
__device__ __forceinline__ bool resolve_start(dev_entry const &S, dev_entry const &D, uint32_t dst_id, uint32_t src_pos, uint32_t &col, uint32_t &r, uint32_t skip)
{
	if (0 == skip)
		return false;
	if (PWIN_NO_ANCHOR != skip)
	{
		uint32_t c = col, rr = r;
		if (advance_anchor(rows_of(S, D, dst_id), c, rr, skip - 1))
		{
			col = c; r = rr;
			return true;
		}
	}
	thread_locate(S, D, dst_id, src_pos, col, r);
	return true;
}

__device__ __forceinline__ uint32_t thread_project_position(dev_entry const &S, dev_entry const &D, uint32_t dst_id, uint32_t pos)
{
	if (pos >= S.n_zeros || S.n_cols != D.n_cols)
		return PVC3_NO_MATE;
	row_pair const R = rows_of(S, D, dst_id);
	// Both probes at the sample's guess: the destination's is right whenever no gap lies between the sample and the target.
	col_pos const g = locate(R.src.step, select_guess(S.sel, pos, 0).col);
	dir_probe const ps = dir_load(R.src, g.blk, pair_of(g));
	dir_probe const pd = dir_load(R.dst, g.blk, pair_of(g));
	return dir_rank(R.dst, pd, locate_in(R.dst.step, dir_select0(R.src, pos, ps), g.blk));
}

}
