/*
 * pvc3_oracle.c — CPU oracle for the project_alignments hot path (see pvc3_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker and the reported CPU baseline, never the product path.
 * A literal scalar restatement of the reference algorithm; citations are into /root/reference
 * (tsnorri/panvc3 @ 2024-10-16).  Build: make -C oracle   (gcc -O2 -ffp-contract=off).
 *
 * PARITY UNPINNED for the realignment DP's co-optimal tie-break and score type (seqan3 is absent
 * from the reference snapshot); everything else is pinned by the reference's TSV fixtures.
 */
#include "pvc3_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Index: plain gap bit-vectors (1 = gap) + per-word cumulative zero counts.
 * Stands in for sdsl::rrr_vector<15> + rank_0/select_0 supports (msa_index.hh:43-45); semantics only.
 * ---------------------------------------------------------------------------------------------- */

typedef struct
{
	uint64_t n_cols;
	uint64_t n_words;
	uint64_t n_zeros;
	uint64_t *bits;		/* bit (i & 63) of word (i >> 6) = 1 iff column i is a gap */
	uint64_t *cum;		/* cum[w] = #zeros in words [0, w); n_words + 1 entries */
	char const *text;	/* unaligned sequence (borrowed), may be NULL */
	uint64_t text_len;
} orc_entry;

struct orc_index
{
	orc_entry *entries;
	uint32_t n_entries;
	uint32_t cap;
};

orc_index *orc_index_new(void)
{
	return (orc_index *) calloc(1, sizeof(orc_index));
}

void orc_index_free(orc_index *idx)
{
	if (!idx) return;
	for (uint32_t i = 0; i < idx->n_entries; ++i)
	{
		free(idx->entries[i].bits);
		free(idx->entries[i].cum);
	}
	free(idx->entries);
	free(idx);
}

static orc_entry *add_entry(orc_index *idx, uint64_t n_cols)
{
	if (idx->n_entries == idx->cap)
	{
		idx->cap = idx->cap ? 2 * idx->cap : 16;
		idx->entries = (orc_entry *) realloc(idx->entries, idx->cap * sizeof(orc_entry));
	}
	orc_entry *e = &idx->entries[idx->n_entries++];
	memset(e, 0, sizeof(*e));
	e->n_cols = n_cols;
	e->n_words = (n_cols + 63) / 64;
	e->bits = (uint64_t *) calloc(e->n_words + 1, sizeof(uint64_t));
	e->cum = (uint64_t *) calloc(e->n_words + 2, sizeof(uint64_t));
	return e;
}

static void finish_entry(orc_entry *e)
{
	/* Columns past n_cols in the last word are neither gap nor non-gap; mark them as gaps so they are not zeros. */
	if (e->n_cols & 63)
		e->bits[e->n_words - 1] |= ~UINT64_C(0) << (e->n_cols & 63);
	uint64_t z = 0;
	for (uint64_t w = 0; w < e->n_words; ++w)
	{
		e->cum[w] = z;
		z += (uint64_t) __builtin_popcountll(~e->bits[w]);
	}
	e->cum[e->n_words] = z;
	e->n_zeros = z;
}

uint32_t orc_index_add_aligned(orc_index *idx, char const *aligned, uint64_t n_cols)
{
	orc_entry *e = add_entry(idx, n_cols);
	for (uint64_t i = 0; i < n_cols; ++i)
		if ('-' == aligned[i])	/* msa_index.hh:28 / index_vector_builder.cc:33-40 */
			e->bits[i >> 6] |= UINT64_C(1) << (i & 63);
	finish_entry(e);
	return idx->n_entries - 1;
}

uint32_t orc_index_add_bits(orc_index *idx, uint64_t const *gap_words, uint64_t n_cols)
{
	orc_entry *e = add_entry(idx, n_cols);
	memcpy(e->bits, gap_words, e->n_words * sizeof(uint64_t));
	finish_entry(e);
	return idx->n_entries - 1;
}

void orc_index_set_reference(orc_index *idx, uint32_t entry, char const *text, uint64_t len)
{
	idx->entries[entry].text = text;
	idx->entries[entry].text_len = len;
}

uint64_t orc_entry_cols(orc_index const *idx, uint32_t entry) { return idx->entries[entry].n_cols; }
uint64_t orc_entry_zeros(orc_index const *idx, uint32_t entry) { return idx->entries[entry].n_zeros; }

int orc_bit(orc_index const *idx, uint32_t entry, uint64_t i)
{
	orc_entry const *e = &idx->entries[entry];
	return (int) ((e->bits[i >> 6] >> (i & 63)) & 1);
}

/* rank_0(i): number of zeros in [0, i) — "may be zero" (msa_index.hh:246-250). */
static uint64_t entry_rank0(orc_entry const *e, uint64_t i)
{
	uint64_t const w = i >> 6;
	uint64_t r = e->cum[w];
	if (i & 63)
		r += (uint64_t) __builtin_popcountll(~e->bits[w] & ((UINT64_C(1) << (i & 63)) - 1));
	return r;
}

/* select_0(k): position of the k-th zero, k 1-based (msa_index.hh:239-243 calls it with 1 + pos). */
static uint64_t entry_select0(orc_entry const *e, uint64_t k)
{
	/* Largest w with cum[w] < k. */
	uint64_t lo = 0, hi = e->n_words;	/* invariant: cum[lo] < k <= cum[hi] */
	while (hi - lo > 1)
	{
		uint64_t const mid = lo + (hi - lo) / 2;
		if (e->cum[mid] < k) lo = mid; else hi = mid;
	}
	uint64_t rem = k - e->cum[lo];		/* rem-th zero inside word lo, rem >= 1 */
	uint64_t z = ~e->bits[lo];
	while (--rem)
		z &= z - 1;
	return (lo << 6) + (uint64_t) __builtin_ctzll(z);
}

uint64_t orc_rank0(orc_index const *idx, uint32_t entry, uint64_t i) { return entry_rank0(&idx->entries[entry], i); }
uint64_t orc_select0(orc_index const *idx, uint32_t entry, uint64_t k) { return entry_select0(&idx->entries[entry], k); }

uint64_t orc_rank0_naive(orc_index const *idx, uint32_t entry, uint64_t i)
{
	uint64_t r = 0;
	for (uint64_t c = 0; c < i; ++c)
		r += !orc_bit(idx, entry, c);
	return r;
}

uint64_t orc_select0_naive(orc_index const *idx, uint32_t entry, uint64_t k)
{
	orc_entry const *e = &idx->entries[entry];
	uint64_t seen = 0;
	for (uint64_t c = 0; c < e->n_cols; ++c)
	{
		if (!orc_bit(idx, entry, c) && ++seen == k)
			return c;
	}
	return e->n_cols;
}

/* msa_index.hh:253-258: dst.rank0(src.select0(1 + pos)). */
uint64_t orc_project_position(orc_index const *idx, uint32_t src, uint32_t dst, uint64_t pos)
{
	uint64_t const aln_pos = entry_select0(&idx->entries[src], 1 + pos);
	return entry_rank0(&idx->entries[dst], aln_pos);
}


/* ------------------------------------------------------------------------------------------------
 * cigar_buffer (cigar.hh:28-43, cigar.cc:54-95)
 * ---------------------------------------------------------------------------------------------- */

typedef struct
{
	uint32_t *ops;
	uint32_t cap;
	uint32_t n;
	uint32_t cur_op;
	uint32_t cur_count;
	int is_first;
	int overflow;
} cigar_buffer;

static inline uint32_t cig_make(uint32_t op, uint32_t count) { return (count << 4) | op; }
static inline uint32_t cig_op(uint32_t c) { return c & 0xf; }
static inline uint32_t cig_count(uint32_t c) { return c >> 4; }

static void cb_init(cigar_buffer *cb, uint32_t *ops, uint32_t cap)
{
	cb->ops = ops; cb->cap = cap; cb->n = 0; cb->cur_op = 0; cb->cur_count = 0; cb->is_first = 1; cb->overflow = 0;
}

static void cb_emit(cigar_buffer *cb)
{
	if (cb->n < cb->cap)
		cb->ops[cb->n] = cig_make(cb->cur_op, cb->cur_count);
	else
		cb->overflow = 1;
	++cb->n;
}

/* cigar.cc:61-87 */
static void cb_push_back(cigar_buffer *cb, uint32_t op, uint32_t count)
{
	if (0 == count)
		return;

	if (cb->is_first)
	{
		cb->is_first = 0;
		cb->cur_op = op;
		cb->cur_count = count;
	}
	else if (cb->cur_op == op)
	{
		cb->cur_count += count;
	}
	else
	{
		if (cb->cur_count)
			cb_emit(cb);
		cb->cur_op = op;
		cb->cur_count = count;
	}
}

/* cigar.cc:90-95 */
static void cb_finish(cigar_buffer *cb)
{
	if (!cb->is_first && cb->cur_count)
		cb_emit(cb);
}

/* cigar.cc:13-51.  (The reference is UB on an empty vector; we return 0.) */
uint32_t orc_collapse_cigar(uint32_t *ops, uint32_t n)
{
	if (0 == n)
		return 0;
	uint32_t it = 0;
	for (uint32_t it_ = 1; it_ < n; ++it_)
	{
		if (cig_op(ops[it]) == cig_op(ops[it_]))
			ops[it] = cig_make(cig_op(ops[it]), cig_count(ops[it]) + cig_count(ops[it_]));
		else
			ops[++it] = ops[it_];
	}
	return it + 1;
}


/* ------------------------------------------------------------------------------------------------
 * rewrite_cigar (rewrite_cigar.hh:18-157) — per-base select0 / rank0 / bit access, literally.
 * ---------------------------------------------------------------------------------------------- */

enum { ORC_OK = 0, ORC_ERR_SRC_RANGE = 1, ORC_ERR_CIGAR_OP = 2, ORC_ERR_QUERY_RANGE = 3, ORC_ERR_REF_RANGE = 4, ORC_ERR_CAPACITY = 5 };

static int rewrite_cigar_impl(
	orc_entry const *src, orc_entry const *dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len,
	cigar_buffer *destination, uint64_t *dst_pos
)
{
	uint64_t query_pos = 0;
	if (!(src_pos < src->n_zeros))
		return ORC_ERR_SRC_RANGE;
	uint64_t const aln_pos0 = entry_select0(src, 1 + src_pos);		/* :51 */
	uint64_t prev_excess = entry_rank0(dst, aln_pos0);				/* :52 */
	*dst_pos = prev_excess;											/* :56 */

	for (uint32_t ci = 0; ci < n_cigar; ++ci)
	{
		uint32_t const op_count = cig_count(cigar[ci]);
		uint32_t const operation = cig_op(cigar[ci]);

		switch (operation)
		{
			case ORC_I:
			case ORC_S:		/* :69-73 */
				cb_push_back(destination, operation, op_count);
				query_pos += op_count;
				break;

			case ORC_H:
			case ORC_P:		/* :75-78 */
				cb_push_back(destination, operation, op_count);
				break;

			case ORC_M:
			case ORC_EQ:
			case ORC_X:		/* :80-118 */
				for (uint32_t i = 0; i < op_count; ++i)
				{
					if (!(src_pos < src->n_zeros))
						return ORC_ERR_SRC_RANGE;
					uint64_t const aln_pos = entry_select0(src, 1 + src_pos);	/* :88 */
					uint64_t const excess = entry_rank0(dst, aln_pos);			/* :89 */
					if (prev_excess < excess)									/* :90-91 */
						cb_push_back(destination, ORC_D, (uint32_t) (excess - prev_excess));

					if (1 == ((dst->bits[aln_pos >> 6] >> (aln_pos & 63)) & 1))	/* :93 */
					{
						cb_push_back(destination, ORC_I, 1);
						prev_excess = excess;
					}
					else
					{
						if (!(query_pos < query_len))							/* :102 */
							return ORC_ERR_QUERY_RANGE;
						if (!(excess < dst->text_len))							/* :103 */
							return ORC_ERR_REF_RANGE;
						char const query_cc = query[query_pos];
						char const dst_cc = dst->text[excess];
						if (query_cc == dst_cc)									/* :107-110, raw byte compare (quirk Q3) */
							cb_push_back(destination, ORC_EQ, 1);
						else
							cb_push_back(destination, ORC_X, 1);
						prev_excess = excess + 1;
					}

					++query_pos;
					++src_pos;
				}
				break;

			case ORC_D:
			case ORC_N:		/* :121-147; N comes out as D (quirk Q4) */
				for (uint32_t i = 0; i < op_count; ++i)
				{
					if (!(src_pos < src->n_zeros))
						return ORC_ERR_SRC_RANGE;
					uint64_t const aln_pos = entry_select0(src, 1 + src_pos);	/* :128 */
					uint64_t const excess = entry_rank0(dst, aln_pos);			/* :129 */
					if (prev_excess < excess)									/* :130-131 */
						cb_push_back(destination, ORC_D, (uint32_t) (excess - prev_excess));

					if (0 == ((dst->bits[aln_pos >> 6] >> (aln_pos & 63)) & 1))	/* :133 */
					{
						cb_push_back(destination, ORC_D, 1);
						prev_excess = excess + 1;
					}
					else
					{
						prev_excess = excess;
					}
					++src_pos;
				}
				break;

			default:		/* :149-151 libbio_fail */
				return ORC_ERR_CIGAR_OP;
		}
	}

	cb_finish(destination);			/* :155 */
	return destination->overflow ? ORC_ERR_CAPACITY : ORC_OK;
}

int orc_rewrite_cigar(
	orc_index const *idx, uint32_t src, uint32_t dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out, uint64_t *dst_pos
)
{
	cigar_buffer cb;
	cb_init(&cb, out_cigar, out_cap);
	int const st = rewrite_cigar_impl(&idx->entries[src], &idx->entries[dst], src_pos, cigar, n_cigar, query, query_len, &cb, dst_pos);
	*n_out = cb.n;
	return -st;
}


/* ------------------------------------------------------------------------------------------------
 * indel_run_checker (indel_run_checker.hh:18-58, indel_run_checker.cc:13-121)
 * ---------------------------------------------------------------------------------------------- */

enum { RUN_HAS_INSERTIONS = 0x1, RUN_HAS_DELETIONS = 0x2, RUN_HAS_BOTH = 0x3, RUN_HAS_NEITHER = 0x0 };

typedef struct
{
	uint32_t const *cigar;
	uint32_t first, second;					/* m_cigar_range */
	uint32_t realigned_first, realigned_second;	/* m_cigar_realigned_range */
	uint64_t ref_range_location, ref_range_length;
	uint64_t query_range_location, query_range_length;
	uint64_t ref_pos, query_pos;
	uint8_t run_type;
} indel_run_checker;

/* indel_run_checker.cc:13-21 */
static void irc_reset(indel_run_checker *c, uint32_t const *cigar, uint32_t n, uint64_t ref_pos)
{
	c->cigar = cigar; c->first = 0; c->second = n;
	c->realigned_first = 0; c->realigned_second = 0;
	c->ref_pos = ref_pos; c->query_pos = 0;
	c->ref_range_location = ref_pos; c->ref_range_length = 0;
	c->query_range_location = 0; c->query_range_length = 0;
	c->run_type = RUN_HAS_NEITHER;
}

/* indel_run_checker.cc:31-121; returns 1 if a run was found, 0 at the end, -1 on a bad op. */
static int irc_find_next(indel_run_checker *c)
{
	while (c->first != c->second)
	{
		uint64_t const ref_pos = c->ref_pos;
		uint64_t const query_pos = c->query_pos;
		uint8_t const run_type = c->run_type;
		int report = 0;

		if (RUN_HAS_NEITHER == c->run_type)		/* :45-50 */
		{
			c->ref_range_location = c->ref_pos;
			c->query_range_location = c->query_pos;
			c->realigned_first = c->first;
		}

		uint32_t const count = cig_count(c->cigar[c->first]);
		switch (cig_op(c->cigar[c->first]))
		{
			case ORC_H:
			case ORC_P:		/* :55-60 */
				c->run_type = RUN_HAS_NEITHER;
				report = (RUN_HAS_BOTH == run_type);
				break;
			case ORC_I:		/* :62-65 */
				c->query_pos += count;
				c->run_type |= RUN_HAS_INSERTIONS;
				break;
			case ORC_D:		/* :67-70 */
				c->ref_pos += count;
				c->run_type |= RUN_HAS_DELETIONS;
				break;
			case ORC_S:		/* :72-77 */
				c->query_pos += count;
				c->run_type = RUN_HAS_NEITHER;
				report = (RUN_HAS_BOTH == run_type);
				break;
			case ORC_N:		/* :79-84 */
				c->ref_pos += count;
				c->run_type = RUN_HAS_NEITHER;
				report = (RUN_HAS_BOTH == run_type);
				break;
			case ORC_M:
			case ORC_EQ:
			case ORC_X:		/* :86-94 */
				c->ref_pos += count;
				c->query_pos += count;
				c->run_type = RUN_HAS_NEITHER;
				report = (RUN_HAS_BOTH == run_type);
				break;
			default:
				return -1;
		}

		if (report)			/* :104-108 */
		{
			c->ref_range_length = ref_pos - c->ref_range_location;
			c->query_range_length = query_pos - c->query_range_location;
			c->realigned_second = c->first;
			++c->first;
			return 1;
		}
		++c->first;
	}

	if (RUN_HAS_BOTH == c->run_type)	/* :111-118 */
	{
		c->run_type = RUN_HAS_NEITHER;
		c->ref_range_length = c->ref_pos - c->ref_range_location;
		c->query_range_length = c->query_pos - c->query_range_location;
		c->realigned_second = c->first;
		return 1;
	}
	return 0;
}

int orc_indel_runs(
	uint32_t const *cigar, uint32_t n_cigar, uint64_t ref_pos,
	orc_run *runs, uint32_t *run_ref_location, uint32_t *cigar_first, uint32_t *cigar_second,
	uint32_t run_cap, uint32_t *n_runs, uint64_t *final_query_pos, uint64_t *final_ref_pos
)
{
	indel_run_checker c;
	irc_reset(&c, cigar, n_cigar, ref_pos);
	uint32_t k = 0;
	int res;
	while ((res = irc_find_next(&c)) > 0)
	{
		if (k < run_cap)
		{
			runs[k].query_location = (uint32_t) c.query_range_location;
			runs[k].query_length = (uint32_t) c.query_range_length;
			runs[k].ref_position = (uint32_t) c.ref_pos;
			runs[k].ref_length = (uint32_t) c.ref_range_length;
			run_ref_location[k] = (uint32_t) c.ref_range_location;
			cigar_first[k] = c.realigned_first;
			cigar_second[k] = c.realigned_second;
		}
		++k;
	}
	*n_runs = k;
	*final_query_pos = c.query_pos;
	*final_ref_pos = c.ref_pos;
	if (res < 0) return -ORC_ERR_CIGAR_OP;
	return (k > run_cap) ? -ORC_ERR_CAPACITY : 0;
}


/* ------------------------------------------------------------------------------------------------
 * align_global (align.hh:59-112) with seqan3::align_pairwise restated.
 *
 * seqan3 is NOT in the reference snapshot.  Restated from upstream seqan3 3.x:
 *   - global alignment, affine gaps: a gap of length L costs open + L * extension
 *     (align_cfg::gap_cost_affine; align.hh:85-88 passes the CLI values as open_score/extension_score);
 *   - seq1 (reference slice) is the horizontal sequence (matrix columns), seq2 (query slice) the vertical;
 *   - cell update (policy_affine_gap_with_trace_recursion::compute_inner_cell):
 *         best = diag + s;  if (best < V) best = V [up];  if (best < H) best = H [left];      (strict <)
 *         V' = max(V + ext, best + open + ext), H' likewise; the extension is kept on ties     (strict <)
 *   - traceback from the bottom-right cell (trace_iterator_base): at a cell where the direction is
 *     (re)decided the priority is diagonal, then up (gap in seq1 => 'I'), then left ('D'); inside a gap
 *     it continues until the cell whose gap score was formed by opening.
 *   - scoring: <false> seqan3::nucleotide_scoring_scheme<>{} = match 0 / mismatch -1 (hamming default);
 *              <true>  Malde base-quality scores as doubles (align.hh:23-38).
 *   - score type: ORC_SCORE_FP64 keeps doubles; ORC_SCORE_INT32_TRUNC converts each substitution score
 *     to int32 (C++ double -> int32 conversion = truncation toward zero), upstream's default score_type.
 * Both modes are evaluated in double arithmetic (every int32 intermediate is exactly representable).
 * ---------------------------------------------------------------------------------------------- */

/* seqan3::dna5 ranks A=0 C=1 G=2 N=3 T=4; assign_char is case-insensitive, U -> T, everything else -> N. */
static inline uint8_t dna5_rank(char c)
{
	switch (c)
	{
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': case 'U': case 'u': return 4;
		default: return 3;
	}
}

/* align.hh:28-37, evaluation order as written. */
double orc_malde_score(int matches, int lhs_phred, int rhs_phred)
{
	double const lhs_e = 1.0 / pow(10.0, lhs_phred / 10.0);
	double const rhs_e = 1.0 / pow(10.0, rhs_phred / 10.0);
	double const combined_e = lhs_e + rhs_e - lhs_e / 3.0 * rhs_e * 4.0;
	if (matches)
		return 2.0 + log2(1.0 - combined_e);
	else
		return 2.0 - log2(3.0) + log2(combined_e);
}

enum { TR_DIAG = 0, TR_UP = 1, TR_LEFT = 2, TR_DIR_MASK = 3, TR_V_OPEN = 4, TR_H_OPEN = 8 };

#define ORC_MAX_PHRED 41	/* max_letter<phred42> (utility.hh:75-80, alignment_projector.cc:108) */

/* Tie-break policies (seqan3's choice among co-optimal alignments is not pinned by the reference's tests
 * beyond four sub-problems).  They travel in bits 8-15 of the score_mode argument (ORC_TIEBREAK(mode)):
 *   mode = 2 * p + o;  p = 0..5: the order in which equal candidates of a cell are preferred,
 *       0 diagonal > up > left (the restated seqan3 order; the default), 1 diagonal > left > up, 2 up > diagonal > left,
 *       3 up > left > diagonal, 4 left > diagonal > up, 5 left > up > diagonal  (up = gap in the reference = 'I', left = 'D');
 *   o = 0: a gap that could be extended or opened at the same score is extended (default), 1: it is opened.
 * The scores are the same under every policy; only the path among the co-optimal ones differs. */
static int const TB_ORDER[6][3] = {{TR_DIAG, TR_UP, TR_LEFT}, {TR_DIAG, TR_LEFT, TR_UP}, {TR_UP, TR_DIAG, TR_LEFT}, {TR_UP, TR_LEFT, TR_DIAG}, {TR_LEFT, TR_DIAG, TR_UP}, {TR_LEFT, TR_UP, TR_DIAG}};

int orc_align_global(
	char const *ref, uint32_t m, char const *query, uint32_t n, uint8_t const *qual,
	int32_t gap_open, int32_t gap_ext, int score_mode,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out, double *score_out
)
{
	int const tiebreak = (score_mode >> 8) & 0xff;
	score_mode &= 0xff;
	if (tiebreak >= 12) return -ORC_ERR_CIGAR_OP;
	int const *const order = TB_ORDER[tiebreak >> 1];
	int const open_on_tie = tiebreak & 1;
#define OPENS(ext, open) (open_on_tie ? !((open) < (ext)) : ((ext) < (open)))
	double const go = (double) gap_open + (double) gap_ext;	/* seqan3's internal gap_open_score = open + extension */
	double const ge = (double) gap_ext;

	/* Substitution scores. */
	double s_match[ORC_MAX_PHRED + 1], s_mismatch[ORC_MAX_PHRED + 1];
	if (qual)
	{
		for (int q = 0; q <= ORC_MAX_PHRED; ++q)
		{
			s_match[q] = orc_malde_score(1, ORC_MAX_PHRED, q);
			s_mismatch[q] = orc_malde_score(0, ORC_MAX_PHRED, q);
			if (ORC_SCORE_INT32_TRUNC == score_mode)
			{
				s_match[q] = (double) (int32_t) s_match[q];
				s_mismatch[q] = (double) (int32_t) s_mismatch[q];
			}
		}
	}

	size_t const W = (size_t) m + 1;
	size_t const cells = ((size_t) n + 1) * W;
	double *M = (double *) malloc(cells * sizeof(double));
	double *V = (double *) malloc(cells * sizeof(double));	/* V(i,j): best score ending in a vertical gap at (i,j) */
	double *H = (double *) malloc(cells * sizeof(double));
	uint8_t *T = (uint8_t *) malloc(cells);
	if (!M || !V || !H || !T) { free(M); free(V); free(H); free(T); return -ORC_ERR_CAPACITY; }

#define AT(i, j) ((size_t) (i) * W + (size_t) (j))
	M[AT(0, 0)] = 0.0; V[AT(0, 0)] = -INFINITY; H[AT(0, 0)] = -INFINITY; T[AT(0, 0)] = 0;
	for (uint32_t j = 1; j <= m; ++j)	/* first row: initialise_first_row_cell */
	{
		double const open = M[AT(0, j - 1)] + go, ext = H[AT(0, j - 1)] + ge;
		uint8_t t = TR_LEFT;
		if (OPENS(ext, open)) { H[AT(0, j)] = open; t |= TR_H_OPEN; } else H[AT(0, j)] = ext;
		M[AT(0, j)] = H[AT(0, j)];
		V[AT(0, j)] = -INFINITY;
		T[AT(0, j)] = t;
	}
	for (uint32_t i = 1; i <= n; ++i)
	{
		{	/* first column: initialise_first_column_cell */
			double const open = M[AT(i - 1, 0)] + go, ext = V[AT(i - 1, 0)] + ge;
			uint8_t t = TR_UP;
			if (OPENS(ext, open)) { V[AT(i, 0)] = open; t |= TR_V_OPEN; } else V[AT(i, 0)] = ext;
			M[AT(i, 0)] = V[AT(i, 0)];
			H[AT(i, 0)] = -INFINITY;
			T[AT(i, 0)] = t;
		}
		uint8_t const qr = dna5_rank(query[i - 1]);
		int const qq = qual ? (qual[i - 1] > ORC_MAX_PHRED ? ORC_MAX_PHRED : qual[i - 1]) : 0;
		for (uint32_t j = 1; j <= m; ++j)
		{
			int const is_match = (dna5_rank(ref[j - 1]) == qr);	/* dna5 equality; N == N matches (quirk Q3) */
			double const s = qual ? (is_match ? s_match[qq] : s_mismatch[qq]) : (is_match ? 0.0 : -1.0);
			uint8_t t = 0;

			/* Gap scores arriving at (i,j); ties keep the extension. */
			double v_open = M[AT(i - 1, j)] + go, v_ext = V[AT(i - 1, j)] + ge, v;
			if (OPENS(v_ext, v_open)) { v = v_open; t |= TR_V_OPEN; } else v = v_ext;
			double h_open = M[AT(i, j - 1)] + go, h_ext = H[AT(i, j - 1)] + ge, h;
			if (OPENS(h_ext, h_open)) { h = h_open; t |= TR_H_OPEN; } else h = h_ext;

			/* the candidates in the policy's order; a later one replaces an earlier one only if it is strictly better */
			double cand[3];
			cand[TR_DIAG] = M[AT(i - 1, j - 1)] + s; cand[TR_UP] = v; cand[TR_LEFT] = h;
			double best = cand[order[0]];
			uint8_t dir = (uint8_t) order[0];
			if (best < cand[order[1]]) { best = cand[order[1]]; dir = (uint8_t) order[1]; }
			if (best < cand[order[2]]) { best = cand[order[2]]; dir = (uint8_t) order[2]; }

			M[AT(i, j)] = best; V[AT(i, j)] = v; H[AT(i, j)] = h; T[AT(i, j)] = (uint8_t) (t | dir);
		}
	}
	if (score_out)
		*score_out = M[AT(n, m)];

	/* Traceback (trace_iterator_base::operator++ / set_trace_direction). Collect ops back to front. */
	uint32_t const max_ops = m + n;
	uint8_t *rev = (uint8_t *) malloc(max_ops ? max_ops : 1);
	uint32_t k = 0;
	{
		uint32_t i = n, j = m;
		enum { ST_M, ST_V, ST_H } st = ST_M;
		while (i || j)
		{
			uint8_t const t = T[AT(i, j)];
			if (ST_M == st)
			{
				uint8_t const dir = t & TR_DIR_MASK;
				if (TR_DIAG == dir)
				{
					rev[k++] = (dna5_rank(ref[j - 1]) == dna5_rank(query[i - 1])) ? ORC_EQ : ORC_X;	/* align.hh:103-106 */
					--i; --j;
					continue;
				}
				st = (TR_UP == dir) ? ST_V : ST_H;
			}
			if (ST_V == st)
			{
				rev[k++] = ORC_I;	/* gap in seq1 (reference): align.hh:99-100 */
				if (t & TR_V_OPEN) st = ST_M;
				--i;
			}
			else
			{
				rev[k++] = ORC_D;	/* gap in seq2 (query): align.hh:101-102 */
				if (t & TR_H_OPEN) st = ST_M;
				--j;
			}
		}
	}
#undef AT

	cigar_buffer cb;
	cb_init(&cb, out_cigar, out_cap);
	for (uint32_t x = k; x > 0; --x)
		cb_push_back(&cb, rev[x - 1], 1);	/* align.hh:94-108 */
	cb_finish(&cb);							/* align.hh:110 */
	*n_out = cb.n;

	free(rev); free(M); free(V); free(H); free(T);
#undef OPENS
	return cb.overflow ? -ORC_ERR_CAPACITY : 0;
}


/* ------------------------------------------------------------------------------------------------
 * alignment_projector::project_alignment (alignment_projector.cc:30-154)
 * ---------------------------------------------------------------------------------------------- */

static int project_alignment_impl(
	orc_entry const *src, orc_entry const *dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len, uint8_t const *qual,
	int32_t gap_open, int32_t gap_ext, int score_mode,
	uint32_t **scratch, uint32_t *scratch_cap,	/* growable rewrite buffer */
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out,
	orc_run *runs, uint32_t run_cap, uint32_t *n_runs, uint64_t *dst_pos
)
{
	*n_out = 0; *n_runs = 0;

	/* :42-50 rewrite into m_rewrite_buffer */
	cigar_buffer rw;
	int st;
	for (;;)
	{
		cb_init(&rw, *scratch, *scratch_cap);
		st = rewrite_cigar_impl(src, dst, src_pos, cigar, n_cigar, query, query_len, &rw, dst_pos);
		if (ORC_ERR_CAPACITY != st)
			break;
		*scratch_cap = rw.n + 16;
		*scratch = (uint32_t *) realloc(*scratch, *scratch_cap * sizeof(uint32_t));
	}
	if (st)
		return st;

	uint32_t const *rewritten = *scratch;
	uint32_t const n_rewritten = rw.n;

	/* :53-56 */
	indel_run_checker checker;
	irc_reset(&checker, rewritten, n_rewritten, *dst_pos);
	uint32_t n = 0;		/* m_cigar_realigned */
	int overflow = 0;
#define APPEND(v) do { if (n < out_cap) out_cigar[n] = (v); else overflow = 1; ++n; } while (0)

	uint32_t cigar_begin = 0;
	int res;
	uint32_t k = 0;
	while ((res = irc_find_next(&checker)) > 0)		/* :58 */
	{
		for (uint32_t x = cigar_begin; x < checker.realigned_first; ++x)	/* :61-65 */
			APPEND(rewritten[x]);
		cigar_begin = checker.realigned_second;							/* :66 */

		/* :69-73 — reference_position() is the position AFTER the terminating op (quirk Q7). */
		if (k < run_cap)
		{
			runs[k].ref_position = (uint32_t) checker.ref_pos;
			runs[k].ref_length = (uint32_t) checker.ref_range_length;
			runs[k].query_location = (uint32_t) checker.query_range_location;
			runs[k].query_length = (uint32_t) checker.query_range_length;
		}
		else overflow = 1;
		++k;

		/* :76-137 */
		uint32_t const m = (uint32_t) checker.ref_range_length;
		uint32_t const nq = (uint32_t) checker.query_range_length;
		if (checker.ref_range_location + m > dst->text_len)
			return ORC_ERR_REF_RANGE;
		if (checker.query_range_location + nq > query_len)
			return ORC_ERR_QUERY_RANGE;
		uint32_t const realn_cap = m + nq + 1;
		uint32_t *realn = (uint32_t *) malloc(realn_cap * sizeof(uint32_t));
		uint32_t realn_n = 0;
		int const ast = orc_align_global(
			dst->text + checker.ref_range_location, m,
			query + checker.query_range_location, nq,
			qual ? qual + checker.query_range_location : NULL,
			gap_open, gap_ext, score_mode,
			realn, realn_cap, &realn_n, NULL
		);
		if (ast) { free(realn); return -ast; }

		for (uint32_t x = 0; x < realn_n; ++x)		/* :140-141 */
			APPEND(realn[x]);
		free(realn);
	}
	if (res < 0)
		return ORC_ERR_CIGAR_OP;

	for (uint32_t x = cigar_begin; x < n_rewritten; ++x)	/* :145-148 */
		APPEND(rewritten[x]);
#undef APPEND

	*n_runs = k;
	*n_out = n;
	if (overflow)
		return ORC_ERR_CAPACITY;

	*n_out = orc_collapse_cigar(out_cigar, n);		/* :151 */
	return ORC_OK;
}

int orc_project_alignment(
	orc_index const *idx, uint32_t src, uint32_t dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len, uint8_t const *qual,
	int32_t gap_open, int32_t gap_ext, int score_mode,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out,
	orc_run *runs, uint32_t run_cap, uint32_t *n_runs, uint64_t *dst_pos
)
{
	uint32_t *scratch = NULL;
	uint32_t scratch_cap = 0;
	int const st = project_alignment_impl(
		&idx->entries[src], &idx->entries[dst], src_pos, cigar, n_cigar, query, query_len, qual,
		gap_open, gap_ext, score_mode, &scratch, &scratch_cap,
		out_cigar, out_cap, n_out, runs, run_cap, n_runs, dst_pos
	);
	free(scratch);
	return -st;
}


/* ------------------------------------------------------------------------------------------------
 * Batch driver (the loop of project_alignments.cc:700-732 + :897-901 over an in-memory SoA batch).
 * ---------------------------------------------------------------------------------------------- */

typedef struct
{
	orc_index const *idx;
	orc_batch_in const *in;
	orc_batch_out *out;
	int32_t gap_open, gap_ext;
	int score_mode;
	uint64_t begin, end;
	/* thread-local packed outputs */
	uint32_t *cig; uint64_t cig_n, cig_cap;
	orc_run *runs; uint64_t runs_n, runs_cap;
	uint32_t *cig_counts;	/* per record */
	uint32_t *run_counts;
} batch_task;

/* BAM 4-bit code -> seqan3::dna5::to_char(): "=ACMGRSVTWYHKDBN" read through dna5::assign_char. */
static char const seq4_to_dna5_char[16] = {'N', 'A', 'C', 'N', 'G', 'N', 'N', 'N', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};

static void *batch_worker(void *arg)
{
	batch_task *t = (batch_task *) arg;
	orc_batch_in const *in = t->in;
	orc_batch_out *out = t->out;
	uint32_t *scratch = NULL; uint32_t scratch_cap = 0;
	char *query = NULL; uint8_t *qual = NULL; uint32_t query_cap = 0;
	uint32_t *rec_cig = NULL; uint32_t rec_cig_cap = 64;
	orc_run *rec_runs = NULL; uint32_t rec_runs_cap = 16;
	rec_cig = (uint32_t *) malloc(rec_cig_cap * sizeof(uint32_t));
	rec_runs = (orc_run *) malloc(rec_runs_cap * sizeof(orc_run));

	for (uint64_t r = t->begin; r < t->end; ++r)
	{
		uint32_t const l_seq = in->l_seq[r];
		if (l_seq > query_cap)
		{
			query_cap = l_seq + 64;
			query = (char *) realloc(query, query_cap);
			qual = (uint8_t *) realloc(qual, query_cap);
		}
		uint8_t const *s4 = in->seq4 + in->seq_off[r];
		for (uint32_t i = 0; i < l_seq; ++i)
			query[i] = seq4_to_dna5_char[(s4[i >> 1] >> ((~i & 1) << 2)) & 0xf];
		int const has_qual = (in->flags[r] & ORC_FLAG_HAS_QUAL) && in->qual;
		if (has_qual)
		{
			uint8_t const *qq = in->qual + in->qual_off[r];
			for (uint32_t i = 0; i < l_seq; ++i)
				qual[i] = qq[i] > ORC_MAX_PHRED ? ORC_MAX_PHRED : qq[i];	/* phred42 saturates */
		}

		uint32_t const *cigar = in->cigar + in->cigar_off[r];
		uint32_t const n_cigar = (uint32_t) (in->cigar_off[r + 1] - in->cigar_off[r]);
		orc_entry const *src = &t->idx->entries[in->src_entry[r]];
		orc_entry const *dst = &t->idx->entries[in->dst_entry[r]];

		/* Up-front validation defines the per-record status (shared with the HIP path, include/pvc3.h). */
		int st = ORC_OK;
		uint64_t ref_consumed = 0, query_consumed = 0;
		for (uint32_t c = 0; c < n_cigar; ++c)
		{
			uint32_t const op = cig_op(cigar[c]), cnt = cig_count(cigar[c]);
			if (op > ORC_X) { st = ORC_ERR_CIGAR_OP; break; }
			if (ORC_M == op || ORC_EQ == op || ORC_X == op) { ref_consumed += cnt; query_consumed += cnt; }
			else if (ORC_D == op || ORC_N == op) ref_consumed += cnt;
			else if (ORC_I == op || ORC_S == op) query_consumed += cnt;
		}
		if (!st && (src->n_cols != dst->n_cols || (uint64_t) in->src_pos[r] + (ref_consumed ? ref_consumed : 1) > src->n_zeros)) st = ORC_ERR_SRC_RANGE;
		if (!st && query_consumed > l_seq) st = ORC_ERR_QUERY_RANGE;

		uint32_t n_out = 0, n_runs = 0;
		uint64_t dst_pos = 0;
		if (!st)
		{
			for (;;)
			{
				st = project_alignment_impl(
					src, dst, in->src_pos[r], cigar, n_cigar, query, l_seq, has_qual ? qual : NULL,
					t->gap_open, t->gap_ext, t->score_mode, &scratch, &scratch_cap,
					rec_cig, rec_cig_cap, &n_out, rec_runs, rec_runs_cap, &n_runs, &dst_pos
				);
				if (ORC_ERR_CAPACITY != st)
					break;
				if (n_out > rec_cig_cap) { rec_cig_cap = n_out + 64; rec_cig = (uint32_t *) realloc(rec_cig, rec_cig_cap * sizeof(uint32_t)); }
				if (n_runs > rec_runs_cap) { rec_runs_cap = n_runs + 16; rec_runs = (orc_run *) realloc(rec_runs, rec_runs_cap * sizeof(orc_run)); }
			}
		}
		if (st) { n_out = 0; n_runs = 0; dst_pos = 0; }

		out->status[r] = st;
		out->dst_pos[r] = (uint32_t) dst_pos;

		/* Mate: project_position (project_alignments.cc:897-901). */
		uint32_t mate = 0xFFFFFFFFu;
		if (in->flags[r] & ORC_FLAG_HAS_MATE)
		{
			orc_entry const *ms = &t->idx->entries[in->mate_src_entry[r]];
			orc_entry const *md = &t->idx->entries[in->mate_dst_entry[r]];
			if (in->mate_pos[r] < ms->n_zeros && ms->n_cols == md->n_cols)
				mate = (uint32_t) entry_rank0(md, entry_select0(ms, 1 + (uint64_t) in->mate_pos[r]));
		}
		out->mate_dst_pos[r] = mate;

		if (t->cig_n + n_out > t->cig_cap)
		{
			t->cig_cap = 2 * (t->cig_n + n_out) + 1024;
			t->cig = (uint32_t *) realloc(t->cig, t->cig_cap * sizeof(uint32_t));
		}
		memcpy(t->cig + t->cig_n, rec_cig, n_out * sizeof(uint32_t));
		t->cig_n += n_out;
		if (t->runs_n + n_runs > t->runs_cap)
		{
			t->runs_cap = 2 * (t->runs_n + n_runs) + 64;
			t->runs = (orc_run *) realloc(t->runs, t->runs_cap * sizeof(orc_run));
		}
		memcpy(t->runs + t->runs_n, rec_runs, n_runs * sizeof(orc_run));
		t->runs_n += n_runs;
		t->cig_counts[r - t->begin] = n_out;
		t->run_counts[r - t->begin] = n_runs;
	}

	free(scratch); free(query); free(qual); free(rec_cig); free(rec_runs);
	return NULL;
}

int orc_batch_project(
	orc_index const *idx, orc_batch_in const *in, orc_batch_out *out,
	int32_t gap_open, int32_t gap_ext, int score_mode, int n_threads
)
{
	if (n_threads < 1) n_threads = 1;
	if ((uint64_t) n_threads > in->n && in->n) n_threads = (int) in->n;
	batch_task *tasks = (batch_task *) calloc((size_t) n_threads, sizeof(batch_task));
	pthread_t *threads = (pthread_t *) calloc((size_t) n_threads, sizeof(pthread_t));

	for (int i = 0; i < n_threads; ++i)
	{
		batch_task *t = &tasks[i];
		t->idx = idx; t->in = in; t->out = out;
		t->gap_open = gap_open; t->gap_ext = gap_ext; t->score_mode = score_mode;
		t->begin = in->n * (uint64_t) i / (uint64_t) n_threads;
		t->end = in->n * (uint64_t) (i + 1) / (uint64_t) n_threads;
		t->cig_counts = (uint32_t *) malloc((t->end - t->begin + 1) * sizeof(uint32_t));
		t->run_counts = (uint32_t *) malloc((t->end - t->begin + 1) * sizeof(uint32_t));
	}
	if (1 == n_threads)
		batch_worker(&tasks[0]);
	else
	{
		for (int i = 0; i < n_threads; ++i)
			pthread_create(&threads[i], NULL, batch_worker, &tasks[i]);
		for (int i = 0; i < n_threads; ++i)
			pthread_join(threads[i], NULL);
	}

	/* Stitch the thread-local packed outputs in record order. */
	uint64_t cig_total = 0, runs_total = 0;
	for (int i = 0; i < n_threads; ++i) { cig_total += tasks[i].cig_n; runs_total += tasks[i].runs_n; }
	int rc = 0;
	if (cig_total > out->out_cigar_cap || runs_total > out->run_cap)
		rc = -2;
	uint64_t co = 0, ro = 0;
	for (int i = 0; i < n_threads; ++i)
	{
		batch_task *t = &tasks[i];
		if (!rc)
		{
			memcpy(out->out_cigar + co, t->cig, t->cig_n * sizeof(uint32_t));
			memcpy(out->runs + ro, t->runs, t->runs_n * sizeof(orc_run));
		}
		for (uint64_t r = t->begin; r < t->end; ++r)
		{
			out->out_cigar_off[r] = co; co += t->cig_counts[r - t->begin];
			out->run_off[r] = ro; ro += t->run_counts[r - t->begin];
		}
		free(t->cig); free(t->runs); free(t->cig_counts); free(t->run_counts);
	}
	out->out_cigar_off[in->n] = co;
	out->run_off[in->n] = ro;
	free(tasks); free(threads);
	return rc;
}
