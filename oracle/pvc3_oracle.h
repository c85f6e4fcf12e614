/*
 * pvc3_oracle.h — CPU oracle for the project_alignments hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under panvc3_amd/ (the product) may include, link,
 * import or call anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * What it is: a deliberately literal, scalar C restatement of the reference algorithm
 * (tsnorri/panvc3 @ 2024-10-16).  Each function cites the reference file:line it follows.
 *
 * Pinning status
 *   - rank0/select0/rewrite_cigar/indel_run_checker/cigar_buffer/collapse: PINNED by the
 *     reference's own fixtures tests/rewrite_cigar_inputs.tsv, tests/indel_run_checker_inputs.tsv
 *     (copied as data into tests/golden/), see tests/test_oracle_golden.py.
 *   - align_global (the realignment DP): the arithmetic lives in seqan3 (tsnorri fork, commit
 *     unrecoverable: lib/seqan3 is an empty directory in the reference snapshot).  The recurrence
 *     and traceback priority below restate upstream seqan3 3.x's published affine-gap algorithm
 *     (policy_affine_gap_with_trace_recursion + trace_iterator_base).  It is pinned only by
 *     tests/alignment_projector_inputs.tsv modulo the reference's own cigar_eq comparator
 *     (TSV line 10 contradicts line 9).  Exact I/D order among co-optimal
 *     alignments and the score type (double vs int32 truncation): PARITY UNPINNED.
 */
#ifndef PVC3_ORACLE_H
#define PVC3_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BAM CIGAR op codes, "MIDNSHP=X" (SAMv1 §4.2; seqan3 cigar alphabet order). */
enum { ORC_M = 0, ORC_I = 1, ORC_D = 2, ORC_N = 3, ORC_S = 4, ORC_H = 5, ORC_P = 6, ORC_EQ = 7, ORC_X = 8 };

enum { ORC_SCORE_FP64 = 0, ORC_SCORE_INT32_TRUNC = 1 };
/* The tie-break policy 0 .. 11 (see orc_align_global) travels in bits 8-15 of every score_mode argument. */
#define ORC_TIEBREAK(mode) ((int) (mode) << 8)

/* Record flag bits (shared meaning with include/pvc3.h). */
enum { ORC_FLAG_HAS_QUAL = 1u, ORC_FLAG_HAS_MATE = 2u };

typedef struct orc_index orc_index;

typedef struct
{
	uint32_t query_location;	/* query range [location, location + length), includes soft-clipped bases */
	uint32_t query_length;
	uint32_t ref_position;		/* indel_run_checker::reference_position() at report time (quirk Q7) */
	uint32_t ref_length;
} orc_run;

orc_index *orc_index_new(void);
void orc_index_free(orc_index *idx);
/* Bit i set iff aligned[i] == '-'  (index-msa/index_vector_builder.cc:33-40, msa_index.hh:19-33). */
uint32_t orc_index_add_aligned(orc_index *idx, char const *aligned, uint64_t n_cols);
uint32_t orc_index_add_bits(orc_index *idx, uint64_t const *gap_words, uint64_t n_cols);
/* Unaligned text of an entry (the dst contig as raw FASTA bytes); borrowed pointer. */
void orc_index_set_reference(orc_index *idx, uint32_t entry, char const *text, uint64_t len);
uint64_t orc_entry_cols(orc_index const *idx, uint32_t entry);
uint64_t orc_entry_zeros(orc_index const *idx, uint32_t entry);

/* sdsl semantics (msa_index.hh:239-258): rank0(i) = #zeros in [0, i); select0(k), k 1-based. */
uint64_t orc_rank0(orc_index const *idx, uint32_t entry, uint64_t i);
uint64_t orc_select0(orc_index const *idx, uint32_t entry, uint64_t k);
uint64_t orc_rank0_naive(orc_index const *idx, uint32_t entry, uint64_t i);
uint64_t orc_select0_naive(orc_index const *idx, uint32_t entry, uint64_t k);
int orc_bit(orc_index const *idx, uint32_t entry, uint64_t i);
uint64_t orc_project_position(orc_index const *idx, uint32_t src, uint32_t dst, uint64_t pos);

/* rewrite_cigar.hh:18-157.  query = dna5::to_char() characters.  Returns 0 or a negative error. */
int orc_rewrite_cigar(
	orc_index const *idx, uint32_t src, uint32_t dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out, uint64_t *dst_pos
);

/* indel_run_checker.cc:13-121.  Reports every run; cigar_first/cigar_second = realigned op index range. */
int orc_indel_runs(
	uint32_t const *cigar, uint32_t n_cigar, uint64_t ref_pos,
	orc_run *runs, uint32_t *run_ref_location, uint32_t *cigar_first, uint32_t *cigar_second,
	uint32_t run_cap, uint32_t *n_runs, uint64_t *final_query_pos, uint64_t *final_ref_pos
);

/* align.hh:59-112 (+ seqan3 align_pairwise restated).  ref/query are raw chars; qual = phred values or NULL. */
int orc_align_global(
	char const *ref, uint32_t m, char const *query, uint32_t n, uint8_t const *qual,
	int32_t gap_open, int32_t gap_ext, int score_mode,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out, double *score
);

/* cigar.cc:13-51. */
uint32_t orc_collapse_cigar(uint32_t *cigar, uint32_t n);

/* alignment_projector.cc:30-154. */
int orc_project_alignment(
	orc_index const *idx, uint32_t src, uint32_t dst, uint64_t src_pos,
	uint32_t const *cigar, uint32_t n_cigar,
	char const *query, uint32_t query_len, uint8_t const *qual /* NULL = empty */,
	int32_t gap_open, int32_t gap_ext, int score_mode,
	uint32_t *out_cigar, uint32_t out_cap, uint32_t *n_out,
	orc_run *runs, uint32_t run_cap, uint32_t *n_runs, uint64_t *dst_pos
);

/* Malde score table entry (align.hh:28-37); lhs quality fixed by the caller. */
double orc_malde_score(int matches, int lhs_phred, int rhs_phred);

/* Batch driver over the same SoA batch the HIP path receives (see include/pvc3.h for field meaning).
 * Outputs are packed in record order: out_cigar_off[n+1], run_off[n+1].  Returns 0, or -2 if a
 * capacity was too small (needed sizes are then in out_cigar_off[n] / run_off[n]). */
typedef struct
{
	uint64_t n;
	uint32_t const *src_pos;
	uint32_t const *src_entry;
	uint32_t const *dst_entry;
	uint32_t const *flags;
	uint64_t const *cigar_off;	/* n + 1 */
	uint32_t const *cigar;
	uint64_t const *seq_off;	/* n + 1; byte offsets into seq4; l_seq gives the base count */
	uint32_t const *l_seq;
	uint8_t const *seq4;		/* BAM 4-bit packed, high nibble first */
	uint64_t const *qual_off;	/* n + 1 (may be NULL if no record has QUAL) */
	uint8_t const *qual;
	uint32_t const *mate_src_entry;
	uint32_t const *mate_dst_entry;
	uint32_t const *mate_pos;
} orc_batch_in;

typedef struct
{
	uint32_t *dst_pos;			/* n */
	uint32_t *mate_dst_pos;		/* n (0xFFFFFFFF when the record has no mate) */
	int32_t *status;			/* n */
	uint64_t *out_cigar_off;	/* n + 1 */
	uint32_t *out_cigar;
	uint64_t out_cigar_cap;
	uint64_t *run_off;			/* n + 1 */
	orc_run *runs;
	uint64_t run_cap;
} orc_batch_out;

int orc_batch_project(
	orc_index const *idx, orc_batch_in const *in, orc_batch_out *out,
	int32_t gap_open, int32_t gap_ext, int score_mode, int n_threads
);

#ifdef __cplusplus
}
#endif
#endif
