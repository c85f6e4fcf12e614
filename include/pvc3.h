/*
 * pvc3.h — C ABI of the MI355X-native project_alignments hot path (libpvc3_hip.so).
 *
 * Drop-in boundary for PanVC 3's libpanvc3 alignment_projector: everything the reference's
 * project_task::process() computes per record between reading a BAM record and writing it back
 * (project-alignments/project_alignments.cc:721-732 and :897-901) is computed here for a whole batch
 * of records on one GPU.  Plain pointers and sizes only; no exceptions cross the boundary; every
 * function returns PVC3_OK (0) or a negative PVC3_E_* code (pvc3_last_error() gives the message).
 *
 * Reference interface replaced by each entry point (paths relative to the reference tree):
 *   pvc3_index_add_aligned / _add_bits   msa_index::sequence_entry(seq_id, bit_vector) include/panvc3/msa_index.hh:58-65,
 *                                        make_sequence_entry_pair :214-236, index-msa/index_vector_builder.cc:29-52
 *   pvc3_index_set_reference             the dst contig buffer handed to project_alignment (ref_seq),
 *                                        project_alignments.cc:575-594, :726
 *   pvc3_project_positions               msa_index::sequence_entry::project_position include/panvc3/msa_index.hh:253-258
 *                                        (call sites project_alignments.cc:900, convert-bed-positions/main.cc:60-73)
 *   pvc3_split_match_ops                 rewrite_cigar_sequence_match + fill_ref_n_positions rewrite-cigar/main.cc:191-284, :126-148
 *   pvc3_rescore_alignments              cigar_alignment_scorer::calculate_score recalculate-mapq/main.cc:189-289
 *   pvc3_recalculate_mapq                mapq_scorer::process_alignment_group recalculate-mapq/main.cc:962-1180 and
 *                                        bowtie2_v2_score_calculator::calculate_mapq :441-566
 *   pvc3_pipeline_*                      the record loop input_processor::process_input project_alignments.cc:484-655 (:600-627)
 *   pvc3_batch_project / pvc3_dbatch_*   alignment_projector::project_alignment include/panvc3/alignment_projector.hh:58-68
 *                                        + accessors :70-74 (alignment(), realigned_reference_ranges(),
 *                                        realigned_query_ranges()), libpanvc3/alignment_projector.cc:30-154
 *
 * Threading: a pvc3_context is used by one host thread at a time; a finalized pvc3_index is read-only and
 * may be shared by several contexts (one per GPU).  There is no CPU fallback: without a GPU
 * pvc3_context_create() fails with PVC3_E_NO_DEVICE.
 */
#ifndef PVC3_H
#define PVC3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVC3_API __attribute__((visibility("default")))

#define PVC3_VERSION 100	/* 0.1.0 */

/* Return codes. */
enum
{
	PVC3_OK = 0,
	PVC3_E_INVALID = -1,		/* bad argument */
	PVC3_E_NOMEM = -2,			/* host or device allocation failed */
	PVC3_E_HIP = -3,			/* HIP runtime error */
	PVC3_E_NO_DEVICE = -4,		/* no usable GPU */
	PVC3_E_CAPACITY = -5,		/* caller-provided output buffer too small (needed sizes are reported) */
	PVC3_E_STATE = -6			/* call sequence error (e.g. index not uploaded) */
};

/* Per-record status (pvc3_batch_out.status).  A record with a non-zero status has empty outputs. */
enum
{
	PVC3_ST_OK = 0,
	PVC3_ST_SRC_RANGE = 1,		/* src_pos (+ reference-consuming length) beyond the source sequence */
	PVC3_ST_CIGAR_OP = 2,		/* CIGAR operation code > 8 (rewrite_cigar.hh:149-151 libbio_fail) */
	PVC3_ST_QUERY_RANGE = 3,	/* CIGAR consumes more query bases than l_seq (rewrite_cigar.hh:102 assert) */
	PVC3_ST_REF_RANGE = 4		/* dst text missing or too short (rewrite_cigar.hh:103 assert) */
};

/* BAM CIGAR encoding: (count << 4) | op, op codes "MIDNSHP=X" = 0..8. */
enum { PVC3_CIGAR_M = 0, PVC3_CIGAR_I, PVC3_CIGAR_D, PVC3_CIGAR_N, PVC3_CIGAR_S, PVC3_CIGAR_H, PVC3_CIGAR_P, PVC3_CIGAR_EQ, PVC3_CIGAR_X };

/* Record flags. */
enum
{
	PVC3_FLAG_HAS_QUAL = 1u,	/* base qualities present => Malde base-quality scoring (alignment_projector.cc:76, :97-137) */
	PVC3_FLAG_HAS_MATE = 2u		/* mate_* fields valid => mate_dst_pos is computed (project_alignments.cc:880-905) */
};

#define PVC3_NO_MATE 0xFFFFFFFFu

/* Realignment score type (seqan3 is not part of the reference snapshot). */
enum
{
	PVC3_SCORE_FP64 = 0,		/* Malde scores as doubles (base_quality_scorer::score_type, align.hh:26) */
	PVC3_SCORE_INT32_TRUNC = 1	/* each substitution score truncated to int32 (upstream seqan3's default score type) */
};

typedef struct pvc3_index pvc3_index;
typedef struct pvc3_context pvc3_context;
typedef struct pvc3_dbatch pvc3_dbatch;

typedef struct pvc3_params
{
	int32_t gap_open;			/* --gap-opening-cost, seqan3 open_score (cmdline.ggo:43; default 0) */
	int32_t gap_ext;			/* --gap-extension-cost, seqan3 extension_score (cmdline.ggo:44; default -1) */
	int32_t score_mode;			/* PVC3_SCORE_* */
	int32_t tiebreak_mode;		/* which of the co-optimal alignments: 2 * p + o, PVC3_TIEBREAK_MODES of them.
								   p: the order in which equal candidates of a cell are preferred — 0 diagonal > up > left (the restated
								   seqan3 order, the default), 1 diagonal > left > up, 2 up > diagonal > left, 3 up > left > diagonal,
								   4 left > diagonal > up, 5 left > up > diagonal (up = gap in the reference = 'I', left = 'D');
								   o: 0 = a gap that could be extended or opened at the same score is extended, 1 = it is opened.
								   The scores are the same under every policy.  Other than 0 is for pinning the order against the real
								   binary (tests/parity_kit/): every run then goes through the one-run-per-wavefront kernel. */
} pvc3_params;
#define PVC3_TIEBREAK_MODES 12

/* One realigned indel run (alignment_projector.cc:69-73). */
typedef struct pvc3_run
{
	uint32_t query_location;	/* realigned_query_ranges(): [location, location + length), soft-clipped bases included */
	uint32_t query_length;
	uint32_t ref_position;		/* realigned_reference_ranges().location: indel_run_checker::reference_position() at
								   report time, i.e. the position AFTER the run-terminating operation (quirk Q7) */
	uint32_t ref_length;
} pvc3_run;

/* Input batch, structure of arrays.  All pointers are host pointers for pvc3_batch_project / pvc3_dbatch_create. */
typedef struct pvc3_batch_in
{
	uint64_t n;						/* number of records */
	uint32_t const *src_pos;		/* [n] 0-based position in the unaligned source sequence (BAM POS) */
	uint32_t const *src_entry;		/* [n] index entry of the source sequence (the record's RNAME, chr<sep>seq) */
	uint32_t const *dst_entry;		/* [n] index entry of the destination row (--reference-msa-id of the same chromosome) */
	uint32_t const *flags;			/* [n] PVC3_FLAG_* */
	uint64_t const *cigar_off;		/* [n + 1] offsets into cigar */
	uint32_t const *cigar;			/* BAM-encoded operations */
	uint64_t const *seq_off;		/* [n + 1] byte offsets into seq4 (each record starts on a byte boundary, as in BAM) */
	uint32_t const *l_seq;			/* [n] number of bases */
	uint8_t const *seq4;			/* BAM 4-bit bases "=ACMGRSVTWYHKDBN", high nibble first */
	uint64_t const *qual_off;		/* [n + 1] offsets into qual; may be NULL if no record has PVC3_FLAG_HAS_QUAL */
	uint8_t const *qual;			/* raw phred values (BAM QUAL); values > 41 saturate (seqan3::phred42) */
	uint32_t const *mate_src_entry;	/* [n] index entry of the mate's source sequence (RNEXT); may be NULL */
	uint32_t const *mate_dst_entry;	/* [n] destination row of the mate's chromosome; may be NULL */
	uint32_t const *mate_pos;		/* [n] 0-based mate position (PNEXT); may be NULL */
} pvc3_batch_in;

/* Output batch.  Caller-allocated host buffers; variable-length outputs are packed in record order. */
typedef struct pvc3_batch_out
{
	uint32_t *dst_pos;				/* [n] return value of project_alignment(): new 0-based POS */
	uint32_t *mate_dst_pos;			/* [n] projected mate position, PVC3_NO_MATE if the record has no mate */
	int32_t *status;				/* [n] PVC3_ST_* */
	uint64_t *out_cigar_off;		/* [n + 1] */
	uint32_t *out_cigar;			/* alignment(): rewritten + realigned + collapsed CIGAR, BAM-encoded */
	uint64_t out_cigar_cap;			/* capacity of out_cigar in operations */
	uint64_t *run_off;				/* [n + 1] */
	pvc3_run *runs;					/* realigned ranges */
	uint64_t run_cap;
} pvc3_batch_out;

typedef struct pvc3_batch_stats
{
	uint64_t n_records;
	uint64_t n_failed;				/* records with status != 0 */
	uint64_t n_realigned_records;	/* records with at least one indel run */
	uint64_t n_runs;				/* total realigned indel runs */
	uint64_t out_cigar_total;		/* total output CIGAR operations (required out_cigar_cap) */
	uint64_t algorithmic_bytes;		/* SURVEY.md §8(d) bytes of this batch (reads + writes), see DESIGN.md */
	uint64_t algorithmic_bytes_project;	/* the share of k_project_fast alone (accumulated per record by the kernel) */
	uint64_t n_fast_path;			/* records finished by the lane-parallel fast path of k_project_fast */
	uint64_t n_mid_path;			/* records finished by k_project_mid (lane-parallel walk of the remaining records) */
	float ms_project;				/* k_project_fast: last launch duration (hipEvent), 0 if timing is disabled */
	float ms_mid;					/* k_project_mid */
	float ms_general;				/* k_project_general */
	float ms_realign;				/* k_realign */
	float ms_pack;					/* scans + gather */
	uint32_t attempts;				/* 1 unless a workspace had to grow */
} pvc3_batch_stats;

/* ---- library ---------------------------------------------------------------------------------------------------- */
PVC3_API int pvc3_version(void);
PVC3_API char const *pvc3_strerror(int code);
PVC3_API char const *pvc3_last_error(void);				/* thread-local message of the last failing call */
PVC3_API int pvc3_device_count(int *count);

/* ---- index (host side; msa_index.hh) ---------------------------------------------------------------------------- */
PVC3_API int pvc3_index_create(pvc3_index **idx);
PVC3_API void pvc3_index_destroy(pvc3_index *idx);
/* Column i is a gap iff aligned[i] == '-'.  Returns the new entry id in *entry. */
PVC3_API int pvc3_index_add_aligned(pvc3_index *idx, char const *aligned, uint64_t n_cols, uint32_t *entry);
/* Bit (i & 63) of gap_words[i >> 6] is 1 iff column i is a gap. */
PVC3_API int pvc3_index_add_bits(pvc3_index *idx, uint64_t const *gap_words, uint64_t n_cols, uint32_t *entry);
/* Unaligned text (raw FASTA bytes, case preserved) of an entry used as a destination; len >= number of non-gap columns.
 * The bytes are copied. */
PVC3_API int pvc3_index_set_reference(pvc3_index *idx, uint32_t entry, char const *text, uint64_t len);
/* The entry that `entry`'s records are projected to (the destination row of its chromosome; it must have as many columns).  The
 * paired blocks and the position windows of `entry` are made towards it.  Without this call the nearest entry with a text and as
 * many columns is taken, which is right for one MSA per column count; with several MSAs of one length a caller that knows the
 * pairs (project_alignments.cc:1271-1316 does) should name them: records whose destination is another entry than the one the
 * windows were made for are projected correctly, but by the slower kernels. */
PVC3_API int pvc3_index_set_destination(pvc3_index *idx, uint32_t entry, uint32_t dst_entry);
PVC3_API int pvc3_index_entry_count(pvc3_index const *idx, uint32_t *count);
PVC3_API int pvc3_index_entry_info(pvc3_index const *idx, uint32_t entry, uint64_t *n_cols, uint64_t *n_non_gap);
/* Host view of the HBM layout of one entry (for inspection and host-logic tests):
 * blocks = n_blocks x 8 uint64 {bits 0-31 non-gap columns before the block, bits 32-40 / 41-49 / 50-58 non-gap columns in the
 * block's data words 0 / 0..2 / 0..4; then 7 x 64 gap bits}; sel[j] = column of non-gap #(4096 j). */
PVC3_API int pvc3_index_entry_layout(pvc3_index const *idx, uint32_t entry, uint64_t const **blocks, uint64_t *n_blocks, uint32_t const **sel, uint64_t *n_sel);

/* ---- device context ------------------------------------------------------------------------------------------------ */
PVC3_API int pvc3_context_create(int device, pvc3_context **ctx);
PVC3_API void pvc3_context_destroy(pvc3_context *ctx);
/* Launch on a caller-owned hipStream_t (NULL restores the context's own stream). */
PVC3_API int pvc3_context_set_stream(pvc3_context *ctx, void *hip_stream);
PVC3_API int pvc3_context_set_timing(pvc3_context *ctx, int enabled);	/* hipEvent pairs around each kernel group */
PVC3_API int pvc3_context_synchronize(pvc3_context *ctx);
/* Tuning / test knobs: "realign_waves", "big_waves", "realign_lds_bytes", "scratch_per_wave", "scratch_budget" (bytes: the
 * alignment scratch slots together never exceed it), "trace_initial" (32-bit words of the alignment trace buffer), "arena_initial", "runs_initial",
 * "fast_path" (0: no lane-parallel compare in k_project_fast), "mid_path" (0: skip k_project_mid; with both at 0 every
 * record goes through the cooperative column walk of k_project_general), "packed_dp" (0: every indel run is aligned by the
 * striped one-run-per-wavefront kernel instead of the packed kernels), "staging_threads" (default 4: the threads that move
 * ORDINARY host arrays — not page-locked ones — through the context's page-locked staging chunks on their way to and from
 * the device; set before the first such copy); and those that take effect at the next
 * pvc3_context_upload_index and cost HBM only:
 * "position_windows" (default 1): for every entry and the entry it is projected to, one 128- or 256-byte line per 64 or 128
 *   source positions that holds what a record starting there needs of the index — start column, projected position, the
 *   number of columns without a gap in either row, the destination text of those columns.  Two bytes per source position of
 *   every entry; left out (and the look-ups done through the select samples and paired blocks alone) when they would not
 *   leave "window_reserve" bytes (default 24 GiB) of the device free.
 * "window_read_length" (default 150): the read length the windows are sized for — up to 158 bases 128-byte windows, beyond
 *   that 256-byte windows (reads of up to 255 bases are then compared from their window; longer ones are walked).
 * "paired_blocks" (default 1): every entry's bit-vector blocks also stored side by side with those of the entry it is
 *   projected to, one 128-byte HBM line for both rows (what the walk kernels read).
 * "text_window" (default 1): without position windows, the packed reference texts also stored as overlapping 128-byte lines. */
PVC3_API int pvc3_context_set_option(pvc3_context *ctx, char const *key, int64_t value);
/* Replicates the index (bit-vector blocks, select samples, reference texts, and the two copies above) into this context's HBM. */
PVC3_API int pvc3_context_upload_index(pvc3_context *ctx, pvc3_index const *idx);
PVC3_API int pvc3_context_hbm_bytes(pvc3_context const *ctx, uint64_t *index_bytes, uint64_t *workspace_bytes);

/* ---- hot path ------------------------------------------------------------------------------------------------------ */
/* Host buffers in, host buffers out: upload, project + realign, pack, download.  On PVC3_E_CAPACITY the required
 * capacities are stored in out->out_cigar_off[n] and out->run_off[n]. */
PVC3_API int pvc3_batch_project(pvc3_context *ctx, pvc3_batch_in const *in, pvc3_batch_out *out, pvc3_params const *params, pvc3_batch_stats *stats);

/* Device-resident form: inputs are uploaded once, pvc3_dbatch_project() only enqueues kernels on the context's stream
 * (inputs and outputs stay in HBM), pvc3_dbatch_finish() waits and reports, pvc3_dbatch_download() copies results out. */
PVC3_API int pvc3_dbatch_create(pvc3_context *ctx, pvc3_batch_in const *in, pvc3_dbatch **batch);
PVC3_API void pvc3_dbatch_destroy(pvc3_dbatch *batch);
PVC3_API int pvc3_dbatch_project(pvc3_context *ctx, pvc3_dbatch *batch, pvc3_params const *params);
PVC3_API int pvc3_dbatch_finish(pvc3_context *ctx, pvc3_dbatch *batch, pvc3_batch_stats *stats);
PVC3_API int pvc3_dbatch_download(pvc3_context *ctx, pvc3_dbatch *batch, pvc3_batch_out *out);
/* 64-bit checksum of all outputs, computed on the device: the wrapping sum over records of a hash of (first_record + i,
 * dst_pos, mate_dst_pos, status, CIGAR, runs).  Additive, so the checksums of shards add up to that of the whole. */
PVC3_API int pvc3_dbatch_checksum(pvc3_context *ctx, pvc3_dbatch *batch, uint64_t first_record, uint64_t *checksum);

/* Pipelined form of pvc3_batch_project for a stream of batches (the loop project_alignments.cc:600-627 runs record by
 * record): batch k + 1 is copied to the GPU while batch k is projected and the results of batch k - 1 are copied back.  The
 * device batches are allocated once, for the capacities given here.  Input and output arrays should be page-locked
 * (pvc3_host_alloc): the copies are then asynchronous, and QUAL is not copied at all — the kernels fetch it from the
 * caller's arrays for the records with an indel run only.  Arrays handed to pvc3_pipeline_submit must stay valid and
 * unchanged until pvc3_pipeline_wait has returned that batch.  Batches are returned in submission order. */
typedef struct pvc3_pipeline pvc3_pipeline;
typedef struct pvc3_pipeline_caps
{
	uint64_t max_records;			/* per batch */
	uint64_t max_cigar_ops;
	uint64_t max_seq_bytes;
	uint64_t max_qual_bytes;		/* only used for batches whose QUAL arrays are not page-locked (then QUAL is uploaded); may be 0 */
	uint32_t depth;					/* batches in flight, 2 .. 8; 0 = 3 */
} pvc3_pipeline_caps;
PVC3_API int pvc3_pipeline_create(pvc3_context *ctx, pvc3_pipeline_caps const *caps, pvc3_pipeline **pipeline);
PVC3_API void pvc3_pipeline_destroy(pvc3_pipeline *pipeline);
/* Enqueues one batch and returns; PVC3_E_STATE when `depth` batches are in flight (call pvc3_pipeline_wait), PVC3_E_CAPACITY
 * when the batch exceeds the capacities.  out->out_cigar_cap / run_cap bound the variable-length results. */
PVC3_API int pvc3_pipeline_submit(pvc3_pipeline *pipeline, pvc3_batch_in const *in, pvc3_batch_out const *out, pvc3_params const *params);
/* Waits for the oldest batch in flight: its results are in the arrays given at submission.  Returns that batch's error, if
 * it had one (PVC3_E_CAPACITY: results larger than the output arrays; PVC3_E_INVALID: malformed offsets). */
PVC3_API int pvc3_pipeline_wait(pvc3_pipeline *pipeline, pvc3_batch_stats *stats);
PVC3_API int pvc3_pipeline_in_flight(pvc3_pipeline const *pipeline, uint32_t *count);
/* Waits until every copy and kernel queued by this pipeline has run (the batches stay in flight: their results are still
 * collected with pvc3_pipeline_wait).  A caller that runs a batch of its own on the same context next to the pipeline
 * (pvc3_dbatch_*) calls this first: the context's alignment scratch and trace buffers are shared by all its batches. */
PVC3_API int pvc3_pipeline_synchronize(pvc3_pipeline *pipeline);

/* Page-locked host memory for the batch arrays (hipHostMalloc): with pageable arrays the copies of pvc3_dbatch_create /
 * pvc3_dbatch_download run at about a third of the PCIe rate.  Any thread; free with pvc3_host_free. */
PVC3_API int pvc3_host_alloc(uint64_t bytes, void **ptr);
PVC3_API void pvc3_host_free(void *ptr);

/* ---- recalculate_mapq (recalculate-mapq/main.cc; the step after project_alignments in both workflows) -------------------- */
#define PVC3_NO_POS 0xFFFFFFFFu		/* a reference id / position field or tag that is not set */
#define PVC3_MAPQ_NO_NEXT_RECORD 255	/* main.cc:51 */

/* Alignment score from the CIGAR, the base qualities and the XN bit map (cigar_alignment_scorer, main.cc:155-289). */
typedef struct pvc3_rescore_params
{
	int32_t min_mismatch_penalty;	/* --min-mismatch-penalty, default 2 */
	int32_t max_mismatch_penalty;	/* --max-mismatch-penalty, default 6 */
	int32_t n_penalty;				/* --n-penalty, default 1 */
	int32_t gap_opening_penalty;	/* --gap-opening-penalty, default 5 */
	int32_t gap_extension_penalty;	/* --gap-extension-penalty, default 3 */
} pvc3_rescore_params;
enum
{
	PVC3_RESCORE_OK = 0,
	PVC3_RESCORE_CIGAR_OP = 1,		/* an M (or unknown) operation: the CIGAR must use = and X (main.cc:275-279 libbio_fail) */
	PVC3_RESCORE_XN_RANGE = 2,		/* an X operation beyond the XN bit map (main.cc:266 libbio_always_assert_lt) */
	PVC3_RESCORE_QUERY_RANGE = 3	/* the CIGAR consumes more bases than l_seq */
};
/* score[i] = alignment score of record i (a non-positive whole number as double); status[i] = PVC3_RESCORE_*.  The arrays are
 * those of pvc3_batch_in (BAM-encoded CIGARs, 4-bit SEQ, raw phred QUAL; qual may be NULL: quality 0) plus xn / xn_off
 * [n + 1]: the bytes of each record's XN tag (B:C array; bit rp % 8 of byte rp / 8 <=> reference N at alignment-relative rp). */
PVC3_API int pvc3_rescore_alignments(pvc3_context *ctx, uint64_t n, uint64_t const *cigar_off, uint32_t const *cigar, uint64_t const *seq_off, uint8_t const *seq4,
	uint32_t const *l_seq, uint64_t const *qual_off, uint8_t const *qual, uint64_t const *xn_off, uint8_t const *xn, pvc3_rescore_params const *params, double *score, int32_t *status);

/* Records grouped by QNAME (consecutive records of one group), as mapq_scorer::process_alignment_group sees them
 * (main.cc:962-1180).  Unmapped records are not part of the input (main.cc:1236-1238). */
typedef struct pvc3_mapq_batch
{
	uint64_t n_groups;
	uint64_t const *group_off;			/* [n_groups + 1] record ranges of the groups */
	uint32_t const *ref_id, *pos;		/* [n] projected RNAME (numeric id) and 0-based POS; PVC3_NO_POS where unset */
	uint32_t const *mate_ref_id, *mate_pos;				/* RNEXT / PNEXT */
	uint32_t const *orig_ref_id, *orig_pos;				/* --original-rname-tag / --original-pos-tag (XR / XP); PVC3_NO_POS where missing */
	uint32_t const *orig_mate_ref_id, *orig_mate_pos;	/* --original-rnext-tag / --original-pnext-tag (YR / YP) */
	double const *score;				/* the AS value (or the rescored one); -DBL_MAX where the record has no AS tag (main.cc:177-178) */
	uint32_t const *l_seq;
	uint32_t const *seq_class;			/* equal inside a group <=> equal SEQ (consulted for groups of unpaired records only) */
} pvc3_mapq_batch;
typedef struct pvc3_mapq_stats			/* the counters the reference prints at the end (main.cc:1268-1276) */
{
	uint64_t total_alignments, unpaired_alignments, reads_with_and_without_mate, mate_not_found, reads_without_valid_position;
} pvc3_mapq_stats;
/* mapq[i] = new MAPQ of record i (PVC3_MAPQ_NO_NEXT_RECORD as in the reference); group_emitted[g] = 0 for the groups the
 * reference skips (records with and without a mate under one name, main.cc:1039-1044), whose records are not written. */
PVC3_API int pvc3_recalculate_mapq(pvc3_context *ctx, pvc3_mapq_batch const *in, uint8_t *mapq, uint8_t *group_emitted, pvc3_mapq_stats *stats);

/* project_position for n independent positions: out[i] = rank0_dst(select0_src(1 + pos[i])); PVC3_NO_MATE if out of range. */
PVC3_API int pvc3_project_positions(pvc3_context *ctx, uint64_t n, uint32_t const *src_entry, uint32_t const *dst_entry, uint32_t const *pos, uint32_t *out);

/* ---- the rewrite_cigar tool's sequence-match mode (rewrite-cigar/main.cc:191-284): every M of a record's CIGAR is split into = / X
 * by comparing seqan3::dna5::to_char of the read's bases with the raw bytes of the record's reference sequence (runs inside one M
 * are not merged with the operations around it), every other operation is kept; and the positions of 'N' in the reference from the
 * alignment's start on, over the bases the CIGAR consumes, as a bit map (:126-148: bit p % 8 of byte p / 8; (consumed + 7) / 8
 * bytes, none if the CIGAR consumes no reference base).  The reference sequences live in HBM (pvc3_reference). ---- */
typedef struct pvc3_reference pvc3_reference;
PVC3_API int pvc3_reference_create(pvc3_context *ctx, uint32_t n_sequences, uint8_t const *const *bytes, uint64_t const *lengths, pvc3_reference **out);
PVC3_API void pvc3_reference_destroy(pvc3_reference *ref);
enum
{
	PVC3_SPLIT_OK = 0,
	PVC3_SPLIT_BAD_OP = 1,			/* operation code > 8 (main.cc:279-281 libbio_fail) */
	PVC3_SPLIT_QUERY_RANGE = 2,		/* an M reaches past the end of SEQ (main.cc:249 libbio_assert_lte) */
	PVC3_SPLIT_EMPTY_M = 3,			/* an M of no bases (the reference reads query_seq[query_pos] regardless: undefined there, refused here) */
	PVC3_SPLIT_REF_RANGE = 4		/* the alignment reaches past the end of its reference sequence, or the sequence does not exist */
};
typedef struct pvc3_split_in
{
	uint64_t n;
	uint32_t const *sequence;		/* [n] which of the reference's sequences */
	uint32_t const *pos;			/* [n] 0-based start of the alignment in it */
	uint64_t const *cigar_off;		/* [n + 1] */
	uint32_t const *cigar;			/* BAM encoding: count << 4 | op */
	uint64_t const *seq_off;		/* [n] BYTE offset of the record's packed bases in seq4 (a record's bases begin a byte, as in BAM) */
	uint8_t const *seq4;			/* 4-bit codes, high nibble first; seq4_bytes bytes */
	uint64_t seq4_bytes;
	uint32_t const *l_seq;			/* [n] */
} pvc3_split_in;
typedef struct pvc3_split_out
{
	uint64_t *cigar_off;			/* [n + 1] */
	uint32_t *cigar; uint64_t cigar_cap;
	uint64_t *n_map_off;			/* [n + 1] */
	uint8_t *n_map; uint64_t n_map_cap;
	int32_t *status;				/* [n] PVC3_SPLIT_*; a record with a status has no operations and no map */
} pvc3_split_out;
/* PVC3_E_CAPACITY: *n_cigar operations / *n_map_bytes bytes are needed (both are always set); nothing else was written. */
PVC3_API int pvc3_split_match_ops(pvc3_context *ctx, pvc3_reference const *ref, pvc3_split_in const *in, pvc3_split_out *out, uint64_t *n_cigar, uint64_t *n_map_bytes);

#ifdef __cplusplus
}
#endif
#endif
