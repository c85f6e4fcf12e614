/*
 * pvc3_records.h — C ABI of the record path of project_alignments on the GPU (libpvc3_hip.so).
 *
 * pvc3.h takes and returns structure-of-array batches; somebody has to cut them out of BAM records and to put the results
 * back into BAM records.  In the reference that is seqan3's BAM reader / writer plus the body of project_task::process();
 * here it is done on the GPU from the inflated bytes of the file, so that between the two BGZF codecs (include/pvc3_bgzf.h)
 * the records never leave the device:
 *
 *   pvc3_records_frame     the records of a piece of the uncompressed BAM stream found (block_size chains), filtered and
 *                          scattered into the batch arrays of pvc3.h in place — SEQ and QUAL are not copied, their offsets
 *                          point into the stream.
 *                          Replaces project-alignments/project_alignments.cc:512 (the record loop over seqan3's reader),
 *                          :528-570 (filters), :600-627 (four records per task; the deep copy of :627 does not exist here).
 *   pvc3_records_project   pvc3_dbatch_project + _finish on those arrays (project_alignments.cc:721-732, :897-901).
 *   pvc3_records_rewrite   every record written anew from its old bytes and its results: POS, bin, CIGAR, RNAME / RNEXT /
 *                          PNEXT, the tag filter, OA, the realigned ranges, XI / XR / XP / YR / YP, tags in key order, SEQ
 *                          through dna5, QUAL through phred42 — as BAM bytes laid out for the BGZF compressor.
 *                          Replaces project_alignments.cc:760-918 (tag and field rewrite) and :969 (seqan3's BAM output).
 *
 * One pvc3_records per pvc3_context, used by one host thread; every call is ordered on the context's stream and returns
 * after the device has finished it (the host needs the counts).  Returns PVC3_OK or a negative PVC3_E_* code
 * (pvc3_last_error()).  No CPU fallback.
 */
#ifndef PVC3_RECORDS_H
#define PVC3_RECORDS_H

#include "pvc3.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvc3_records pvc3_records;

#define PVC3_RECORDS_SLOTS 3u			/* stream buffers: one being inflated into, one being worked on, one of whose records the compressor still reads nothing but whose successor is already being inflated */
#define PVC3_RECORDS_MAX_TAGS 32u		/* preserved tags per record at most (more: PVC3_REC_E_TAGS) */

typedef struct pvc3_records_config
{
	uint32_t n_refs;					/* @SQ lines of the input header */
	uint32_t const *src_entry;			/* [n_refs] index entry of the input reference (project_alignments.cc:1271-1316) */
	uint32_t const *dst_entry;			/* [n_refs] index entry of the destination row of its chromosome */
	int32_t const *out_ref_id;			/* [n_refs] reference id in the output header (process_headers, :1092-1253) */
	char const *ref_names;				/* the input reference names back to back (for OA, :805-844) */
	uint32_t const *ref_name_off;		/* [n_refs + 1] */
	uint16_t const *preserved_tags;		/* keys c0 * 256 + c1 of the tags that are kept (:53-87 and --preserve-tag) */
	uint32_t n_preserved_tags;
	/* Tag names, two characters each, "\0\0" = not written (cmdline.ggo:28-41): original RNAME / POS / RNEXT / PNEXT,
	 * realigned query / reference ranges, record index.  Must be pairwise different. */
	char rname_tag[2], pos_tag[2], rnext_tag[2], pnext_tag[2], query_ranges_tag[2], ref_ranges_tag[2], rec_idx_tag[2];
	uint64_t max_stream_bytes;			/* bytes of uncompressed BAM per piece at most (what one pvc3_records_frame sees) */
	uint64_t max_records;				/* records per piece at most */
} pvc3_records_config;

typedef struct pvc3_records_frame_result
{
	uint64_t n_new_records;				/* whole records that began in this piece (filtered ones included) */
	uint64_t n_batch;					/* records of the batch now on the device: those that pass the filters, a multiple of four unless `last` */
	uint64_t carry_bytes;				/* bytes kept for the next piece: up to three records that wait for their chunk of four, and the incomplete last record */
	uint64_t matched, flags_not_matched, ref_id_missing, seq_missing;	/* the counters of project_alignments.cc:1047-1056 for the new records */
} pvc3_records_frame_result;

enum
{
	PVC3_REC_OK = 0,
	PVC3_REC_E_TRUNCATED = 1,			/* block_size < 32, or a record's name, CIGAR, SEQ and QUAL exceed its block_size */
	PVC3_REC_E_REF_ID = 2,				/* refID or next_refID outside the header */
	PVC3_REC_E_STATUS = 3,				/* a record could not be projected (status != 0; the reference asserts there) */
	PVC3_REC_E_CIGAR_COUNT = 4,			/* more than 65535 operations after projection */
	PVC3_REC_E_MATE = 5,				/* the mate position lies outside its sequence */
	PVC3_REC_E_AUX = 6,					/* malformed tag data */
	PVC3_REC_E_TAGS = 7,				/* more than PVC3_RECORDS_MAX_TAGS preserved tags in one record */
	PVC3_REC_E_CARRY = 8				/* a record larger than the carry buffer */
};

PVC3_API int pvc3_records_create(pvc3_context *ctx, pvc3_records_config const *config, pvc3_records **out);
PVC3_API void pvc3_records_destroy(pvc3_records *r);

/* Device memory of stream buffer `slot` (0 .. PVC3_RECORDS_SLOTS - 1): the caller has a piece's inflated bytes written to
 * *base + *front .. (device pointers; at most max_stream_bytes of them); the bytes in front of that belong to this object (the
 * carry is put there). */
PVC3_API int pvc3_records_buffer(pvc3_records *r, uint32_t slot, void **base, uint64_t *front);

/* n_bytes of host memory copied to base + front + offset of `slot` (a caller that inflates on the host, a block the GPU's
 * inflater refused, tests); ordered on the context's stream like everything else here; the source may be changed when the call
 * has returned. */
PVC3_API int pvc3_records_upload(pvc3_records *r, uint32_t slot, uint64_t offset, void const *bytes, uint64_t n_bytes);

/* The piece in `slot`: n_bytes of the stream at base + front, which the BGZF blocks cut at block_off[0] = 0 < block_off[1] < ...
 * < block_off[n_blocks] = n_bytes (host array; where the file's blocks begin is where records are looked for first).
 * skip: bytes at the beginning that are not records (the BAM header, first piece only).  last: nothing follows.
 * On PVC3_E_INVALID with *error_code (a PVC3_REC_E_* value) and *error_record (index among all input records) set, a record
 * is malformed. */
PVC3_API int pvc3_records_frame(pvc3_records *r, uint32_t slot, uint64_t n_bytes, uint64_t const *block_off, uint32_t n_blocks, uint64_t skip, int last,
	pvc3_records_frame_result *result, int *error_code, uint64_t *error_record);

/* Projects the batch that pvc3_records_frame left on the device (pvc3_dbatch_project + pvc3_dbatch_finish). */
PVC3_API int pvc3_records_project(pvc3_records *r, pvc3_params const *params, pvc3_batch_stats *stats);

/* Writes the batch's records as BAM bytes (block_size prefix included) for the BGZF compressor: byte o of the output goes to
 * d_out[(o / block_bytes) * block_stride + block_pad + o % block_bytes] (pvc3_bgzf.h: 65280, 65536, 64).  *n_bytes: their number
 * (PVC3_E_CAPACITY if more than capacity_bytes; nothing is written then).  n_runs (may be null): realigned ranges of the batch. */
PVC3_API int pvc3_records_rewrite(pvc3_records *r, void *d_out, uint64_t capacity_bytes, uint32_t block_bytes, uint32_t block_stride, uint32_t block_pad,
	uint64_t *n_bytes, uint64_t *n_runs, int *error_code, uint64_t *error_record);
/* The same into a buffer of this object, contiguous, and from there to host memory (a caller that compresses on the host, and
 * tests): dst holds capacity_bytes; *n_bytes as above. */
PVC3_API int pvc3_records_rewrite_to_host(pvc3_records *r, void *dst, uint64_t capacity_bytes, uint64_t *n_bytes, uint64_t *n_runs, int *error_code, uint64_t *error_record);

/* (location, length) of the batch's realigned ranges in record order (alignment_projector::realigned_reference_ranges, for
 * --output-realigned-ranges); `runs` holds the n_runs pvc3_records_rewrite reported. */
PVC3_API int pvc3_records_runs(pvc3_records *r, pvc3_run *runs, uint64_t n_runs);

/* counts[key] += number of tags with that key removed so far (65536 counters; project_alignments.cc:1058-1069). */
PVC3_API int pvc3_records_removed_tags(pvc3_records *r, uint64_t *counts);

/* Test access: the batch arrays pvc3_records_frame made, copied to the host (pvc3_batch_in with caller-allocated arrays of the
 * sizes below; seq_off / qual_off are offsets from the stream buffer's base) and the records' places and input indices. */
PVC3_API int pvc3_records_batch_sizes(pvc3_records *r, uint64_t *n, uint64_t *n_cigar);
PVC3_API int pvc3_records_download_batch(pvc3_records *r, uint32_t *src_pos, uint32_t *src_entry, uint32_t *dst_entry, uint32_t *flags, uint64_t *cigar_off, uint32_t *cigar,
	uint64_t *seq_off, uint32_t *l_seq, uint64_t *qual_off, uint32_t *mate_src_entry, uint32_t *mate_dst_entry, uint32_t *mate_pos, uint64_t *record_off, uint64_t *input_index);

#ifdef __cplusplus
}
#endif
#endif
