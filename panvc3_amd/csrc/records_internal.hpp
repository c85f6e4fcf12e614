// records_internal.hpp — shared by records.hip (kernels) and records_api.cpp (the C ABI of include/pvc3_records.h).
#pragma once

#include <cstdint>

#include "../../include/pvc3_records.h"

namespace pvc3 {

// Counters and small results of one piece, on the device; the host reads a copy after every step.
struct rec_state
{
	uint32_t n_records;				// whole records in the piece (those carried over from the last piece included)
	uint32_t tail_off;				// where the incomplete last record begins
	uint32_t n_kept, n_take, n_deferred;
	uint32_t carry_len;
	uint32_t frame_error_at;		// 0xFFFFFFFF, or where the chain of records met a block_size below 32
	uint32_t pad;
	unsigned long long err_key;		// min over the malformed records of (input index << 8 | PVC3_REC_E_*); ~0 if none
	unsigned long long counts[4];	// matched, flags not matched, reference id missing, sequence missing (new records only)
	unsigned long long pending_idx[2][4];	// input indices of the records that wait for their chunk of four: [parity] this piece's, [parity ^ 1] the next one's
	uint32_t deferred_off[4];
};

// The new tags in key order (the host sorts them); which apply to a record is decided per record.
enum { NEW_OA = 0, NEW_XB, NEW_YB, NEW_XI, NEW_XR, NEW_XP, NEW_YR, NEW_YP };

struct rec_tables
{
	uint32_t const *src_entry;		// [n_refs]
	uint32_t const *dst_entry;		// [n_refs]
	int32_t const *out_ref_id;		// [n_refs]
	char const *ref_names;
	uint32_t const *ref_name_off;	// [n_refs + 1]
	uint32_t const *preserved;		// 65536 bits
	unsigned long long *removed_counts;	// [65536]
	uint32_t n_new;
	uint32_t new_key[8], new_kind[8];
};

struct rec_frame_work
{
	uint32_t *guess, *cnt, *exits, *tails, *rec_base;	// [blocks]
	uint32_t *rec_off;				// [records] where each record's block_size field lies
	uint32_t *keep;					// [records]
	uint64_t *kept_pos;				// [records + 1]
	uint32_t *kept_off;				// [batch]
	uint64_t *kept_input;			// [batch] index among all input records
	uint32_t *n_cigar;				// [batch]
};

struct rec_batch_arrays
{
	uint32_t *src_pos, *src_entry, *dst_entry, *flags, *l_seq, *mate_se, *mate_de, *mate_pos;
	uint64_t *seq_off, *qual_off;
};

struct rec_results
{
	uint32_t const *dst_pos, *mate_dst_pos, *rec_flags;
	uint64_t const *out_cigar_off;
	uint32_t const *out_cigar;
	uint64_t const *run_off;
	pvc3_run const *runs;
};

int rec_launch_reset(rec_state *st, void *stream);
int rec_launch_frame(uint8_t const *U, uint32_t const *blk_off, uint32_t n_blocks, uint32_t start0, uint32_t end, uint32_t n_refs, rec_frame_work const &w, rec_state *st, void *stream);
int rec_launch_scan(uint32_t const *in, uint32_t const *n_ptr, uint32_t n_max, uint64_t *out, uint32_t *tile_sums, uint64_t *tile_base, void *stream);
uint32_t rec_scan_tiles(uint32_t n_max);
int rec_launch_filter(uint8_t const *U, uint32_t n_max, rec_frame_work const &w, rec_state *st, uint32_t n_refs, uint32_t n_pending, uint32_t parity, uint64_t seen, void *stream);
int rec_launch_compact(uint32_t n_max, rec_frame_work const &w, rec_state *st, uint32_t n_pending, uint32_t parity, uint64_t seen, int last, void *stream);
int rec_launch_carry(uint8_t const *U, uint32_t end, rec_state *st, uint8_t *carry, uint32_t carry_cap, void *stream);
int rec_launch_scatter(uint8_t const *U, uint32_t end, uint32_t n_take, rec_frame_work const &w, rec_state const *st, rec_tables const &T, rec_batch_arrays const &B, void *stream);
int rec_launch_scatter_cigar(uint8_t const *U, uint32_t n_take, rec_frame_work const &w, rec_state const *st, uint64_t const *cigar_off, uint64_t cigar_cap, uint32_t *cigar, void *stream);
int rec_launch_rewrite(bool emit, uint8_t const *U, uint32_t n_take, rec_frame_work const &w, rec_state *st, rec_tables const &T, rec_results const &R, uint32_t *out_size, uint64_t const *out_off,
	uint8_t *out, uint32_t block_bytes, uint32_t block_stride, uint32_t block_pad, void *stream);

}
