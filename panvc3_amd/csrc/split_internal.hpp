// split_internal.hpp — what split_api.cpp and split_ops.hip share (pvc3_split_match_ops).
#pragma once
#include <cstdint>

namespace pvc3 {

struct split_work
{
	uint64_t n;
	uint32_t n_sequences;
	uint8_t const *const *ref_bytes;		// [n_sequences] device pointers
	uint64_t const *ref_len;
	uint32_t const *sequence, *pos, *l_seq;
	uint64_t const *cigar_off, *seq_off;
	uint32_t const *cigar;
	uint8_t const *seq4;
	int32_t *status;
	uint32_t *n_out, *n_map;				// counts of the first pass
	uint64_t const *out_cigar_off, *out_map_off;		// their scans
	uint32_t *out_cigar;
	uint8_t *out_map;
};

int launch_split_match_ops(bool emit, split_work const &W, void *stream);

}
