// host_state.hpp — host-side objects behind the opaque ABI handles.
#pragma once

#include "pvc3_internal.hpp"

#include <deque>

struct pvc3_index
{
	std::deque <pvc3::host_entry> entries;
};

namespace pvc3 {

// Records the message for pvc3_last_error() and returns `code`.
int set_error(int code, char const *fmt, ...) __attribute__((format(printf, 2, 3)));

int index_add_aligned(pvc3_index *idx, char const *aligned, uint64_t n_cols, uint32_t *entry);
int index_add_bits(pvc3_index *idx, uint64_t const *gap_words, uint64_t n_cols, uint32_t *entry);
int index_set_reference(pvc3_index *idx, uint32_t entry, char const *text, uint64_t len);

}
