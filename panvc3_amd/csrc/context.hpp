// context.hpp — the objects behind the handles of include/pvc3.h (shared by api.cpp and records_api.cpp; not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "pvc3_internal.hpp"
#include "staging.hpp"

using namespace pvc3;

struct pvc3_context
{
	int device{};
	hipStream_t own_stream{};
	hipStream_t stream{};
	bool timing{};
	hipEvent_t ev[8]{};

	// Index replica in HBM.
	std::vector <void *> index_allocs;
	dev_entry *d_entries{};
	uint32_t n_entries{};
	uint64_t index_bytes{};
	uint64_t workspace_bytes{};

	dev_params *d_params{};
	pvc3_params params{0, 0, -1, 0};

	// Options.
	bool gap_dense{};					// the index has rows with a gap column in sixteen or more (pvc3_context_upload_index)
	uint32_t realign_waves{8192};		// packed alignment kernels: one 64-thread block each
	uint32_t big_waves{1024};			// striped alignment kernel (runs that do not fit the packed kernels): waves = scratch slots
	uint32_t realign_lds_bytes{};
	uint64_t scratch_per_wave{128 * 1024};
	uint64_t scratch_budget{UINT64_C(8) << 30};	// the scratch slots together never exceed this: fewer waves when one run needs a large slot
	uint64_t arena_initial{};
	uint64_t runs_initial{};
	uint32_t fast_path{1};
	uint32_t mid_path{1};
	uint32_t packed_dp{1};
	uint32_t paired_blocks{1};			// upload_index: paired copies of the gap bit-vector blocks (dev_entry::paired)
	uint32_t text_window{1};			// upload_index: windowed copies of the packed texts (dev_entry::text4w; only without position windows)
	uint32_t position_windows{1};		// upload_index: position windows (dev_entry::pwin) when they fit the device
	uint32_t window_read_length{150};	// longest read whose compare is to be served from one window: up to 158 bases -> 128-byte windows, else 256
	uint64_t window_reserve{UINT64_C(24) << 30};	// HBM that must stay free after the windows (batches, arenas, trace)
	pvc3_dbatch *kept_batch{};			// pvc3_batch_project keeps its device batch for the next call (no hipMalloc / hipFree per batch)
	bool have_windows{};				// the uploaded index has them: k_project_fast runs, otherwise k_project_fast_probes
	bool wide_windows{};				// ... of 256 bytes
	uint8_t *scratch{};
	uint64_t scratch_alloc{};
	uint32_t *trace{};					// trace words of the packed alignment kernels
	uint64_t trace_cap{};				// in words
	uint64_t trace_initial{};			// option: words to start with (0: sized by the batch)
	host_stager stager;					// ordinary (pageable) host arrays travel through its page-locked chunks (staging.hpp)
};

struct batch_caps
{
	uint64_t n{}, cigar{}, seq{}, qual{};		// records, CIGAR operations, SEQ bytes, QUAL bytes uploaded with the batch
	uint64_t qual_compact{};					// bytes of the compact QUAL buffer of a pipelined batch (0: QUAL travels with the batch)
	bool mates{};
};

struct pvc3_dbatch
{
	pvc3_context *ctx{};
	uint64_t n{};
	dev_batch_in in{};
	dev_batch_work work{};
	dev_batch_packed packed{};
	uint64_t *scan_tmp{};
	unsigned long long *d_checksum{};
	std::vector <void *> allocs;		// everything except the growable arenas
	uint64_t bytes{};
	batch_caps caps{};
	// optional inputs: device arrays exist when the capacities ask for them, `in` points to them when the batch has them
	uint64_t *up_qual_off{};
	uint8_t *up_qual{};
	uint32_t *up_mate_src_entry{}, *up_mate_dst_entry{}, *up_mate_pos{};
	uint32_t *d_invalid{};				// the first malformed record, 0xFFFFFFFF if the layout is sound (k_validate_batch)
	bool projected{};
	bool finished{};
	bool timed{};						// the context's timing events bracket THIS batch's kernels (never in a pipeline: the next batch re-records them)
	pvc3_params params{};
	uint32_t attempts{};
	dev_counters counters{};
	uint64_t totals[2]{};				// packed CIGAR operations and runs of the finished batch
};


namespace pvc3 {

int set_error(int code, char const *fmt, ...);

// For the device record path (records_api.cpp): a batch object whose input arrays are filled by kernels instead of copies.
int batch_object_alloc(pvc3_context *ctx, uint64_t max_records, uint64_t max_cigar_ops, pvc3_dbatch **out);
int batch_object_project(pvc3_context *ctx, pvc3_dbatch *b, pvc3_params const *params);			// enqueues the projection on ctx->stream
int batch_object_complete(pvc3_context *ctx, pvc3_dbatch *b, pvc3_batch_stats *stats);		// waits, grows and repeats if a workspace overflowed

}
