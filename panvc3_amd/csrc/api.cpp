// api.cpp — the C ABI of include/pvc3.h over the HIP kernels.  No CPU fallback anywhere: every compute entry point
// needs a pvc3_context, and pvc3_context_create() fails without a GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <string>
#include <vector>

#include "host_state.hpp"
#include "pvc3_internal.hpp"
#include "staging.hpp"
#include "context.hpp"

using namespace pvc3;

namespace pvc3 {

static thread_local char g_error[768] = "";

int set_error(int code, char const *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	return code;
}

}

#define HIP_TRY(expr) do { hipError_t const e_ = (expr); if (hipSuccess != e_) return set_error((hipErrorOutOfMemory == e_) ? PVC3_E_NOMEM : PVC3_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define PVC3_TRY(expr) do { int const rc_ = (expr); if (rc_) return rc_; } while (0)

namespace {

int check_device(pvc3_context *ctx)
{
	HIP_TRY(hipSetDevice(ctx->device));
	return PVC3_OK;
}

template <typename t_value>
int dev_alloc(t_value **dst, uint64_t count, std::vector <void *> *track, uint64_t *bytes_acc)
{
	void *p{};
	uint64_t const bytes = (count ? count : 1) * sizeof(t_value);
	HIP_TRY(hipMalloc(&p, bytes));
	*dst = (t_value *) p;
	if (track) track->push_back(p);
	if (bytes_acc) *bytes_acc += bytes;
	return PVC3_OK;
}

template <typename t_value>
int dev_upload(t_value const **dst, t_value const *src, uint64_t count, uint64_t pad_bytes, hipStream_t stream, std::vector <void *> *track, uint64_t *bytes_acc)
{
	uint8_t *p{};
	PVC3_TRY(dev_alloc(&p, count * sizeof(t_value) + pad_bytes, track, bytes_acc));
	if (count)
		HIP_TRY(hipMemcpyAsync(p, src, count * sizeof(t_value), hipMemcpyHostToDevice, stream));
	if (pad_bytes)
		HIP_TRY(hipMemsetAsync(p + count * sizeof(t_value), 0, pad_bytes, stream));
	*dst = (t_value const *) p;
	return PVC3_OK;
}

// align.hh:28-37, evaluation order as written; reference base quality fixed at 41 (alignment_projector.cc:108).
double malde_score(bool matches, int lhs_phred, int rhs_phred)
{
	double const lhs_e = 1.0 / std::pow(10.0, lhs_phred / 10.0);
	double const rhs_e = 1.0 / std::pow(10.0, rhs_phred / 10.0);
	double const combined_e = lhs_e + rhs_e - lhs_e / 3.0 * rhs_e * 4.0;
	if (matches)
		return 2.0 + std::log2(1.0 - combined_e);
	else
		return 2.0 - std::log2(3.0) + std::log2(combined_e);
}

int upload_params(pvc3_context *ctx, pvc3_params const &p)
{
	if (ctx->d_params && 0 == std::memcmp(&p, &ctx->params, sizeof(p)))
		return PVC3_OK;
	if (p.score_mode != PVC3_SCORE_FP64 && p.score_mode != PVC3_SCORE_INT32_TRUNC)
		return set_error(PVC3_E_INVALID, "pvc3_params.score_mode %d is not a PVC3_SCORE_* value", p.score_mode);
	if (p.tiebreak_mode < 0 || p.tiebreak_mode >= PVC3_TIEBREAK_MODES)
		return set_error(PVC3_E_INVALID, "pvc3_params.tiebreak_mode %d: 0 .. %d are defined", p.tiebreak_mode, PVC3_TIEBREAK_MODES - 1);
	dev_params hp{};
	hp.gap_open = p.gap_open; hp.gap_ext = p.gap_ext; hp.score_mode = p.score_mode; hp.tiebreak_mode = p.tiebreak_mode;
	for (int q = 0; q <= 41; ++q)
	{
		hp.s_match[q] = malde_score(true, 41, q);
		hp.s_mismatch[q] = malde_score(false, 41, q);
		if (PVC3_SCORE_INT32_TRUNC == p.score_mode)
		{
			hp.s_match[q] = (double) (int32_t) hp.s_match[q];
			hp.s_mismatch[q] = (double) (int32_t) hp.s_mismatch[q];
		}
	}
	if (!ctx->d_params)
		PVC3_TRY(dev_alloc(&ctx->d_params, 1, nullptr, &ctx->workspace_bytes));
	// Stream-ordered so that an in-flight projection with the previous parameters is not disturbed.
	HIP_TRY(hipMemcpyAsync(ctx->d_params, &hp, sizeof(hp), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	ctx->params = p;
	return PVC3_OK;
}

void free_index_replica(pvc3_context *ctx)
{
	for (void *p : ctx->index_allocs) (void) hipFree(p);
	ctx->index_allocs.clear();
	ctx->d_entries = nullptr;
	ctx->n_entries = 0;
	ctx->index_bytes = 0;
}

// Waves of the striped alignment kernel: one scratch slot each, all slots within the budget.  A single long run (e.g. a
// 250-base query slice against a 60 kb deletion needs 16 MB) must not multiply by thousands of waves.
uint32_t big_wave_count(pvc3_context const *ctx)
{
	uint64_t const fit = ctx->scratch_budget / (ctx->scratch_per_wave ? ctx->scratch_per_wave : 1);
	uint64_t const w = (fit < ctx->big_waves) ? fit : ctx->big_waves;
	return (uint32_t) (w < 1 ? 1 : w);
}

int ensure_scratch(pvc3_context *ctx)
{
	uint64_t const need = ctx->scratch_per_wave * big_wave_count(ctx);
	if (ctx->scratch && ctx->scratch_alloc >= need)
		return PVC3_OK;
	if (ctx->scratch) { (void) hipFree(ctx->scratch); ctx->workspace_bytes -= ctx->scratch_alloc; ctx->scratch = nullptr; }
	PVC3_TRY(dev_alloc(&ctx->scratch, need, nullptr, &ctx->workspace_bytes));
	ctx->scratch_alloc = need;
	return PVC3_OK;
}

// The packed alignment kernels leave 32 bytes of trace per anti-diagonal step of a task; config C5 (5 % of 250-base reads over
// a 20 - 60 base event) needs about 95 bytes per record of the batch.  Kept by the context and reused by every batch.
constexpr uint64_t TRACE_FRONT = 16;

int ensure_trace(pvc3_context *ctx, uint64_t words)
{
	if (ctx->trace && ctx->trace_cap >= words)
		return PVC3_OK;
	if (ctx->trace) { (void) hipFree(ctx->trace - TRACE_FRONT); ctx->workspace_bytes -= (ctx->trace_cap + 2 * TRACE_FRONT) * sizeof(uint32_t); ctx->trace = nullptr; ctx->trace_cap = 0; }
	uint32_t *p{};
	PVC3_TRY(dev_alloc(&p, words + 2 * TRACE_FRONT, nullptr, &ctx->workspace_bytes));		// k_realign_traceback reads 16-byte windows that may begin before a task's first word
	ctx->trace = p + TRACE_FRONT;
	ctx->trace_cap = words;
	return PVC3_OK;
}

int alloc_arenas(pvc3_dbatch *b, uint64_t arena_cap, uint64_t runs_cap)
{
	pvc3_context *ctx = b->ctx;
	if (b->work.arena) { (void) hipFree(b->work.arena); (void) hipFree(b->packed.out_cigar); b->bytes -= 8 * b->work.arena_cap; }
	if (b->work.runs_arena)
	{
		(void) hipFree(b->work.runs_arena); (void) hipFree(b->packed.runs); (void) hipFree(b->work.run_items); (void) hipFree(b->work.dp_list);
		b->bytes -= (2 * sizeof(pvc3_run) + sizeof(run_work) + DP_CLASSES * sizeof(uint32_t)) * b->work.runs_cap;
	}
	b->work.arena = nullptr; b->packed.out_cigar = nullptr; b->work.runs_arena = nullptr; b->packed.runs = nullptr; b->work.run_items = nullptr; b->work.dp_list = nullptr;
	PVC3_TRY(dev_alloc(&b->work.arena, arena_cap + 16, nullptr, &b->bytes));		// 16-byte reads of a record's operations may pass its last one
	PVC3_TRY(dev_alloc(&b->packed.out_cigar, arena_cap, nullptr, &b->bytes));
	PVC3_TRY(dev_alloc(&b->work.runs_arena, runs_cap, nullptr, &b->bytes));
	PVC3_TRY(dev_alloc(&b->packed.runs, runs_cap, nullptr, &b->bytes));
	PVC3_TRY(dev_alloc(&b->work.run_items, runs_cap, nullptr, &b->bytes));
	PVC3_TRY(dev_alloc(&b->work.dp_list, DP_CLASSES * runs_cap, nullptr, &b->bytes));
	b->work.arena_cap = arena_cap; b->packed.out_cigar_cap = arena_cap;
	b->work.runs_cap = runs_cap; b->packed.run_cap = runs_cap;
	(void) ctx;
	return PVC3_OK;
}

int enqueue_projection(pvc3_context *ctx, pvc3_dbatch *b)
{
	hipStream_t const s = ctx->stream;
	b->work.scratch = ctx->scratch;
	b->work.scratch_per_wave = ctx->scratch_per_wave;
	PVC3_TRY(ensure_trace(ctx, ctx->trace_initial ? ctx->trace_initial : ((UINT64_C(16) << 20) + 25 * b->n)));		// 64 MB + 100 bytes per record
	b->work.trace = ctx->trace;
	b->work.trace_cap = ctx->trace_cap;
	HIP_TRY(hipMemsetAsync(b->work.counters, 0, sizeof(dev_counters), s));
	HIP_TRY(hipMemsetAsync(b->work.wave_stats, 0, 3 * (uint64_t) project_wave_count(b->n) * WAVE_STATS * sizeof(unsigned long long), s));
	b->timed = ctx->timing;
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[0], s));
	if (int const e = ctx->have_windows ? launch_project_fast(ctx->d_entries, ctx->n_entries, b->in, b->work, ctx->fast_path, ctx->wide_windows, ctx->gap_dense, s)
			: launch_project_probes(ctx->d_entries, ctx->n_entries, b->in, b->work, ctx->fast_path, s))
		return set_error(PVC3_E_HIP, "k_project_fast launch: %s", hipGetErrorString((hipError_t) e));
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[1], s));
	if (ctx->mid_path)
		if (int const e = launch_project_mid(ctx->d_entries, ctx->n_entries, b->in, b->work, s)) return set_error(PVC3_E_HIP, "k_project_mid launch: %s", hipGetErrorString((hipError_t) e));
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[5], s));
	if (int const e = launch_project_general(ctx->d_entries, ctx->n_entries, b->in, b->work, s)) return set_error(PVC3_E_HIP, "k_project_general launch: %s", hipGetErrorString((hipError_t) e));
	if (int const e = launch_reduce_stats(b->n, b->work, s)) return set_error(PVC3_E_HIP, "k_reduce_stats launch: %s", hipGetErrorString((hipError_t) e));
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[2], s));
	uint32_t const lds = ctx->realign_lds_bytes ? ctx->realign_lds_bytes : realign_lds_bytes();
	if (int const e = launch_realign(ctx->d_entries, ctx->n_entries, b->in, b->work, ctx->d_params, ctx->realign_waves, big_wave_count(ctx), lds, (ctx->packed_dp && 0 == b->params.tiebreak_mode) ? 1u : 0u, s)) return set_error(PVC3_E_HIP, "k_realign launch: %s", hipGetErrorString((hipError_t) e));
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[3], s));
	if (int const e = launch_pack(b->n, b->work, b->packed, b->scan_tmp, s)) return set_error(PVC3_E_HIP, "pack launch: %s", hipGetErrorString((hipError_t) e));
	if (ctx->timing) HIP_TRY(hipEventRecord(ctx->ev[4], s));
	return PVC3_OK;
}

}	// anonymous namespace


extern "C" {

int pvc3_version(void) { return PVC3_VERSION; }

char const *pvc3_strerror(int code)
{
	switch (code)
	{
		case PVC3_OK: return "ok";
		case PVC3_E_INVALID: return "invalid argument";
		case PVC3_E_NOMEM: return "out of memory";
		case PVC3_E_HIP: return "HIP runtime error";
		case PVC3_E_NO_DEVICE: return "no usable GPU";
		case PVC3_E_CAPACITY: return "output buffer too small";
		case PVC3_E_STATE: return "invalid call sequence";
		default: return "unknown error";
	}
}

char const *pvc3_last_error(void) { return g_error; }

int pvc3_device_count(int *count)
{
	if (!count) return set_error(PVC3_E_INVALID, "pvc3_device_count: null argument");
	int n = 0;
	hipError_t const e = hipGetDeviceCount(&n);
	if (hipSuccess != e) { (void) hipGetLastError(); n = 0; }
	*count = n;
	return PVC3_OK;
}

// ---- index -------------------------------------------------------------------------------------------------------
int pvc3_index_create(pvc3_index **idx)
{
	if (!idx) return set_error(PVC3_E_INVALID, "pvc3_index_create: null argument");
	*idx = new (std::nothrow) pvc3_index;
	return *idx ? PVC3_OK : set_error(PVC3_E_NOMEM, "pvc3_index_create: out of memory");
}

void pvc3_index_destroy(pvc3_index *idx) { delete idx; }

int pvc3_index_add_aligned(pvc3_index *idx, char const *aligned, uint64_t n_cols, uint32_t *entry)
{
	if (!idx || (!aligned && n_cols) || !entry) return set_error(PVC3_E_INVALID, "pvc3_index_add_aligned: null argument");
	try { return index_add_aligned(idx, aligned, n_cols, entry); }
	catch (std::bad_alloc const &) { return set_error(PVC3_E_NOMEM, "pvc3_index_add_aligned: out of memory"); }
}

int pvc3_index_add_bits(pvc3_index *idx, uint64_t const *gap_words, uint64_t n_cols, uint32_t *entry)
{
	if (!idx || (!gap_words && n_cols) || !entry) return set_error(PVC3_E_INVALID, "pvc3_index_add_bits: null argument");
	try { return index_add_bits(idx, gap_words, n_cols, entry); }
	catch (std::bad_alloc const &) { return set_error(PVC3_E_NOMEM, "pvc3_index_add_bits: out of memory"); }
}

int pvc3_index_set_reference(pvc3_index *idx, uint32_t entry, char const *text, uint64_t len)
{
	if (!idx || (!text && len)) return set_error(PVC3_E_INVALID, "pvc3_index_set_reference: null argument");
	try { return index_set_reference(idx, entry, text, len); }
	catch (std::bad_alloc const &) { return set_error(PVC3_E_NOMEM, "pvc3_index_set_reference: out of memory"); }
}

int pvc3_index_set_destination(pvc3_index *idx, uint32_t entry, uint32_t dst_entry)
{
	if (!idx || entry >= idx->entries.size() || dst_entry >= idx->entries.size()) return set_error(PVC3_E_INVALID, "pvc3_index_set_destination: no such entry");
	if (idx->entries[entry].n_cols != idx->entries[dst_entry].n_cols)
		return set_error(PVC3_E_INVALID, "pvc3_index_set_destination: entries %u and %u have different numbers of columns", entry, dst_entry);
	idx->entries[entry].destination = dst_entry;
	return PVC3_OK;
}

int pvc3_index_entry_count(pvc3_index const *idx, uint32_t *count)
{
	if (!idx || !count) return set_error(PVC3_E_INVALID, "pvc3_index_entry_count: null argument");
	*count = (uint32_t) idx->entries.size();
	return PVC3_OK;
}

int pvc3_index_entry_info(pvc3_index const *idx, uint32_t entry, uint64_t *n_cols, uint64_t *n_non_gap)
{
	if (!idx || entry >= idx->entries.size()) return set_error(PVC3_E_INVALID, "pvc3_index_entry_info: no entry %u", entry);
	if (n_cols) *n_cols = idx->entries[entry].n_cols;
	if (n_non_gap) *n_non_gap = idx->entries[entry].n_zeros;
	return PVC3_OK;
}

int pvc3_index_entry_layout(pvc3_index const *idx, uint32_t entry, uint64_t const **blocks, uint64_t *n_blocks, uint32_t const **sel, uint64_t *n_sel)
{
	if (!idx || entry >= idx->entries.size()) return set_error(PVC3_E_INVALID, "pvc3_index_entry_layout: no entry %u", entry);
	host_entry const &e = idx->entries[entry];
	if (blocks) *blocks = e.blocks.data();
	if (n_blocks) *n_blocks = e.n_blocks + PAD_BLOCKS;
	if (sel) *sel = e.sel.data();
	if (n_sel) *n_sel = e.sel.size();
	return PVC3_OK;
}

// ---- context -------------------------------------------------------------------------------------------------------
int pvc3_context_create(int device, pvc3_context **out)
{
	if (!out) return set_error(PVC3_E_INVALID, "pvc3_context_create: null argument");
	*out = nullptr;
	int n = 0;
	if (hipSuccess != hipGetDeviceCount(&n) || n <= 0)
	{
		(void) hipGetLastError();
		return set_error(PVC3_E_NO_DEVICE, "pvc3_context_create: no HIP device is visible; this library has no CPU path");
	}
	if (device < 0 || device >= n)
		return set_error(PVC3_E_NO_DEVICE, "pvc3_context_create: device %d out of range (%d visible)", device, n);
	HIP_TRY(hipSetDevice(device));
	pvc3_context *ctx = new (std::nothrow) pvc3_context;
	if (!ctx) return set_error(PVC3_E_NOMEM, "pvc3_context_create: out of memory");
	ctx->device = device;
	hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
	if (hipSuccess != e) { delete ctx; return set_error(PVC3_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
	ctx->stream = ctx->own_stream;
	for (auto &ev : ctx->ev)
	{
		e = hipEventCreate(&ev);
		if (hipSuccess != e) { delete ctx; return set_error(PVC3_E_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
	}
	*out = ctx;
	return PVC3_OK;
}

void pvc3_context_destroy(pvc3_context *ctx)
{
	if (!ctx) return;
	(void) hipSetDevice(ctx->device);
	(void) hipStreamSynchronize(ctx->stream);
	if (ctx->kept_batch) { pvc3_dbatch_destroy(ctx->kept_batch); ctx->kept_batch = nullptr; }
	free_index_replica(ctx);
	if (ctx->d_params) (void) hipFree(ctx->d_params);
	if (ctx->scratch) (void) hipFree(ctx->scratch);
	if (ctx->trace) (void) hipFree(ctx->trace - 16);
	for (auto &ev : ctx->ev) if (ev) (void) hipEventDestroy(ev);
	if (ctx->own_stream) (void) hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

int pvc3_context_set_stream(pvc3_context *ctx, void *hip_stream)
{
	if (!ctx) return set_error(PVC3_E_INVALID, "pvc3_context_set_stream: null context");
	PVC3_TRY(check_device(ctx));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	ctx->stream = hip_stream ? (hipStream_t) hip_stream : ctx->own_stream;
	return PVC3_OK;
}

int pvc3_context_set_timing(pvc3_context *ctx, int enabled)
{
	if (!ctx) return set_error(PVC3_E_INVALID, "pvc3_context_set_timing: null context");
	ctx->timing = (0 != enabled);
	return PVC3_OK;
}

int pvc3_context_set_option(pvc3_context *ctx, char const *key, int64_t value)
{
	if (!ctx || !key) return set_error(PVC3_E_INVALID, "pvc3_context_set_option: null argument");
	std::string const k(key);
	if (value < 0) return set_error(PVC3_E_INVALID, "pvc3_context_set_option: negative value for %s", key);
	if ("realign_waves" == k) { if (value < 1 || value > 65536) return set_error(PVC3_E_INVALID, "realign_waves out of range"); ctx->realign_waves = (uint32_t) value; }
	else if ("realign_lds_bytes" == k) ctx->realign_lds_bytes = (uint32_t) value;
	else if ("scratch_per_wave" == k) ctx->scratch_per_wave = (uint64_t) value;
	else if ("scratch_budget" == k) ctx->scratch_budget = (uint64_t) value;
	else if ("trace_initial" == k) ctx->trace_initial = (uint64_t) value;
	else if ("big_waves" == k) { if (value < 1 || value > 65536) return set_error(PVC3_E_INVALID, "big_waves out of range"); ctx->big_waves = (uint32_t) value; }
	else if ("arena_initial" == k) ctx->arena_initial = (uint64_t) value;
	else if ("runs_initial" == k) ctx->runs_initial = (uint64_t) value;
	else if ("fast_path" == k) ctx->fast_path = (0 != value);
	else if ("mid_path" == k) ctx->mid_path = (0 != value);
	else if ("packed_dp" == k) ctx->packed_dp = (0 != value);
	else if ("paired_blocks" == k) ctx->paired_blocks = (0 != value);		// takes effect at the next pvc3_context_upload_index
	else if ("text_window" == k) ctx->text_window = (0 != value);			// likewise
	else if ("position_windows" == k) ctx->position_windows = (0 != value);	// likewise
	else if ("window_read_length" == k) ctx->window_read_length = (uint32_t) ((value > 4096) ? 4096 : value);	// likewise
	else if ("window_reserve" == k) ctx->window_reserve = (uint64_t) value;	// likewise
	else if ("staging_threads" == k) ctx->stager.set_threads((uint32_t) ((value > 64) ? 64 : value));	// before the first ordinary array is copied
	else return set_error(PVC3_E_INVALID, "pvc3_context_set_option: unknown option %s", key);
	return PVC3_OK;
}

int pvc3_context_synchronize(pvc3_context *ctx)
{
	if (!ctx) return set_error(PVC3_E_INVALID, "pvc3_context_synchronize: null context");
	PVC3_TRY(check_device(ctx));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PVC3_OK;
}

int pvc3_context_upload_index(pvc3_context *ctx, pvc3_index const *idx)
{
	if (!ctx || !idx) return set_error(PVC3_E_INVALID, "pvc3_context_upload_index: null argument");
	bool const trace_ = nullptr != std::getenv("PVC3_UPLOAD_TRACE");
	auto const t0_ = std::chrono::steady_clock::now();
	PVC3_TRY(check_device(ctx));
	if (trace_) std::fprintf(stderr, "upload_index: device set at %.4f s\n", std::chrono::duration <double>(std::chrono::steady_clock::now() - t0_).count());
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if (trace_) std::fprintf(stderr, "upload_index: stream idle at %.4f s\n", std::chrono::duration <double>(std::chrono::steady_clock::now() - t0_).count());
	auto mark_ = [&](char const *what) { if (trace_) { (void) hipStreamSynchronize(ctx->stream); std::fprintf(stderr, "upload_index: %s at %.4f s\n", what, std::chrono::duration <double>(std::chrono::steady_clock::now() - t0_).count()); } };
	free_index_replica(ctx);
	std::vector <dev_entry> table(idx->entries.size());
	// Position windows: two bytes per source position of every entry that has a destination.  Built when they leave
	// `window_reserve` bytes of the device free (288 GB of HBM: a human-scale index with twenty founders fits); otherwise the
	// index is served by the select samples and the paired blocks alone (k_project_fast_probes).
	uint32_t const pwin_shift = (ctx->window_read_length <= 158) ? 6 : 7;
	auto partner_of = [&](size_t i) -> size_t {
		host_entry const &e = idx->entries[i];
		if (NO_PARTNER != e.destination && !idx->entries[e.destination].text.empty()) return e.destination;		// named by the caller (pvc3_index_set_destination)
		if (!e.text.empty()) return i;			// an entry with a text is its own destination (reads aligned to the reference row)
		size_t best = idx->entries.size();
		for (size_t j = 0; j < idx->entries.size(); ++j)
		{
			host_entry const &o = idx->entries[j];
			if (j == i || o.text.empty() || o.n_cols != e.n_cols) continue;
			size_t const dist = (j > i) ? (j - i) : (i - j);
			if (best == idx->entries.size() || dist < ((best > i) ? (best - i) : (i - best))) best = j;
		}
		return best;
	};
	bool windows = 0 != ctx->position_windows;
	if (windows)
	{
		uint64_t need = 0, plain = 0;
		for (size_t i = 0; i < idx->entries.size(); ++i)
		{
			host_entry const &e = idx->entries[i];
			plain += 3 * e.blocks.size() * sizeof(uint64_t) + 2 * e.text.size();
			if (e.n_zeros && partner_of(i) < idx->entries.size())
				need += (((e.n_zeros - 1) >> pwin_shift) + 1) * (UINT64_C(2) << pwin_shift);
		}
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		windows = need && (need + plain + ctx->window_reserve <= free_b);
	}
	mark_("sizes");
	ctx->have_windows = false;
	ctx->wide_windows = pwin_shift > 6;
	// rows with more than one gap column in sixteen (k_project_mid's own measure): hardly a window of theirs is clean (k_project_fast <.., wary>)
	ctx->gap_dense = false;
	for (auto const &e : idx->entries) ctx->gap_dense = ctx->gap_dense || (e.n_zeros && 16 * (e.n_cols - e.n_zeros) > e.n_cols);
	for (size_t i = 0; i < idx->entries.size(); ++i)
	{
		host_entry const &e = idx->entries[i];
		dev_entry &d = table[i];
		d.pwin = nullptr; d.pwin_partner = NO_PARTNER; d.pwin_shift = pwin_shift;
		PVC3_TRY(dev_upload(&d.blocks, e.blocks.data(), e.blocks.size(), 0, ctx->stream, &ctx->index_allocs, &ctx->index_bytes));
		PVC3_TRY(dev_upload(&d.sel, e.sel.data(), e.sel.size(), 64, ctx->stream, &ctx->index_allocs, &ctx->index_bytes));
		if (0 == i) mark_("first entry's blocks and samples");
		d.text = nullptr;
		d.text4 = nullptr;
		d.text4w = nullptr;
		if (!e.text.empty())
		{
			// Front and back padding: the kernels read whole 16 / 32-byte pieces around the bases they need.
			uint8_t *p{};
			PVC3_TRY(dev_alloc(&p, TEXT_FRONT_PAD + e.text.size() + TEXT_PAD, &ctx->index_allocs, &ctx->index_bytes));
			HIP_TRY(hipMemsetAsync(p, 0, TEXT_FRONT_PAD, ctx->stream));
			HIP_TRY(hipMemcpyAsync(p + TEXT_FRONT_PAD, e.text.data(), e.text.size(), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(hipMemsetAsync(p + TEXT_FRONT_PAD + e.text.size(), 0, TEXT_PAD, ctx->stream));
			d.text = p + TEXT_FRONT_PAD;
			// 4-bit copy for the fast path, made on the device from the bytes just uploaded.
			uint8_t *p4{};
			uint64_t const bytes4 = (TEXT4_FRONT + e.text.size() + 1) / 2 + TEXT_PAD;
			PVC3_TRY(dev_alloc(&p4, bytes4, &ctx->index_allocs, &ctx->index_bytes));
			HIP_TRY(hipMemsetAsync(p4, 0, bytes4, ctx->stream));
			if (trace_) mark_("a text uploaded");
			HIP_TRY((hipError_t) launch_pack_text4(d.text, e.text.size(), p4, ctx->stream));
			d.text4 = p4;
			if (trace_) mark_("a text packed");
			if (ctx->text_window && !windows)
			{
				// Every line the kernel can address: line k for a run that starts at nibble n has k = n / STEP <= nibbles / STEP.
				uint64_t const n_lines = (TEXT4_FRONT + e.text.size()) / TEXT_WINDOW_STEP + 1;
				static_assert(TEXT_PAD >= 128 + 16, "the last lines are cut from the packed text's padding");
				uint8_t *pw{};
				PVC3_TRY(dev_alloc(&pw, 128 * n_lines + 16, &ctx->index_allocs, &ctx->index_bytes));
				HIP_TRY((hipError_t) launch_window_text4(p4, n_lines, pw, ctx->stream));
				d.text4w = pw;
			}
		}
		d.n_cols = (uint32_t) e.n_cols;
		d.n_zeros = (uint32_t) e.n_zeros;
		d.text_len = (uint32_t) e.text.size();
		d.n_blocks = (uint32_t) e.n_blocks;
		d.paired = nullptr;
		d.partner = NO_PARTNER;
		d.pad = 0;
		// Partner: the nearest other entry of the same MSA (same number of columns) that carries a text, i.e. the destination of
		// the projections from this entry.  Any choice is correct: the copy is used only for records whose destination it is.
		if (ctx->paired_blocks)
		{
			size_t best = idx->entries.size();
			bool const named = NO_PARTNER != e.destination && e.destination != i && !idx->entries[e.destination].text.empty();		// pvc3_index_set_destination
			if (named) best = e.destination;
			for (size_t j = 0; !named && j < idx->entries.size(); ++j)
			{
				host_entry const &o = idx->entries[j];
				if (j == i || o.text.empty() || o.n_cols != e.n_cols) continue;
				size_t const dist = (j > i) ? (j - i) : (i - j);
				if (best == idx->entries.size() || dist < ((best > i) ? (best - i) : (i - best))) best = j;
			}
			if (best < idx->entries.size())
			{
				d.partner = (uint32_t) best;
				if (PAIRED_STEP != BLOCK_COLS)
				{
					// (overlapping paired blocks, an experiment: built on the host)
					host_entry const &o = idx->entries[best];
					std::vector <uint64_t> both;
					build_paired_blocks(e, o, both);
					PVC3_TRY(dev_upload(&d.paired, both.data(), both.size(), 0, ctx->stream, &ctx->index_allocs, &ctx->index_bytes));
					HIP_TRY(hipStreamSynchronize(ctx->stream));		// `both` is pageable and leaves scope
				}
			}
		}
	}
	mark_("entries");
	if (ctx->paired_blocks && PAIRED_STEP == BLOCK_COLS)
	{
		// The paired copies, interleaved on the device from the blocks uploaded above.
		for (size_t i = 0; i < idx->entries.size(); ++i)
		{
			if (NO_PARTNER == table[i].partner) continue;
			host_entry const &e = idx->entries[i], &o = idx->entries[table[i].partner];
			uint64_t const n_paired = e.n_cols / PAIRED_STEP + 1 + PAD_BLOCKS, n_have = e.n_blocks + PAD_BLOCKS;
			uint64_t *both{};
			PVC3_TRY(dev_alloc(&both, n_paired * 2 * BLOCK_WORDS, &ctx->index_allocs, &ctx->index_bytes));
			if (int const rc = launch_pair_blocks(table[i].blocks, table[table[i].partner].blocks, n_have, n_paired, e.n_zeros, o.n_zeros, both, ctx->stream))
				return set_error(PVC3_E_HIP, "k_pair_blocks launch: %s", hipGetErrorString((hipError_t) rc));
			table[i].paired = both;
		}
	}
	mark_("paired");
	if (windows)
	{
		for (size_t i = 0; i < idx->entries.size(); ++i)
		{
			host_entry const &e = idx->entries[i];
			size_t const partner = partner_of(i);
			if (0 == e.n_zeros || partner >= idx->entries.size()) continue;
			uint64_t const n_windows = ((e.n_zeros - 1) >> pwin_shift) + 1;
			uint8_t *pw{};
			PVC3_TRY(dev_alloc(&pw, n_windows * (UINT64_C(2) << pwin_shift) + 16, &ctx->index_allocs, &ctx->index_bytes));
			if (int const rc = launch_build_pwin(table[i], table[partner], pwin_shift, n_windows, pw, ctx->stream))
				return set_error(PVC3_E_HIP, "k_pwin launch: %s", hipGetErrorString((hipError_t) rc));
			table[i].pwin = pw;
			table[i].pwin_partner = (uint32_t) partner;
			ctx->have_windows = true;
		}
	}
	dev_entry const *d_table{};
	PVC3_TRY(dev_upload(&d_table, table.data(), table.size(), 0, ctx->stream, &ctx->index_allocs, &ctx->index_bytes));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	ctx->d_entries = const_cast <dev_entry *>(d_table);
	ctx->n_entries = (uint32_t) table.size();
	mark_("windows");
	return PVC3_OK;
}

int pvc3_host_alloc(uint64_t bytes, void **ptr)
{
	if (!ptr) return set_error(PVC3_E_INVALID, "pvc3_host_alloc: null argument");
	*ptr = nullptr;
	HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
	return PVC3_OK;
}

void pvc3_host_free(void *ptr)
{
	if (ptr) (void) hipHostFree(ptr);
}

int pvc3_context_hbm_bytes(pvc3_context const *ctx, uint64_t *index_bytes, uint64_t *workspace_bytes)
{
	if (!ctx) return set_error(PVC3_E_INVALID, "pvc3_context_hbm_bytes: null context");
	if (index_bytes) *index_bytes = ctx->index_bytes;
	if (workspace_bytes) *workspace_bytes = ctx->workspace_bytes;
	return PVC3_OK;
}

// ---- device batches ------------------------------------------------------------------------------------------------
void pvc3_dbatch_destroy(pvc3_dbatch *b)
{
	if (!b) return;
	(void) hipSetDevice(b->ctx->device);
	(void) hipStreamSynchronize(b->ctx->stream);
	for (void *p : b->allocs) (void) hipFree(p);
	if (b->work.arena) (void) hipFree(b->work.arena);
	if (b->packed.out_cigar) (void) hipFree(b->packed.out_cigar);
	if (b->work.runs_arena) (void) hipFree(b->work.runs_arena);
	if (b->packed.runs) (void) hipFree(b->packed.runs);
	if (b->work.run_items) (void) hipFree(b->work.run_items);
	if (b->work.dp_list) (void) hipFree(b->work.dp_list);
	if (b->work.qual_compact) (void) hipFree(b->work.qual_compact);
	delete b;
}

}	// extern "C"

namespace {

// Allocates every device array of a batch for up to `caps` (inputs, per-record work arrays, arenas): a batch object is
// filled again and again by dbatch_load() without touching the allocator (hipMalloc / hipFree cost milliseconds each).
int dbatch_alloc(pvc3_context *ctx, batch_caps const &caps, pvc3_dbatch **out)
{
	*out = nullptr;
	if (caps.n >= (UINT64_C(1) << 32))
		return set_error(PVC3_E_INVALID, "at most 2^32 - 1 records per batch");
	pvc3_dbatch *b = new (std::nothrow) pvc3_dbatch;
	if (!b) return set_error(PVC3_E_NOMEM, "pvc3_dbatch: out of memory");
	b->ctx = ctx; b->caps = caps;
	struct guard { pvc3_dbatch *p; ~guard() { if (p) pvc3_dbatch_destroy(p); } } g{b};
	uint64_t const n = caps.n;
	auto in_alloc = [&](auto **dst, uint64_t count, uint64_t pad) -> int {
		uint8_t *p{};
		typedef std::remove_const_t <std::remove_pointer_t <std::remove_pointer_t <decltype(dst)>>> t_value;
		PVC3_TRY(dev_alloc(&p, count * sizeof(t_value) + pad, &b->allocs, &b->bytes));
		*dst = (t_value *) p;
		return PVC3_OK;
	};
	dev_batch_in &d = b->in;
	PVC3_TRY(in_alloc(&d.src_pos, n, 0));
	PVC3_TRY(in_alloc(&d.src_entry, n, 0));
	PVC3_TRY(in_alloc(&d.dst_entry, n, 0));
	PVC3_TRY(in_alloc(&d.flags, n, 0));
	PVC3_TRY(in_alloc(&d.cigar_off, n + 1, 0));
	PVC3_TRY(in_alloc(&d.cigar, caps.cigar, 256));		// padded: whole 16-byte reads of the first operations
	PVC3_TRY(in_alloc(&d.seq_off, n + 1, 0));
	PVC3_TRY(in_alloc(&d.l_seq, n, 0));
	PVC3_TRY(in_alloc(&d.seq4, caps.seq, 256));			// padded: whole-dword reads
	if (caps.qual)
	{
		PVC3_TRY(in_alloc(&b->up_qual_off, n + 1, 0));
		PVC3_TRY(in_alloc(&b->up_qual, caps.qual, 64));		// padded: the alignment kernels read 8-byte pieces
	}
	if (caps.mates)
	{
		PVC3_TRY(in_alloc(&b->up_mate_src_entry, n, 0));
		PVC3_TRY(in_alloc(&b->up_mate_dst_entry, n, 0));
		PVC3_TRY(in_alloc(&b->up_mate_pos, n, 0));
	}

	dev_batch_work &w = b->work;
	PVC3_TRY(dev_alloc(&w.dst_pos, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.mate_dst_pos, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.rec_flags, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.cig_loc, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.cig_n, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.run_loc, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.run_n, n, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.counters, 1, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.slow_list, (n + 63) / 64 * 64 + 64, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.slow_items, (n + 63) / 64 * 64 + 64, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.slow_count, project_wave_count(n), &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.run_list, (n + 63) / 64 * 64 + 64, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.run_count, project_wave_count(n), &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.run_prefix, (uint64_t) project_wave_count(n) + 1, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&w.wave_stats, 3 * (uint64_t) project_wave_count(n) * WAVE_STATS, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&b->packed.out_cigar_off, n + 1, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&b->packed.run_off, n + 1, &b->allocs, &b->bytes));
	b->packed.status = (int32_t *) w.rec_flags;		// every record's flags end as its PVC3_ST_* value: no copy
	PVC3_TRY(dev_alloc(&b->scan_tmp, pack_scan_tmp_words(n), &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&b->d_checksum, 1, &b->allocs, &b->bytes));
	PVC3_TRY(dev_alloc(&b->d_invalid, 1, &b->allocs, &b->bytes));
	w.first_bad = b->d_invalid;
	if (caps.qual_compact)
	{
		PVC3_TRY(dev_alloc(&w.qual_off_dev, n, &b->allocs, &b->bytes));
		PVC3_TRY(dev_alloc(&w.qual_compact, caps.qual_compact + 64, nullptr, &b->bytes));		// (+ 64: the alignment kernels read 8-byte pieces)
		w.qual_compact_cap = caps.qual_compact;
	}

	// Arena sizing: twice the input operations + 8 per record + the chunk slack of every wave that may allocate.
	uint64_t const realign_waves = (ctx->realign_waves < (n + 63) / 64) ? ctx->realign_waves : (n + 63) / 64;
	uint64_t const chunk_slack = ((uint64_t) project_wave_count(n) + realign_waves) * 2048 * 2;
	uint64_t const arena_cap = ctx->arena_initial ? ctx->arena_initial : (2 * caps.cigar + 8 * n + chunk_slack + 4096);
	uint64_t const runs_cap = ctx->runs_initial ? ctx->runs_initial : (n / 16 + realign_waves * 128 * 2 + 1024);
	PVC3_TRY(alloc_arenas(b, arena_cap, runs_cap));
	g.p = nullptr;
	*out = b;
	return PVC3_OK;
}

bool device_visible_host_pointer(void const *p)
{
	if (!p) return false;
	hipPointerAttribute_t attr{};
	if (hipSuccess != hipPointerGetAttributes(&attr, p)) { (void) hipGetLastError(); return false; }
	return hipMemoryTypeHost == attr.type;
}

// Fills a batch object: asynchronous copies on `s` (page-locked source arrays must stay valid until the copies have run) and
// the check of the variable-length layout on the device.  With lazy_qual QUAL is not copied: the kernels fetch it from the
// caller's page-locked arrays for the records that are realigned (dev_batch_work::host_qual).
int dbatch_load(pvc3_context *ctx, pvc3_dbatch *b, pvc3_batch_in const *in, hipStream_t s, bool lazy_qual)
{
	uint64_t const n = in->n;
	batch_caps const &caps = b->caps;
	if (n && (!in->src_pos || !in->src_entry || !in->dst_entry || !in->flags || !in->cigar_off || !in->seq_off || !in->l_seq))
		return set_error(PVC3_E_INVALID, "pvc3 batch: a required input array is null");
	static uint64_t const zero_off[1] = {0};
	uint64_t const n_cigar = n ? in->cigar_off[n] : 0;
	uint64_t const n_seq = n ? in->seq_off[n] : 0;
	bool const has_qual = n && in->qual_off && in->qual;
	uint64_t const n_qual = has_qual ? in->qual_off[n] : 0;
	bool const has_mates = n && in->mate_src_entry && in->mate_dst_entry && in->mate_pos;
	if (n_cigar && !in->cigar) return set_error(PVC3_E_INVALID, "pvc3 batch: cigar is null");
	if (n_seq && !in->seq4) return set_error(PVC3_E_INVALID, "pvc3 batch: seq4 is null");
	lazy_qual = lazy_qual && has_qual && caps.qual_compact;
	if (n > caps.n || n_cigar > caps.cigar || n_seq > caps.seq || (has_qual && !lazy_qual && n_qual > caps.qual) || (has_mates && !caps.mates))
		return set_error(PVC3_E_CAPACITY, "pvc3 batch: %llu records, %llu CIGAR operations, %llu SEQ bytes, %llu QUAL bytes exceed the capacities %llu, %llu, %llu, %llu",
			(unsigned long long) n, (unsigned long long) n_cigar, (unsigned long long) n_seq, (unsigned long long) ((has_qual && !lazy_qual) ? n_qual : 0),
			(unsigned long long) caps.n, (unsigned long long) caps.cigar, (unsigned long long) caps.seq, (unsigned long long) caps.qual);

	auto put = [&](auto const *dst, auto const *src, uint64_t count, uint64_t pad) -> int {
		uint8_t *p = (uint8_t *) const_cast <void *>((void const *) dst);
		if (count) HIP_TRY(ctx->stager.h2d(p, src, count * sizeof(*src), s));		// page-locked arrays directly, ordinary ones through the staging ring
		if (pad) HIP_TRY(hipMemsetAsync(p + count * sizeof(*src), 0, pad, s));
		return PVC3_OK;
	};
	dev_batch_in &d = b->in;
	d.n = n; b->n = n;
	PVC3_TRY(put(d.src_pos, in->src_pos, n, 0));
	PVC3_TRY(put(d.src_entry, in->src_entry, n, 0));
	PVC3_TRY(put(d.dst_entry, in->dst_entry, n, 0));
	PVC3_TRY(put(d.flags, in->flags, n, 0));
	PVC3_TRY(put(d.cigar_off, n ? in->cigar_off : zero_off, n + 1, 0));
	PVC3_TRY(put(d.cigar, in->cigar, n_cigar, 256));
	PVC3_TRY(put(d.seq_off, n ? in->seq_off : zero_off, n + 1, 0));
	PVC3_TRY(put(d.l_seq, in->l_seq, n, 0));
	PVC3_TRY(put(d.seq4, in->seq4, n_seq, 256));
	dev_batch_work &w = b->work;
	d.qual = nullptr; d.qual_off = nullptr; w.host_qual = nullptr; w.host_qual_off = nullptr;
	if (lazy_qual)
	{
		w.host_qual = in->qual; w.host_qual_off = in->qual_off;		// page-locked: the same addresses are valid on the device
		d.qual = w.qual_compact; d.qual_off = w.qual_off_dev;
	}
	else if (has_qual)
	{
		PVC3_TRY(put(b->up_qual_off, in->qual_off, n + 1, 0));
		PVC3_TRY(put(b->up_qual, in->qual, n_qual, 64));
		d.qual = b->up_qual; d.qual_off = b->up_qual_off;
	}
	d.mate_src_entry = nullptr; d.mate_dst_entry = nullptr; d.mate_pos = nullptr;
	if (has_mates)
	{
		PVC3_TRY(put(b->up_mate_src_entry, in->mate_src_entry, n, 0));
		PVC3_TRY(put(b->up_mate_dst_entry, in->mate_dst_entry, n, 0));
		PVC3_TRY(put(b->up_mate_pos, in->mate_pos, n, 0));
		d.mate_src_entry = b->up_mate_src_entry; d.mate_dst_entry = b->up_mate_dst_entry; d.mate_pos = b->up_mate_pos;
	}
	// Offsets must not decrease and every record must own the SEQ / QUAL bytes its l_seq promises: the kernels read
	// seq4 + seq_off and qual + qual_off without further checks.
	HIP_TRY(hipMemsetAsync(b->d_invalid, 0xff, sizeof(uint32_t), s));
	// (QUAL that stays in the caller's page-locked arrays is read by k_gather_qual through those offsets: they are checked
	// just the same, in place — eight bytes per record over PCIe.)
	if (int const e = launch_validate(d, n_cigar, n_seq, n_qual, lazy_qual ? w.host_qual_off : d.qual_off, b->d_invalid, s)) return set_error(PVC3_E_HIP, "k_validate_batch launch: %s", hipGetErrorString((hipError_t) e));
	b->projected = false; b->finished = false;
	return PVC3_OK;
}

// `s`: the stream the batch was loaded on; null when the caller has already waited for it.
int dbatch_check_valid(pvc3_dbatch *b, hipStream_t s)
{
	uint32_t bad = 0;
	if (s) HIP_TRY(hipStreamSynchronize(s));
	HIP_TRY(hipMemcpy(&bad, b->d_invalid, sizeof(bad), hipMemcpyDeviceToHost));
	if (0xFFFFFFFFu != bad)
		return set_error(PVC3_E_INVALID, "pvc3 batch: malformed variable-length layout (offsets decrease, run past the arrays, or a record's SEQ / QUAL slice is shorter than l_seq); first bad record %u", bad);
	return PVC3_OK;
}

}	// anonymous namespace

extern "C" {

int pvc3_dbatch_create(pvc3_context *ctx, pvc3_batch_in const *in, pvc3_dbatch **out)
{
	if (!ctx || !in || !out) return set_error(PVC3_E_INVALID, "pvc3_dbatch_create: null argument");
	*out = nullptr;
	PVC3_TRY(check_device(ctx));
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_dbatch_create: no index has been uploaded to this context");
	uint64_t const n = in->n;
	if (n && (!in->cigar_off || !in->seq_off))
		return set_error(PVC3_E_INVALID, "pvc3_dbatch_create: a required input array is null");
	batch_caps caps;
	caps.n = n;
	caps.cigar = n ? in->cigar_off[n] : 0;
	caps.seq = n ? in->seq_off[n] : 0;
	caps.qual = (n && in->qual_off && in->qual) ? in->qual_off[n] : 0;
	caps.mates = n && in->mate_src_entry && in->mate_dst_entry && in->mate_pos;
	pvc3_dbatch *b{};
	PVC3_TRY(dbatch_alloc(ctx, caps, &b));
	int rc = dbatch_load(ctx, b, in, ctx->stream, false);
	if (!rc) rc = dbatch_check_valid(b, ctx->stream);
	if (rc) { pvc3_dbatch_destroy(b); return rc; }
	*out = b;
	return PVC3_OK;
}

int pvc3_dbatch_project(pvc3_context *ctx, pvc3_dbatch *b, pvc3_params const *params)
{
	if (!ctx || !b || !params) return set_error(PVC3_E_INVALID, "pvc3_dbatch_project: null argument");
	if (b->ctx != ctx) return set_error(PVC3_E_INVALID, "pvc3_dbatch_project: the batch belongs to another context");
	PVC3_TRY(check_device(ctx));
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_dbatch_project: no index has been uploaded to this context");
	PVC3_TRY(upload_params(ctx, *params));
	PVC3_TRY(ensure_scratch(ctx));
	b->params = *params;
	b->attempts = 1;
	b->finished = false;
	PVC3_TRY(enqueue_projection(ctx, b));
	b->projected = true;
	return PVC3_OK;
}

}	// extern "C"

namespace {

// Waits for the projection of `b` on the context's stream; grows whatever overflowed and repeats the batch on the GPU until
// it fits.  Leaves the counters in b->counters and the packed totals in b->totals.
int dbatch_complete(pvc3_context *ctx, pvc3_dbatch *b, bool already_waited = false)
{
	for (;; already_waited = false)
	{
		if (!already_waited) HIP_TRY(hipStreamSynchronize(ctx->stream));
		HIP_TRY(hipMemcpy(&b->counters, b->work.counters, sizeof(dev_counters), hipMemcpyDeviceToHost));
		dev_counters const &c = b->counters;
		if (!c.arena_overflow && !c.runs_overflow && !c.scratch_overflow && !c.qual_overflow && !c.trace_overflow)
			break;
		if (b->attempts >= 8)
			return set_error(PVC3_E_NOMEM, "pvc3_dbatch_finish: workspaces still too small after %u attempts", b->attempts);
		// Grow whatever overflowed and repeat the whole batch on the GPU.
		if (c.scratch_overflow)
		{
			ctx->scratch_per_wave = ((uint64_t) c.scratch_needed + 4095) / 4096 * 4096;
			if (0xFFFFFFFFu == c.scratch_needed)
				return set_error(PVC3_E_NOMEM, "pvc3_dbatch_finish: an indel run needs more than 4 GiB of alignment scratch");
			// ensure_scratch allocates one slot per wave of the striped kernel, and fewer waves when the slots are large, so the
			// total stays within the budget (a slot larger than the budget gets one wave)
			PVC3_TRY(ensure_scratch(ctx));
		}
		if (c.arena_overflow || c.runs_overflow)
		{
			uint64_t const arena_cap = c.arena_overflow ? (2 * (uint64_t) c.arena_head + 4096) : b->work.arena_cap;
			uint64_t const runs_cap = c.runs_overflow ? (2 * (uint64_t) c.runs_head + 1024) : b->work.runs_cap;
			PVC3_TRY(alloc_arenas(b, arena_cap, runs_cap));
		}
		if (c.trace_overflow)
			PVC3_TRY(ensure_trace(ctx, (uint64_t) c.trace_head + (UINT64_C(4) << 20)));		// the head counts every request, also the refused ones
		if (c.qual_overflow)
		{
			(void) hipFree(b->work.qual_compact);
			b->bytes -= b->work.qual_compact_cap;
			b->work.qual_compact = nullptr;
			uint64_t const cap = 2 * (uint64_t) c.qual_head + 4096;
			PVC3_TRY(dev_alloc(&b->work.qual_compact, cap + 64, nullptr, &b->bytes));
			b->work.qual_compact_cap = cap;
			b->in.qual = b->work.qual_compact;
		}
		++b->attempts;
		PVC3_TRY(enqueue_projection(ctx, b));
	}
	HIP_TRY(hipMemcpy(&b->totals[0], b->packed.out_cigar_off + b->n, sizeof(uint64_t), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(&b->totals[1], b->packed.run_off + b->n, sizeof(uint64_t), hipMemcpyDeviceToHost));
	b->finished = true;
	return PVC3_OK;
}

void fill_stats(pvc3_context *ctx, pvc3_dbatch const *b, pvc3_batch_stats *stats)
{
	dev_counters const &c = b->counters;
	std::memset(stats, 0, sizeof(*stats));
	stats->n_records = b->n;
	stats->n_failed = c.n_failed;
	stats->n_realigned_records = c.n_realigned_records;
	stats->n_fast_path = c.n_fast;
	stats->n_mid_path = c.n_mid;
	stats->n_runs = b->totals[1];
	stats->out_cigar_total = b->totals[0];
	// Algorithmic bytes are accumulated per record by the kernels themselves: what one pass must read and
	// write at minimum.  k_project_fast: descriptors, input CIGAR, select sample + rank word, mate fields (+ the mate's
	// blocks when it is not the neighbouring record), results; for fast-path records also SEQ, gap-bit windows, text and
	// the output CIGAR.  k_project_mid / k_project_general: SEQ, gap bits, text, output CIGAR of the records they walk.
	// Realignment: QUAL of realigned records, 16 bytes per realigned range.  Packing traffic is not algorithmic.
	stats->algorithmic_bytes_project = c.fast_bytes;
	stats->algorithmic_bytes = c.fast_bytes + c.general_bytes + c.qual_bytes + sizeof(pvc3_run) * b->totals[1];
	if (b->timed)
	{
		(void) hipEventElapsedTime(&stats->ms_project, ctx->ev[0], ctx->ev[1]);
		(void) hipEventElapsedTime(&stats->ms_mid, ctx->ev[1], ctx->ev[5]);
		(void) hipEventElapsedTime(&stats->ms_general, ctx->ev[5], ctx->ev[2]);
		(void) hipEventElapsedTime(&stats->ms_realign, ctx->ev[2], ctx->ev[3]);
		(void) hipEventElapsedTime(&stats->ms_pack, ctx->ev[3], ctx->ev[4]);
	}
	stats->attempts = b->attempts;
}

}	// anonymous namespace

extern "C" {

int pvc3_dbatch_finish(pvc3_context *ctx, pvc3_dbatch *b, pvc3_batch_stats *stats)
{
	if (!ctx || !b) return set_error(PVC3_E_INVALID, "pvc3_dbatch_finish: null argument");
	if (!b->projected) return set_error(PVC3_E_STATE, "pvc3_dbatch_finish: pvc3_dbatch_project has not been called");
	PVC3_TRY(check_device(ctx));
	PVC3_TRY(dbatch_complete(ctx, b));
	if (stats) fill_stats(ctx, b, stats);
	return PVC3_OK;
}

int pvc3_dbatch_download(pvc3_context *ctx, pvc3_dbatch *b, pvc3_batch_out *out)
{
	if (!ctx || !b || !out) return set_error(PVC3_E_INVALID, "pvc3_dbatch_download: null argument");
	if (!b->finished) return set_error(PVC3_E_STATE, "pvc3_dbatch_download: pvc3_dbatch_finish has not succeeded");
	PVC3_TRY(check_device(ctx));
	uint64_t const n = b->n;
	if (!out->out_cigar_off || !out->run_off || (n && (!out->dst_pos || !out->mate_dst_pos || !out->status)))
		return set_error(PVC3_E_INVALID, "pvc3_dbatch_download: a required output array is null");
	hipStream_t const s = ctx->stream;
	HIP_TRY(ctx->stager.d2h(out->out_cigar_off, b->packed.out_cigar_off, (n + 1) * sizeof(uint64_t), s));
	HIP_TRY(ctx->stager.d2h(out->run_off, b->packed.run_off, (n + 1) * sizeof(uint64_t), s));
	uint64_t const n_cig = out->out_cigar_off[n], n_runs = out->run_off[n];
	if (n_cig > out->out_cigar_cap || n_runs > out->run_cap)
		return set_error(PVC3_E_CAPACITY, "pvc3_dbatch_download: need room for %llu CIGAR operations and %llu runs", (unsigned long long) n_cig, (unsigned long long) n_runs);
	if (n)
	{
		HIP_TRY(ctx->stager.d2h(out->dst_pos, b->work.dst_pos, n * sizeof(uint32_t), s));
		HIP_TRY(ctx->stager.d2h(out->mate_dst_pos, b->work.mate_dst_pos, n * sizeof(uint32_t), s));
		HIP_TRY(ctx->stager.d2h(out->status, b->packed.status, n * sizeof(int32_t), s));
	}
	if (n_cig) HIP_TRY(ctx->stager.d2h(out->out_cigar, b->packed.out_cigar, n_cig * sizeof(uint32_t), s));
	if (n_runs) HIP_TRY(ctx->stager.d2h(out->runs, b->packed.runs, n_runs * sizeof(pvc3_run), s));
	return PVC3_OK;
}

int pvc3_dbatch_checksum(pvc3_context *ctx, pvc3_dbatch *b, uint64_t first_record, uint64_t *checksum)
{
	if (!ctx || !b || !checksum) return set_error(PVC3_E_INVALID, "pvc3_dbatch_checksum: null argument");
	if (!b->finished) return set_error(PVC3_E_STATE, "pvc3_dbatch_checksum: pvc3_dbatch_finish has not succeeded");
	PVC3_TRY(check_device(ctx));
	if (int const e = launch_checksum(b->n, first_record, b->work, b->packed, b->d_checksum, ctx->stream)) return set_error(PVC3_E_HIP, "k_checksum launch: %s", hipGetErrorString((hipError_t) e));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	unsigned long long v = 0;
	HIP_TRY(hipMemcpy(&v, b->d_checksum, sizeof(v), hipMemcpyDeviceToHost));
	*checksum = v;
	return PVC3_OK;
}

#ifdef PVC3_FAST_STAMPS
// Diagnostic build only (tools/fast_phase_cycles.py): the per-wave phase cycles k_project_fast left at the end of the trace buffer.
PVC3_API int pvc3_debug_fast_stamps(pvc3_dbatch *b, unsigned long long *out, uint32_t n_waves)
{
	HIP_TRY(hipMemcpy(out, (unsigned long long *) (b->work.trace + b->work.trace_cap) - 16 * (uint64_t) n_waves, 16 * (uint64_t) n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return PVC3_OK;
}
#endif

#ifdef PVC3_DP_STAMPS
// Diagnostic build only (tools/dp_phase_cycles.py): the cycle sums the alignment kernels left in the wave statistics area.
PVC3_API int pvc3_debug_wave_stats(pvc3_dbatch *b, unsigned long long *out, uint32_t count)
{
	HIP_TRY(hipMemcpy(out, b->work.wave_stats, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return PVC3_OK;
}
#endif

int pvc3_batch_project(pvc3_context *ctx, pvc3_batch_in const *in, pvc3_batch_out *out, pvc3_params const *params, pvc3_batch_stats *stats)
{
	if (!ctx || !in || !out || !params) return set_error(PVC3_E_INVALID, "pvc3_batch_project: null argument");
	PVC3_TRY(check_device(ctx));
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_batch_project: no index has been uploaded to this context");
	uint64_t const n = in->n;
	if (n && (!in->cigar_off || !in->seq_off))
		return set_error(PVC3_E_INVALID, "pvc3_batch_project: a required input array is null");
	// The device batch of the previous call is used again when this batch fits it (the device arrays of a batch are some
	// thirty-five allocations: 5 - 13 ms of hipMalloc / hipFree per four million records); a larger batch gets a new one
	// with a quarter of headroom.  It lives until the context is destroyed.
	batch_caps need;
	need.n = n;
	need.cigar = n ? in->cigar_off[n] : 0;
	need.seq = n ? in->seq_off[n] : 0;
	need.qual = (n && in->qual_off && in->qual) ? in->qual_off[n] : 0;
	need.mates = true;
	pvc3_dbatch *b = ctx->kept_batch;
	if (!b || need.n > b->caps.n || need.cigar > b->caps.cigar || need.seq > b->caps.seq || need.qual > b->caps.qual)
	{
		if (b) { HIP_TRY(hipStreamSynchronize(ctx->stream)); pvc3_dbatch_destroy(b); ctx->kept_batch = nullptr; b = nullptr; }
		batch_caps caps = need;
		caps.n += caps.n / 4; caps.cigar += caps.cigar / 4 + 64; caps.seq += caps.seq / 4 + 64; caps.qual += caps.qual / 4 + 64;
		if (caps.n >= (UINT64_C(1) << 32)) caps.n = (UINT64_C(1) << 32) - 1;
		PVC3_TRY(dbatch_alloc(ctx, caps, &b));
		ctx->kept_batch = b;
	}
	int rc = dbatch_load(ctx, b, in, ctx->stream, false);
	if (!rc) rc = dbatch_check_valid(b, ctx->stream);
	if (!rc) rc = pvc3_dbatch_project(ctx, b, params);
	if (!rc) rc = pvc3_dbatch_finish(ctx, b, stats);
	if (!rc) rc = pvc3_dbatch_download(ctx, b, out);
	return rc;
}


}	// extern "C"

namespace pvc3 {

int batch_object_alloc(pvc3_context *ctx, uint64_t max_records, uint64_t max_cigar_ops, pvc3_dbatch **out)
{
	batch_caps caps;
	caps.n = max_records; caps.cigar = max_cigar_ops; caps.seq = 0; caps.qual = 0; caps.mates = true;
	return dbatch_alloc(ctx, caps, out);
}

int batch_object_project(pvc3_context *ctx, pvc3_dbatch *b, pvc3_params const *params)
{
	return pvc3_dbatch_project(ctx, b, params);
}

int batch_object_complete(pvc3_context *ctx, pvc3_dbatch *b, pvc3_batch_stats *stats)
{
	return pvc3_dbatch_finish(ctx, b, stats);
}

}

extern "C" {

// ---- pipelined entry point -----------------------------------------------------------------------------------------------
// The reference's record loop reads, projects and writes one record at a time (project_alignments.cc:600-627).  Here batch
// k + 1 is copied to the GPU while batch k is projected and the results of batch k - 1 travel back: three device batches
// that are allocated once, three streams, events in between; QUAL stays in the caller's page-locked arrays and only the
// records with an indel run have theirs fetched (k_gather_qual).
}	// extern "C"

struct pvc3_pipeline
{
	pvc3_context *ctx{};
	hipStream_t s_h2d{}, s_kernels{}, s_d2h{};
	struct slot
	{
		pvc3_dbatch *b{};
		hipEvent_t loaded{}, projected{}, downloaded{};
		pvc3_batch_out out{};
		int state{};					// 0 free, 1 kernels enqueued, 2 download enqueued
		int error{};
		char message[256]{};
	};
	std::vector <slot> slots;
	uint64_t submitted{}, retired{};	// batch numbers; slot = number % depth
};

namespace {

// Kernels of the slot's batch have been enqueued: wait for them (the host learns the sizes of the variable-length results),
// repeat the batch if a workspace overflowed, then enqueue the download into the caller's arrays.
int pipeline_start_download(pvc3_pipeline *p, pvc3_pipeline::slot &sl)
{
	pvc3_context *ctx = p->ctx;
	pvc3_dbatch *b = sl.b;
	// Only this batch's kernels are waited for: later batches may already be queued on the same streams.
	HIP_TRY(hipEventSynchronize(sl.projected));
	hipStream_t const keep = ctx->stream;
	bool const keep_timing = ctx->timing;
	ctx->stream = p->s_kernels;
	ctx->timing = false;
	int rc = dbatch_check_valid(b, nullptr);
	if (!rc) rc = dbatch_complete(ctx, b, true);
	ctx->stream = keep;
	ctx->timing = keep_timing;
	if (rc) return rc;
	uint64_t const n = b->n, n_cig = b->totals[0], n_runs = b->totals[1];
	pvc3_batch_out const &out = sl.out;
	if (n_cig > out.out_cigar_cap || n_runs > out.run_cap)
		return set_error(PVC3_E_CAPACITY, "pvc3_pipeline: need room for %llu CIGAR operations and %llu runs", (unsigned long long) n_cig, (unsigned long long) n_runs);
	hipStream_t const s = p->s_d2h;		// the kernels have been waited for: no event needed
	HIP_TRY(hipMemcpyAsync(out.out_cigar_off, b->packed.out_cigar_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(out.run_off, b->packed.run_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	if (n)
	{
		HIP_TRY(hipMemcpyAsync(out.dst_pos, b->work.dst_pos, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out.mate_dst_pos, b->work.mate_dst_pos, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out.status, b->packed.status, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
	}
	if (n_cig) HIP_TRY(hipMemcpyAsync(out.out_cigar, b->packed.out_cigar, n_cig * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	if (n_runs) HIP_TRY(hipMemcpyAsync(out.runs, b->packed.runs, n_runs * sizeof(pvc3_run), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipEventRecord(sl.downloaded, s));
	sl.state = 2;
	return PVC3_OK;
}

void pipeline_note_error(pvc3_pipeline::slot &sl, int rc)
{
	sl.error = rc;
	std::snprintf(sl.message, sizeof(sl.message), "%s", pvc3_last_error());
	sl.state = 2;		// nothing is in flight for it; pvc3_pipeline_wait reports the error
}

}	// anonymous namespace

extern "C" {

int pvc3_pipeline_create(pvc3_context *ctx, pvc3_pipeline_caps const *caps, pvc3_pipeline **out)
{
	if (!ctx || !caps || !out) return set_error(PVC3_E_INVALID, "pvc3_pipeline_create: null argument");
	*out = nullptr;
	PVC3_TRY(check_device(ctx));
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_pipeline_create: no index has been uploaded to this context");
	uint32_t const depth = caps->depth ? caps->depth : 3;
	if (depth < 2 || depth > 8) return set_error(PVC3_E_INVALID, "pvc3_pipeline_create: depth must be 2 .. 8");
	pvc3_pipeline *p = new (std::nothrow) pvc3_pipeline;
	if (!p) return set_error(PVC3_E_NOMEM, "pvc3_pipeline_create: out of memory");
	p->ctx = ctx;
	struct guard { pvc3_pipeline *p; ~guard() { if (p) pvc3_pipeline_destroy(p); } } g{p};
	HIP_TRY(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&p->s_kernels, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
	p->slots.resize(depth);
	batch_caps bc;
	bc.n = caps->max_records; bc.cigar = caps->max_cigar_ops; bc.seq = caps->max_seq_bytes; bc.qual = caps->max_qual_bytes;
	bc.mates = true;
	// compact QUAL: 16 bytes per record (about one record in ten realigned at 150 bases); grows on demand
	bc.qual_compact = caps->max_records * 16 + 65536;
	for (auto &sl : p->slots)
	{
		PVC3_TRY(dbatch_alloc(ctx, bc, &sl.b));
		HIP_TRY(hipEventCreateWithFlags(&sl.loaded, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&sl.projected, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&sl.downloaded, hipEventDisableTiming));
	}
	g.p = nullptr;
	*out = p;
	return PVC3_OK;
}

void pvc3_pipeline_destroy(pvc3_pipeline *p)
{
	if (!p) return;
	(void) hipSetDevice(p->ctx->device);
	for (hipStream_t s : {p->s_h2d, p->s_kernels, p->s_d2h}) if (s) (void) hipStreamSynchronize(s);
	for (auto &sl : p->slots)
	{
		if (sl.b) pvc3_dbatch_destroy(sl.b);
		for (hipEvent_t e : {sl.loaded, sl.projected, sl.downloaded}) if (e) (void) hipEventDestroy(e);
	}
	for (hipStream_t s : {p->s_h2d, p->s_kernels, p->s_d2h}) if (s) (void) hipStreamDestroy(s);
	delete p;
}

int pvc3_pipeline_submit(pvc3_pipeline *p, pvc3_batch_in const *in, pvc3_batch_out const *out, pvc3_params const *params)
{
	if (!p || !in || !out || !params) return set_error(PVC3_E_INVALID, "pvc3_pipeline_submit: null argument");
	uint64_t const depth = p->slots.size();
	if (p->submitted - p->retired >= depth)
		return set_error(PVC3_E_STATE, "pvc3_pipeline_submit: %llu batches in flight; call pvc3_pipeline_wait first", (unsigned long long) depth);
	uint64_t const n = in->n;
	if (!out->out_cigar_off || !out->run_off || (n && (!out->dst_pos || !out->mate_dst_pos || !out->status)))
		return set_error(PVC3_E_INVALID, "pvc3_pipeline_submit: a required output array is null");
	pvc3_context *ctx = p->ctx;
	PVC3_TRY(check_device(ctx));
	{
		// A change of the parameters is ordered on the stream the kernels of the batches in flight read them on.
		hipStream_t const keep = ctx->stream;
		ctx->stream = p->s_kernels;
		int const rc = upload_params(ctx, *params);
		ctx->stream = keep;
		PVC3_TRY(rc);
	}
	// ensure_scratch / ensure_trace (and the growth after an overflow) free the buffers that queued batches hold as kernel
	// arguments; that is safe only because hipFree waits for everything queued on the device — which also stalls the overlap,
	// so the capacities should be set to what the batches need (options trace_initial, scratch_per_wave) before the first batch.
	PVC3_TRY(ensure_scratch(ctx));
	auto &sl = p->slots[p->submitted % depth];
	pvc3_dbatch *b = sl.b;
	// QUAL stays on the host when the kernels can read it there: both arrays page-locked (pvc3_host_alloc).
	bool const lazy = in->qual && in->qual_off && device_visible_host_pointer(in->qual) && device_visible_host_pointer(in->qual_off);
	PVC3_TRY(dbatch_load(ctx, b, in, p->s_h2d, lazy));
	HIP_TRY(hipEventRecord(sl.loaded, p->s_h2d));
	HIP_TRY(hipStreamWaitEvent(p->s_kernels, sl.loaded, 0));
	hipStream_t const keep = ctx->stream;
	bool const keep_timing = ctx->timing;
	ctx->stream = p->s_kernels;
	ctx->timing = false;			// the context's events would be re-recorded by the next batch: pipelined batches report no kernel times
	b->params = *params; b->attempts = 1; b->finished = false;
	int const rc = enqueue_projection(ctx, b);
	ctx->stream = keep;
	ctx->timing = keep_timing;
	PVC3_TRY(rc);
	b->projected = true;
	HIP_TRY(hipEventRecord(sl.projected, p->s_kernels));
	sl.out = *out; sl.state = 1; sl.error = 0;
	++p->submitted;
	// The copies of this batch are queued; now the previous batch (its kernels ran while the host prepared this one) can
	// start its download, which overlaps this batch's upload and kernels.
	if (p->submitted - p->retired >= 2)
	{
		auto &prev = p->slots[(p->submitted - 2) % depth];
		if (1 == prev.state)
			if (int const e = pipeline_start_download(p, prev)) pipeline_note_error(prev, e);
	}
	return PVC3_OK;
}

int pvc3_pipeline_wait(pvc3_pipeline *p, pvc3_batch_stats *stats)
{
	if (!p) return set_error(PVC3_E_INVALID, "pvc3_pipeline_wait: null argument");
	if (p->submitted == p->retired) return set_error(PVC3_E_STATE, "pvc3_pipeline_wait: no batch in flight");
	PVC3_TRY(check_device(p->ctx));
	auto &sl = p->slots[p->retired % p->slots.size()];
	if (1 == sl.state)
		if (int const e = pipeline_start_download(p, sl)) pipeline_note_error(sl, e);
	++p->retired;
	sl.state = 0;
	if (sl.error)
		return set_error(sl.error, "%s", sl.message);
	HIP_TRY(hipEventSynchronize(sl.downloaded));
	if (stats) fill_stats(p->ctx, sl.b, stats);
	return PVC3_OK;
}

int pvc3_pipeline_in_flight(pvc3_pipeline const *p, uint32_t *count)
{
	if (!p || !count) return set_error(PVC3_E_INVALID, "pvc3_pipeline_in_flight: null argument");
	*count = (uint32_t) (p->submitted - p->retired);
	return PVC3_OK;
}

int pvc3_pipeline_synchronize(pvc3_pipeline *p)
{
	if (!p) return set_error(PVC3_E_INVALID, "pvc3_pipeline_synchronize: null argument");
	PVC3_TRY(check_device(p->ctx));
	for (hipStream_t s : {p->s_h2d, p->s_kernels, p->s_d2h}) HIP_TRY(hipStreamSynchronize(s));
	return PVC3_OK;
}

// ---- recalculate_mapq ----------------------------------------------------------------------------------------------------
int pvc3_rescore_alignments(pvc3_context *ctx, uint64_t n, uint64_t const *cigar_off, uint32_t const *cigar, uint64_t const *seq_off, uint8_t const *seq4,
	uint32_t const *l_seq, uint64_t const *qual_off, uint8_t const *qual, uint64_t const *xn_off, uint8_t const *xn, pvc3_rescore_params const *params, double *score, int32_t *status)
{
	if (!ctx || !params || (n && (!cigar_off || !seq_off || !l_seq || !xn_off || !score || !status))) return set_error(PVC3_E_INVALID, "pvc3_rescore_alignments: null argument");
	PVC3_TRY(check_device(ctx));
	if (0 == n) return PVC3_OK;
	for (uint64_t i = 0; i < n; ++i)		// the kernel reads the slices without further checks
		if (cigar_off[i] > cigar_off[i + 1] || seq_off[i] > seq_off[i + 1] || xn_off[i] > xn_off[i + 1] || seq_off[i + 1] - seq_off[i] < ((uint64_t) l_seq[i] + 1) / 2 || (qual && qual_off && qual_off[i] > qual_off[i + 1]))
			return set_error(PVC3_E_INVALID, "pvc3_rescore_alignments: malformed offsets at record %llu", (unsigned long long) i);
	std::vector <void *> allocs;
	struct guard { std::vector <void *> &a; ~guard() { for (void *p : a) (void) hipFree(p); } } g{allocs};
	hipStream_t const s = ctx->stream;
	uint64_t const *d_coff{}, *d_soff{}, *d_qoff{}, *d_xoff{};
	uint32_t const *d_cig{}, *d_lseq{};
	uint8_t const *d_seq{}, *d_qual{}, *d_xn{};
	double *d_score{}; int32_t *d_status{};
	bool const has_qual = qual && qual_off;
	PVC3_TRY(dev_upload(&d_coff, cigar_off, n + 1, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_cig, cigar, cigar_off[n], 16, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_soff, seq_off, n + 1, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_seq, seq4, seq_off[n], 16, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_lseq, l_seq, n, 0, s, &allocs, nullptr));
	if (has_qual)
	{
		PVC3_TRY(dev_upload(&d_qoff, qual_off, n + 1, 0, s, &allocs, nullptr));
		PVC3_TRY(dev_upload(&d_qual, qual, qual_off[n], 16, s, &allocs, nullptr));
	}
	PVC3_TRY(dev_upload(&d_xoff, xn_off, n + 1, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_xn, xn, xn_off[n], 16, s, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_score, n, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_status, n, &allocs, nullptr));
	if (int const e = launch_rescore(n, d_coff, d_cig, d_soff, d_seq, d_lseq, d_qoff, d_qual, d_xoff, d_xn, *params, d_score, d_status, s)) return set_error(PVC3_E_HIP, "k_rescore_alignments launch: %s", hipGetErrorString((hipError_t) e));
	HIP_TRY(hipMemcpyAsync(score, d_score, n * sizeof(double), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(status, d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	return PVC3_OK;
}

int pvc3_recalculate_mapq(pvc3_context *ctx, pvc3_mapq_batch const *in, uint8_t *mapq, uint8_t *group_emitted, pvc3_mapq_stats *stats)
{
	if (!ctx || !in) return set_error(PVC3_E_INVALID, "pvc3_recalculate_mapq: null argument");
	PVC3_TRY(check_device(ctx));
	uint64_t const ng = in->n_groups;
	if (stats) std::memset(stats, 0, sizeof(*stats));
	if (0 == ng) return PVC3_OK;
	if (!in->group_off || !mapq || !group_emitted) return set_error(PVC3_E_INVALID, "pvc3_recalculate_mapq: null argument");
	uint64_t const n = in->group_off[ng];
	for (uint64_t g = 0; g < ng; ++g)
		if (in->group_off[g] > in->group_off[g + 1]) return set_error(PVC3_E_INVALID, "pvc3_recalculate_mapq: group offsets decrease at group %llu", (unsigned long long) g);
	if (n && (!in->ref_id || !in->pos || !in->mate_ref_id || !in->mate_pos || !in->orig_ref_id || !in->orig_pos || !in->orig_mate_ref_id || !in->orig_mate_pos || !in->score || !in->l_seq || !in->seq_class))
		return set_error(PVC3_E_INVALID, "pvc3_recalculate_mapq: a required input array is null");
	std::vector <void *> allocs;
	struct guard { std::vector <void *> &a; ~guard() { for (void *p : a) (void) hipFree(p); } } g{allocs};
	hipStream_t const s = ctx->stream;
	pvc3_mapq_batch d{};
	d.n_groups = ng;
	PVC3_TRY(dev_upload(&d.group_off, in->group_off, ng + 1, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.ref_id, in->ref_id, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.pos, in->pos, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.mate_ref_id, in->mate_ref_id, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.mate_pos, in->mate_pos, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.orig_ref_id, in->orig_ref_id, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.orig_pos, in->orig_pos, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.orig_mate_ref_id, in->orig_mate_ref_id, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.orig_mate_pos, in->orig_mate_pos, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.score, in->score, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.l_seq, in->l_seq, n, 0, s, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d.seq_class, in->seq_class, n, 0, s, &allocs, nullptr));
	double *d_pw{}, *d_bt{};
	uint32_t *d_ml{};
	uint8_t *d_mapq{}, *d_emitted{};
	unsigned long long *d_stats{};
	PVC3_TRY(dev_alloc(&d_pw, n, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_bt, n, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_ml, n, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_mapq, n, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_emitted, ng, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_stats, 4, &allocs, nullptr));
	HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), s));
	if (int const e = launch_mapq_groups(d, d_pw, d_bt, d_ml, d_mapq, d_emitted, d_stats, s)) return set_error(PVC3_E_HIP, "k_mapq_groups launch: %s", hipGetErrorString((hipError_t) e));
	if (n) HIP_TRY(hipMemcpyAsync(mapq, d_mapq, n, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(group_emitted, d_emitted, ng, hipMemcpyDeviceToHost, s));
	unsigned long long h_stats[4] = {0, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if (stats)
	{
		stats->total_alignments = n;
		stats->unpaired_alignments = h_stats[0];
		stats->reads_with_and_without_mate = h_stats[1];
		stats->mate_not_found = h_stats[2];
		stats->reads_without_valid_position = h_stats[3];
	}
	return PVC3_OK;
}

int pvc3_project_positions(pvc3_context *ctx, uint64_t n, uint32_t const *src_entry, uint32_t const *dst_entry, uint32_t const *pos, uint32_t *out)
{
	if (!ctx || (n && (!src_entry || !dst_entry || !pos || !out))) return set_error(PVC3_E_INVALID, "pvc3_project_positions: null argument");
	PVC3_TRY(check_device(ctx));
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_project_positions: no index has been uploaded to this context");
	if (0 == n) return PVC3_OK;
	std::vector <void *> allocs;
	struct guard { std::vector <void *> &a; ~guard() { for (void *p : a) (void) hipFree(p); } } g{allocs};
	uint32_t const *d_src{}, *d_dst{}, *d_pos{};
	uint32_t *d_out{};
	PVC3_TRY(dev_upload(&d_src, src_entry, n, 0, ctx->stream, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_dst, dst_entry, n, 0, ctx->stream, &allocs, nullptr));
	PVC3_TRY(dev_upload(&d_pos, pos, n, 0, ctx->stream, &allocs, nullptr));
	PVC3_TRY(dev_alloc(&d_out, n, &allocs, nullptr));
	if (int const e = launch_project_positions(ctx->d_entries, ctx->n_entries, n, d_src, d_dst, d_pos, d_out, ctx->stream)) return set_error(PVC3_E_HIP, "k_project_positions launch: %s", hipGetErrorString((hipError_t) e));
	HIP_TRY(hipMemcpyAsync(out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PVC3_OK;
}

}	// extern "C"
