// records_api.cpp — the C ABI of include/pvc3_records.h over the kernels of records.hip and the batch objects of api.cpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "context.hpp"
#include "records_internal.hpp"

using namespace pvc3;

#define HIP_TRY(expr) do { hipError_t const e_ = (expr); if (hipSuccess != e_) return set_error((hipErrorOutOfMemory == e_) ? PVC3_E_NOMEM : PVC3_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define PVC3_TRY(expr) do { int const rc_ = (expr); if (rc_) return rc_; } while (0)
#define LAUNCH_TRY(expr) do { int const e_ = (expr); if (e_) return set_error(PVC3_E_HIP, "%s: %s", #expr, hipGetErrorString((hipError_t) e_)); } while (0)

namespace {
constexpr uint64_t FRONT = UINT64_C(8) << 20;		// bytes in front of a piece for what the previous piece left over (also the largest record the path takes)
constexpr uint64_t SLACK = 4096;
constexpr uint32_t NONE = 0xFFFFFFFFu;
}

struct pvc3_records
{
	pvc3_context *ctx{};
	uint64_t max_stream{};
	uint32_t n_refs{};
	uint8_t *U[PVC3_RECORDS_SLOTS]{};
	uint8_t *d_carry{};
	std::vector <void *> allocs;
	rec_tables T{};
	rec_frame_work W{};
	uint32_t rec_cap{}, blk_cap{};
	rec_state *d_state{}, *h_state{};
	uint32_t *d_blk_off{}, *h_blk_off{};
	uint32_t *d_tile_sums{};
	uint64_t *d_tile_base{};
	uint64_t *d_qual_off{};
	uint32_t *d_out_size{};
	uint64_t *d_out_off{};
	uint64_t *h_total{};				// page-locked: a total read back after a scan
	pvc3_dbatch *batch{};
	uint64_t cigar_cap{};
	// the stream so far
	uint64_t carry_len{}, seen{};
	uint32_t n_pending{}, parity{};
	// the piece at hand
	uint32_t slot{}, end{}, n_take{};
	bool framed{}, projected{};
	uint64_t n_runs{};
	uint8_t *d_plain{};					// pvc3_records_rewrite_to_host: the records one behind the other
	uint64_t plain_cap{};
};

namespace {

template <typename t_value>
int alloc(pvc3_records *r, t_value **dst, uint64_t count)
{
	void *p{};
	HIP_TRY(hipMalloc(&p, (count ? count : 1) * sizeof(t_value)));
	r->allocs.push_back(p);
	*dst = (t_value *) p;
	return PVC3_OK;
}

template <typename t_value>
int upload(pvc3_records *r, t_value const **dst, t_value const *src, uint64_t count)
{
	t_value *p{};
	PVC3_TRY(alloc(r, &p, count));
	if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(t_value), hipMemcpyHostToDevice));
	*dst = p;
	return PVC3_OK;
}

int make_batch(pvc3_records *r, uint64_t cigar_cap)
{
	if (r->batch) { pvc3_dbatch_destroy(r->batch); r->batch = nullptr; }
	PVC3_TRY(batch_object_alloc(r->ctx, r->rec_cap, cigar_cap, &r->batch));
	r->cigar_cap = cigar_cap;
	return PVC3_OK;
}

int read_state(pvc3_records *r)
{
	hipStream_t const s = r->ctx->stream;
	HIP_TRY(hipMemcpyAsync(r->h_state, r->d_state, sizeof(rec_state), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	return PVC3_OK;
}

bool take_error(pvc3_records const *r, int *error_code, uint64_t *error_record)
{
	unsigned long long const k = r->h_state->err_key;
	if (~0ull == k) return false;
	if (error_code) *error_code = (int) (k & 0xff);
	if (error_record) *error_record = k >> 8;
	return true;
}

}	// anonymous namespace

extern "C" {

void pvc3_records_destroy(pvc3_records *r)
{
	if (!r) return;
	(void) hipSetDevice(r->ctx->device);
	(void) hipStreamSynchronize(r->ctx->stream);
	if (r->batch) pvc3_dbatch_destroy(r->batch);
	if (r->d_plain) (void) hipFree(r->d_plain);
	for (void *p : r->allocs) (void) hipFree(p);
	if (r->h_state) (void) hipHostFree(r->h_state);
	if (r->h_blk_off) (void) hipHostFree(r->h_blk_off);
	if (r->h_total) (void) hipHostFree(r->h_total);
	delete r;
}

int pvc3_records_create(pvc3_context *ctx, pvc3_records_config const *c, pvc3_records **out)
{
	if (!ctx || !c || !out) return set_error(PVC3_E_INVALID, "pvc3_records_create: null argument");
	*out = nullptr;
	if (!ctx->d_entries) return set_error(PVC3_E_STATE, "pvc3_records_create: no index has been uploaded to this context");
	if (0 == c->n_refs || !c->src_entry || !c->dst_entry || !c->out_ref_id || !c->ref_names || !c->ref_name_off || (c->n_preserved_tags && !c->preserved_tags))
		return set_error(PVC3_E_INVALID, "pvc3_records_create: a required table is null");
	if (c->max_stream_bytes < 65536 || c->max_stream_bytes + FRONT + SLACK >= (UINT64_C(1) << 32) - (UINT64_C(1) << 20))
		return set_error(PVC3_E_INVALID, "pvc3_records_create: max_stream_bytes must lie between 64 KiB and 4 GiB less %llu MiB", (unsigned long long) ((FRONT >> 20) + 2));
	for (uint32_t i = 0; i < c->n_refs; ++i)
		if (c->src_entry[i] >= ctx->n_entries || c->dst_entry[i] >= ctx->n_entries)
			return set_error(PVC3_E_INVALID, "pvc3_records_create: reference %u maps to an index entry that does not exist", i);
	HIP_TRY(hipSetDevice(ctx->device));
	pvc3_records *r = new (std::nothrow) pvc3_records;
	if (!r) return set_error(PVC3_E_NOMEM, "pvc3_records_create: out of memory");
	struct guard { pvc3_records *p; ~guard() { if (p) pvc3_records_destroy(p); } } g{r};
	r->ctx = ctx; r->max_stream = c->max_stream_bytes; r->n_refs = c->n_refs;

	// the new tags in key order (they must differ: the reference's dictionary would keep the last one put, which no table here expresses)
	struct nt { char const *name; uint32_t kind; };
	nt const names[7] = {{c->query_ranges_tag, NEW_XB}, {c->ref_ranges_tag, NEW_YB}, {c->rec_idx_tag, NEW_XI}, {c->rname_tag, NEW_XR}, {c->pos_tag, NEW_XP}, {c->rnext_tag, NEW_YR}, {c->pnext_tag, NEW_YP}};
	std::vector <std::pair <uint32_t, uint32_t>> tags{{((uint32_t) 'O' << 8) | 'A', NEW_OA}};
	for (auto const &t : names)
		if (t.name[0]) tags.emplace_back(((uint32_t) (uint8_t) t.name[0] << 8) | (uint8_t) t.name[1], t.kind);
	std::sort(tags.begin(), tags.end());
	for (size_t i = 1; i < tags.size(); ++i)
		if (tags[i].first == tags[i - 1].first) return set_error(PVC3_E_INVALID, "pvc3_records_create: two of the tags to write have the same name");
	r->T.n_new = (uint32_t) tags.size();
	for (size_t i = 0; i < tags.size(); ++i) { r->T.new_key[i] = tags[i].first; r->T.new_kind[i] = tags[i].second; }

	PVC3_TRY(upload(r, &r->T.src_entry, c->src_entry, c->n_refs));
	PVC3_TRY(upload(r, &r->T.dst_entry, c->dst_entry, c->n_refs));
	PVC3_TRY(upload(r, &r->T.out_ref_id, c->out_ref_id, c->n_refs));
	PVC3_TRY(upload(r, &r->T.ref_name_off, c->ref_name_off, (uint64_t) c->n_refs + 1));
	{
		// (the names are copied a dword at a time: a few bytes of room behind the last one)
		std::vector <char> names(c->ref_names, c->ref_names + c->ref_name_off[c->n_refs]);
		names.resize(names.size() + 16, 0);
		PVC3_TRY(upload(r, &r->T.ref_names, names.data(), names.size()));
	}
	{
		std::vector <uint32_t> bits(2048, 0);
		for (uint32_t i = 0; i < c->n_preserved_tags; ++i) bits[c->preserved_tags[i] >> 5] |= 1u << (c->preserved_tags[i] & 31);
		PVC3_TRY(upload(r, &r->T.preserved, bits.data(), bits.size()));
	}
	PVC3_TRY(alloc(r, &r->T.removed_counts, 65536));
	HIP_TRY(hipMemset(r->T.removed_counts, 0, 65536 * sizeof(unsigned long long)));

	uint64_t const u_bytes = FRONT + r->max_stream + SLACK;
	for (auto &u : r->U) { PVC3_TRY(alloc(r, &u, u_bytes)); HIP_TRY(hipMemset(u + FRONT + r->max_stream, 0, SLACK)); }
	PVC3_TRY(alloc(r, &r->d_carry, FRONT));
	// No record is shorter than 37 bytes with its length field: what a piece can hold at most.
	r->rec_cap = (uint32_t) ((FRONT + r->max_stream) / 36 + 8);
	r->blk_cap = (uint32_t) (r->max_stream / 18 / 64 + 1024);		// (callers list a few thousand blocks; grows on demand)
	PVC3_TRY(alloc(r, &r->d_state, 1));
	HIP_TRY(hipMemset(r->d_state, 0, sizeof(rec_state)));
	HIP_TRY(hipHostMalloc((void **) &r->h_state, sizeof(rec_state), hipHostMallocDefault));
	HIP_TRY(hipHostMalloc((void **) &r->h_total, 64, hipHostMallocDefault));
	HIP_TRY(hipHostMalloc((void **) &r->h_blk_off, ((uint64_t) r->blk_cap + 2) * sizeof(uint32_t), hipHostMallocDefault));
	PVC3_TRY(alloc(r, &r->d_blk_off, (uint64_t) r->blk_cap + 2));
	for (uint32_t **p : {&r->W.guess, &r->W.cnt, &r->W.exits, &r->W.tails, &r->W.rec_base}) PVC3_TRY(alloc(r, p, (uint64_t) r->blk_cap + 2));
	PVC3_TRY(alloc(r, &r->W.rec_off, r->rec_cap));
	PVC3_TRY(alloc(r, &r->W.keep, r->rec_cap));
	PVC3_TRY(alloc(r, &r->W.kept_pos, (uint64_t) r->rec_cap + 1));
	PVC3_TRY(alloc(r, &r->W.kept_off, r->rec_cap));
	PVC3_TRY(alloc(r, &r->W.kept_input, r->rec_cap));
	PVC3_TRY(alloc(r, &r->W.n_cigar, r->rec_cap));
	PVC3_TRY(alloc(r, &r->d_tile_sums, (uint64_t) rec_scan_tiles(r->rec_cap) + 1));
	PVC3_TRY(alloc(r, &r->d_tile_base, (uint64_t) rec_scan_tiles(r->rec_cap) + 2));
	PVC3_TRY(alloc(r, &r->d_qual_off, (uint64_t) r->rec_cap + 1));
	PVC3_TRY(alloc(r, &r->d_out_size, r->rec_cap));
	PVC3_TRY(alloc(r, &r->d_out_off, (uint64_t) r->rec_cap + 1));
	PVC3_TRY(make_batch(r, std::max <uint64_t>(UINT64_C(4) << 20, r->max_stream / 16)));
	HIP_TRY(hipDeviceSynchronize());
	g.p = nullptr;
	*out = r;
	return PVC3_OK;
}

int pvc3_records_buffer(pvc3_records *r, uint32_t slot, void **base, uint64_t *front)
{
	if (!r || slot >= PVC3_RECORDS_SLOTS || !base || !front) return set_error(PVC3_E_INVALID, "pvc3_records_buffer: bad argument");
	*base = r->U[slot]; *front = FRONT;
	return PVC3_OK;
}

int pvc3_records_upload(pvc3_records *r, uint32_t slot, uint64_t offset, void const *bytes, uint64_t n_bytes)
{
	if (!r || slot >= PVC3_RECORDS_SLOTS || (!bytes && n_bytes) || offset > r->max_stream || n_bytes > r->max_stream - offset) return set_error(PVC3_E_INVALID, "pvc3_records_upload: bad argument");
	HIP_TRY(hipSetDevice(r->ctx->device));
	HIP_TRY(r->ctx->stager.h2d(r->U[slot] + FRONT + offset, bytes, n_bytes, r->ctx->stream));
	return PVC3_OK;
}

int pvc3_records_frame(pvc3_records *r, uint32_t slot, uint64_t n_bytes, uint64_t const *block_off, uint32_t n_blocks, uint64_t skip, int last,
	pvc3_records_frame_result *result, int *error_code, uint64_t *error_record)
{
	if (!r || slot >= PVC3_RECORDS_SLOTS || !result || (n_blocks && !block_off)) return set_error(PVC3_E_INVALID, "pvc3_records_frame: bad argument");
	if (n_bytes > r->max_stream || skip > n_bytes) return set_error(PVC3_E_INVALID, "pvc3_records_frame: %llu bytes exceed the %llu this object was made for", (unsigned long long) n_bytes, (unsigned long long) r->max_stream);
	if (skip && (r->carry_len || r->seen)) return set_error(PVC3_E_INVALID, "pvc3_records_frame: bytes can only be skipped at the beginning of the stream");
	if (error_code) *error_code = PVC3_REC_OK;
	pvc3_context *ctx = r->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t const s = ctx->stream;
	uint8_t *U = r->U[slot];
	r->framed = false; r->projected = false;

	uint32_t const start0 = (uint32_t) (FRONT - r->carry_len + skip), end = (uint32_t) (FRONT + n_bytes);
	if (r->carry_len) HIP_TRY(hipMemcpyAsync(U + FRONT - r->carry_len, r->d_carry, r->carry_len, hipMemcpyDeviceToDevice, s));
	// where the file's blocks begin in the buffer (the first "block" begins with what was carried over)
	if (n_blocks + 2 > r->blk_cap)
	{
		HIP_TRY(hipStreamSynchronize(s));
		uint32_t const cap = n_blocks + n_blocks / 4 + 1024;
		if (r->h_blk_off) { (void) hipHostFree(r->h_blk_off); r->h_blk_off = nullptr; }
		HIP_TRY(hipHostMalloc((void **) &r->h_blk_off, ((uint64_t) cap + 2) * sizeof(uint32_t), hipHostMallocDefault));
		PVC3_TRY(alloc(r, &r->d_blk_off, (uint64_t) cap + 2));		// (the old arrays stay until the object goes: a handful of KB)
		for (uint32_t **p : {&r->W.guess, &r->W.cnt, &r->W.exits, &r->W.tails, &r->W.rec_base}) PVC3_TRY(alloc(r, p, (uint64_t) cap + 2));
		r->blk_cap = cap;
	}
	uint32_t nb = 0;
	for (uint32_t b = 0; b < n_blocks; ++b)
	{
		if (block_off[b] > n_bytes || (b && block_off[b] < block_off[b - 1])) return set_error(PVC3_E_INVALID, "pvc3_records_frame: block offsets must not decrease or pass the piece");
		if (b && block_off[b] == block_off[b - 1]) continue;		// (an empty block)
		r->h_blk_off[nb++] = (uint32_t) (FRONT + block_off[b]);
	}
	if (0 == nb) r->h_blk_off[nb++] = (uint32_t) FRONT;
	r->h_blk_off[nb] = end;
	if (nb > 1 && r->h_blk_off[nb - 1] == end) --nb;
	HIP_TRY(hipMemcpyAsync(r->d_blk_off, r->h_blk_off, ((uint64_t) nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));

	LAUNCH_TRY(rec_launch_reset(r->d_state, s));
	LAUNCH_TRY(rec_launch_frame(U, r->d_blk_off, nb, start0, end, r->n_refs, r->W, r->d_state, s));
	// how many records there are decides the grids of what follows (the arrays are made for the most a piece can hold, forty
	// times what it usually holds: grids of that size cost a millisecond a piece)
	PVC3_TRY(read_state(r));
	uint32_t const n_found = r->h_state->n_records;
	if (n_found > r->rec_cap) return set_error(PVC3_E_STATE, "pvc3_records_frame: more records than a piece can hold");
	LAUNCH_TRY(rec_launch_filter(U, n_found, r->W, r->d_state, r->n_refs, r->n_pending, r->parity, r->seen, s));
	LAUNCH_TRY(rec_launch_scan(r->W.keep, nullptr, n_found, r->W.kept_pos, r->d_tile_sums, r->d_tile_base, s));
	LAUNCH_TRY(rec_launch_compact(n_found, r->W, r->d_state, r->n_pending, r->parity, r->seen, last, s));
	LAUNCH_TRY(rec_launch_carry(U, end, r->d_state, r->d_carry, (uint32_t) FRONT, s));
	PVC3_TRY(read_state(r));
	rec_state const &st = *r->h_state;
	if (NONE != st.frame_error_at)
	{
		if (error_code) *error_code = PVC3_REC_E_TRUNCATED;
		if (error_record) *error_record = r->seen + ((st.n_records > r->n_pending) ? (st.n_records - r->n_pending) : 0);
		return set_error(PVC3_E_INVALID, "pvc3_records_frame: truncated BAM record (block_size below 32) at byte %u of the piece", st.frame_error_at - (uint32_t) FRONT);
	}
	if (take_error(r, error_code, error_record))
		return set_error(PVC3_E_INVALID, "pvc3_records_frame: malformed BAM record (input record %llu, code %d)", (unsigned long long) (st.err_key >> 8), (int) (st.err_key & 0xff));
	if (st.carry_len > FRONT)
	{
		if (error_code) *error_code = PVC3_REC_E_CARRY;
		if (error_record) *error_record = r->seen + (st.n_records - r->n_pending);
		return set_error(PVC3_E_INVALID, "pvc3_records_frame: a record of more than %llu MiB", (unsigned long long) (FRONT >> 20));
	}
	if (last && st.tail_off != end)
	{
		if (error_code) *error_code = PVC3_REC_E_TRUNCATED;
		if (error_record) *error_record = r->seen + (st.n_records - r->n_pending);
		return set_error(PVC3_E_INVALID, "pvc3_records_frame: the stream ends inside a record");
	}

	// ---- the batch arrays -----------------------------------------------------------------------------------------------------------
	uint32_t const n_take = st.n_take;
	pvc3_dbatch *b = r->batch;
	for (int attempt = 0; ; ++attempt)
	{
		dev_batch_in &d = b->in;
		rec_batch_arrays B{};
		B.src_pos = const_cast <uint32_t *>(d.src_pos); B.src_entry = const_cast <uint32_t *>(d.src_entry); B.dst_entry = const_cast <uint32_t *>(d.dst_entry);
		B.flags = const_cast <uint32_t *>(d.flags); B.l_seq = const_cast <uint32_t *>(d.l_seq); B.seq_off = const_cast <uint64_t *>(d.seq_off); B.qual_off = r->d_qual_off;
		B.mate_se = b->up_mate_src_entry; B.mate_de = b->up_mate_dst_entry; B.mate_pos = b->up_mate_pos;
		LAUNCH_TRY(rec_launch_scatter(U, end, n_take, r->W, r->d_state, r->T, B, s));
		LAUNCH_TRY(rec_launch_scan(r->W.n_cigar, nullptr, n_take, const_cast <uint64_t *>(d.cigar_off), r->d_tile_sums, r->d_tile_base, s));
		HIP_TRY(hipMemcpyAsync(r->h_total, d.cigar_off + n_take, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		uint64_t const n_cigar = *r->h_total;
		if (n_cigar + 64 > r->cigar_cap)
		{
			if (attempt) return set_error(PVC3_E_NOMEM, "pvc3_records_frame: the batch object did not grow");
			PVC3_TRY(make_batch(r, 2 * n_cigar + 4096));
			b = r->batch;
			continue;
		}
		LAUNCH_TRY(rec_launch_scatter_cigar(U, n_take, r->W, r->d_state, d.cigar_off, r->cigar_cap, const_cast <uint32_t *>(d.cigar), s));
		d.n = n_take; b->n = n_take;
		d.seq4 = U; d.qual = U; d.qual_off = r->d_qual_off;
		d.mate_src_entry = b->up_mate_src_entry; d.mate_dst_entry = b->up_mate_dst_entry; d.mate_pos = b->up_mate_pos;
		b->work.host_qual = nullptr; b->work.host_qual_off = nullptr;
		HIP_TRY(hipMemsetAsync(b->d_invalid, 0xff, sizeof(uint32_t), s));		// the layout is this object's own: nothing to refuse
		b->projected = false; b->finished = false;
		break;
	}

	result->n_new_records = st.n_records - r->n_pending;
	result->n_batch = n_take;
	result->carry_bytes = st.carry_len;
	result->matched = st.counts[0]; result->flags_not_matched = st.counts[1]; result->ref_id_missing = st.counts[2]; result->seq_missing = st.counts[3];
	r->seen += st.n_records - r->n_pending;
	r->n_pending = st.n_deferred;
	r->parity ^= 1;
	r->carry_len = st.carry_len;
	r->slot = slot; r->end = end; r->n_take = n_take;
	r->framed = true;
	return PVC3_OK;
}

int pvc3_records_project(pvc3_records *r, pvc3_params const *params, pvc3_batch_stats *stats)
{
	if (!r || !params) return set_error(PVC3_E_INVALID, "pvc3_records_project: null argument");
	if (!r->framed) return set_error(PVC3_E_STATE, "pvc3_records_project: pvc3_records_frame has not succeeded");
	if (stats) std::memset(stats, 0, sizeof(*stats));
	r->n_runs = 0;
	if (r->n_take)
	{
		PVC3_TRY(batch_object_project(r->ctx, r->batch, params));
		pvc3_batch_stats st{};
		PVC3_TRY(batch_object_complete(r->ctx, r->batch, &st));
		r->n_runs = st.n_runs;
		if (stats) *stats = st;
	}
	r->projected = true;
	return PVC3_OK;
}

int pvc3_records_rewrite(pvc3_records *r, void *d_out, uint64_t capacity_bytes, uint32_t block_bytes, uint32_t block_stride, uint32_t block_pad,
	uint64_t *n_bytes, uint64_t *n_runs, int *error_code, uint64_t *error_record)
{
	if (!r || !n_bytes || (!d_out && capacity_bytes) || 0 == block_bytes || block_stride < block_bytes + block_pad) return set_error(PVC3_E_INVALID, "pvc3_records_rewrite: bad argument");
	if (!r->projected) return set_error(PVC3_E_STATE, "pvc3_records_rewrite: pvc3_records_project has not succeeded");
	if (error_code) *error_code = PVC3_REC_OK;
	*n_bytes = 0;
	if (n_runs) *n_runs = r->n_runs;
	if (0 == r->n_take) return PVC3_OK;
	pvc3_context *ctx = r->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t const s = ctx->stream;
	pvc3_dbatch *b = r->batch;
	rec_results R{};
	R.dst_pos = b->work.dst_pos; R.mate_dst_pos = b->work.mate_dst_pos; R.rec_flags = b->work.rec_flags;
	R.out_cigar_off = b->packed.out_cigar_off; R.out_cigar = b->packed.out_cigar; R.run_off = b->packed.run_off; R.runs = b->packed.runs;
	uint8_t const *U = r->U[r->slot];
	LAUNCH_TRY(rec_launch_rewrite(false, U, r->n_take, r->W, r->d_state, r->T, R, r->d_out_size, r->d_out_off, nullptr, block_bytes, block_stride, block_pad, s));
	LAUNCH_TRY(rec_launch_scan(r->d_out_size, nullptr, r->n_take, r->d_out_off, r->d_tile_sums, r->d_tile_base, s));
	HIP_TRY(hipMemcpyAsync(r->h_total, r->d_out_off + r->n_take, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	PVC3_TRY(read_state(r));
	if (take_error(r, error_code, error_record))
		return set_error(PVC3_E_INVALID, "pvc3_records_rewrite: input record %llu cannot be written (code %d)", (unsigned long long) (r->h_state->err_key >> 8), (int) (r->h_state->err_key & 0xff));
	uint64_t const total = *r->h_total;
	*n_bytes = total;
	if (total > capacity_bytes) return set_error(PVC3_E_CAPACITY, "pvc3_records_rewrite: %llu bytes do not fit %llu", (unsigned long long) total, (unsigned long long) capacity_bytes);
	LAUNCH_TRY(rec_launch_rewrite(true, U, r->n_take, r->W, r->d_state, r->T, R, r->d_out_size, r->d_out_off, (uint8_t *) d_out, block_bytes, block_stride, block_pad, s));
	HIP_TRY(hipStreamSynchronize(s));
	return PVC3_OK;
}

int pvc3_records_rewrite_to_host(pvc3_records *r, void *dst, uint64_t capacity_bytes, uint64_t *n_bytes, uint64_t *n_runs, int *error_code, uint64_t *error_record)
{
	if (!r || !n_bytes || (!dst && capacity_bytes)) return set_error(PVC3_E_INVALID, "pvc3_records_rewrite_to_host: bad argument");
	if (r->plain_cap < capacity_bytes + 64)
	{
		HIP_TRY(hipSetDevice(r->ctx->device));
		HIP_TRY(hipStreamSynchronize(r->ctx->stream));
		if (r->d_plain) { (void) hipFree(r->d_plain); r->d_plain = nullptr; r->plain_cap = 0; }
		HIP_TRY(hipMalloc((void **) &r->d_plain, capacity_bytes + 64));
		r->plain_cap = capacity_bytes + 64;
	}
	// one "block" as long as the whole: the bytes lie one behind the other
	PVC3_TRY(pvc3_records_rewrite(r, r->d_plain, capacity_bytes, 0x80000000u, 0x80000000u, 0, n_bytes, n_runs, error_code, error_record));
	if (*n_bytes) HIP_TRY(r->ctx->stager.d2h(dst, r->d_plain, *n_bytes, r->ctx->stream));
	return PVC3_OK;
}

int pvc3_records_runs(pvc3_records *r, pvc3_run *runs, uint64_t n_runs)
{
	if (!r || (n_runs && !runs)) return set_error(PVC3_E_INVALID, "pvc3_records_runs: null argument");
	if (!r->projected || n_runs != r->n_runs) return set_error(PVC3_E_STATE, "pvc3_records_runs: the batch has %llu ranges", (unsigned long long) r->n_runs);
	if (n_runs) HIP_TRY(r->ctx->stager.d2h(runs, r->batch->packed.runs, n_runs * sizeof(pvc3_run), r->ctx->stream));
	return PVC3_OK;
}

int pvc3_records_removed_tags(pvc3_records *r, uint64_t *counts)
{
	if (!r || !counts) return set_error(PVC3_E_INVALID, "pvc3_records_removed_tags: null argument");
	std::vector <unsigned long long> h(65536);
	HIP_TRY(hipStreamSynchronize(r->ctx->stream));
	HIP_TRY(hipMemcpy(h.data(), r->T.removed_counts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemset(r->T.removed_counts, 0, h.size() * sizeof(unsigned long long)));
	for (size_t i = 0; i < h.size(); ++i) counts[i] += h[i];
	return PVC3_OK;
}

int pvc3_records_batch_sizes(pvc3_records *r, uint64_t *n, uint64_t *n_cigar)
{
	if (!r || !n || !n_cigar) return set_error(PVC3_E_INVALID, "pvc3_records_batch_sizes: null argument");
	if (!r->framed) return set_error(PVC3_E_STATE, "pvc3_records_batch_sizes: no batch");
	*n = r->n_take;
	*n_cigar = 0;
	if (r->n_take) HIP_TRY(hipMemcpy(n_cigar, r->batch->in.cigar_off + r->n_take, sizeof(uint64_t), hipMemcpyDeviceToHost));
	return PVC3_OK;
}

int pvc3_records_download_batch(pvc3_records *r, uint32_t *src_pos, uint32_t *src_entry, uint32_t *dst_entry, uint32_t *flags, uint64_t *cigar_off, uint32_t *cigar,
	uint64_t *seq_off, uint32_t *l_seq, uint64_t *qual_off, uint32_t *mate_src_entry, uint32_t *mate_dst_entry, uint32_t *mate_pos, uint64_t *record_off, uint64_t *input_index)
{
	if (!r) return set_error(PVC3_E_INVALID, "pvc3_records_download_batch: null argument");
	if (!r->framed) return set_error(PVC3_E_STATE, "pvc3_records_download_batch: no batch");
	uint64_t const n = r->n_take;
	dev_batch_in const &d = r->batch->in;
	HIP_TRY(hipStreamSynchronize(r->ctx->stream));
	auto get = [&](void *dst, void const *src, uint64_t bytes) -> int { if (dst && bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return PVC3_OK; };
	PVC3_TRY(get(src_pos, d.src_pos, 4 * n)); PVC3_TRY(get(src_entry, d.src_entry, 4 * n)); PVC3_TRY(get(dst_entry, d.dst_entry, 4 * n)); PVC3_TRY(get(flags, d.flags, 4 * n));
	PVC3_TRY(get(cigar_off, d.cigar_off, 8 * (n + 1)));
	if (cigar && n) { uint64_t nc = 0; HIP_TRY(hipMemcpy(&nc, d.cigar_off + n, 8, hipMemcpyDeviceToHost)); PVC3_TRY(get(cigar, d.cigar, 4 * nc)); }
	PVC3_TRY(get(seq_off, d.seq_off, 8 * (n + 1))); PVC3_TRY(get(l_seq, d.l_seq, 4 * n)); PVC3_TRY(get(qual_off, r->d_qual_off, 8 * (n + 1)));
	PVC3_TRY(get(mate_src_entry, d.mate_src_entry, 4 * n)); PVC3_TRY(get(mate_dst_entry, d.mate_dst_entry, 4 * n)); PVC3_TRY(get(mate_pos, d.mate_pos, 4 * n));
	if (record_off && n)
	{
		std::vector <uint32_t> off(n);
		PVC3_TRY(get(off.data(), r->W.kept_off, 4 * n));
		for (uint64_t i = 0; i < n; ++i) record_off[i] = off[i];
	}
	PVC3_TRY(get(input_index, r->W.kept_input, 8 * n));
	return PVC3_OK;
}

}	// extern "C"
