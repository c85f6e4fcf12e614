// split_api.cpp — pvc3_reference_* and pvc3_split_match_ops (include/pvc3.h) over the kernel of split_ops.hip.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "context.hpp"
#include "records_internal.hpp"		// rec_launch_scan
#include "split_internal.hpp"

using namespace pvc3;

#define HIP_TRY(expr) do { hipError_t const e_ = (expr); if (hipSuccess != e_) return set_error((hipErrorOutOfMemory == e_) ? PVC3_E_NOMEM : PVC3_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define LAUNCH_TRY(expr) do { int const e_ = (expr); if (e_) return set_error(PVC3_E_HIP, "%s: %s", #expr, hipGetErrorString((hipError_t) e_)); } while (0)

struct pvc3_reference
{
	pvc3_context *ctx{};
	uint32_t n{};
	std::vector <uint8_t *> bytes;		// device copies
	uint8_t **d_bytes{};
	uint64_t *d_len{};
};

namespace {

// The device arrays of one call; freed when the call ends, whatever way it ends.
struct scoped_arrays
{
	std::vector <void *> all;
	~scoped_arrays() { for (void *p : all) (void) hipFree(p); }
	template <typename t_value> hipError_t make(t_value **dst, uint64_t count)
	{
		void *p{};
		hipError_t const e = hipMalloc(&p, (count ? count : 1) * sizeof(t_value));
		if (hipSuccess == e) { all.push_back(p); *dst = (t_value *) p; }
		return e;
	}
	template <typename t_value> hipError_t upload(t_value const **dst, t_value const *src, uint64_t count, hipStream_t s)
	{
		t_value *p{};
		hipError_t e = make(&p, count);
		if (hipSuccess == e && count) e = hipMemcpyAsync(p, src, count * sizeof(t_value), hipMemcpyHostToDevice, s);
		*dst = p;
		return e;
	}
};

}

extern "C" {

void pvc3_reference_destroy(pvc3_reference *ref)
{
	if (!ref) return;
	(void) hipSetDevice(ref->ctx->device);
	for (uint8_t *p : ref->bytes) if (p) (void) hipFree(p);
	if (ref->d_bytes) (void) hipFree(ref->d_bytes);
	if (ref->d_len) (void) hipFree(ref->d_len);
	delete ref;
}

int pvc3_reference_create(pvc3_context *ctx, uint32_t n_sequences, uint8_t const *const *bytes, uint64_t const *lengths, pvc3_reference **out)
{
	if (!ctx || !out || (n_sequences && (!bytes || !lengths))) return set_error(PVC3_E_INVALID, "pvc3_reference_create: null argument");
	*out = nullptr;
	HIP_TRY(hipSetDevice(ctx->device));
	pvc3_reference *ref = new (std::nothrow) pvc3_reference;
	if (!ref) return set_error(PVC3_E_NOMEM, "pvc3_reference_create: out of memory");
	ref->ctx = ctx; ref->n = n_sequences;
	ref->bytes.assign(n_sequences, nullptr);
	auto fail = [&](hipError_t e, char const *what) { pvc3_reference_destroy(ref); return set_error((hipErrorOutOfMemory == e) ? PVC3_E_NOMEM : PVC3_E_HIP, "pvc3_reference_create: %s: %s", what, hipGetErrorString(e)); };
	for (uint32_t i = 0; i < n_sequences; ++i)
	{
		if (lengths[i] && !bytes[i]) { pvc3_reference_destroy(ref); return set_error(PVC3_E_INVALID, "pvc3_reference_create: sequence %u has a length and no bytes", i); }
		hipError_t e = hipMalloc((void **) &ref->bytes[i], lengths[i] ? lengths[i] : 1);
		if (hipSuccess != e) return fail(e, "hipMalloc");
		if (lengths[i] && hipSuccess != (e = hipMemcpy(ref->bytes[i], bytes[i], lengths[i], hipMemcpyHostToDevice))) return fail(e, "hipMemcpy");
	}
	hipError_t e = hipMalloc((void **) &ref->d_bytes, (n_sequences ? n_sequences : 1) * sizeof(uint8_t *));
	if (hipSuccess != e) return fail(e, "hipMalloc");
	if (hipSuccess != (e = hipMalloc((void **) &ref->d_len, (n_sequences ? n_sequences : 1) * sizeof(uint64_t)))) return fail(e, "hipMalloc");
	if (n_sequences)
	{
		if (hipSuccess != (e = hipMemcpy(ref->d_bytes, ref->bytes.data(), n_sequences * sizeof(uint8_t *), hipMemcpyHostToDevice))) return fail(e, "hipMemcpy");
		if (hipSuccess != (e = hipMemcpy(ref->d_len, lengths, n_sequences * sizeof(uint64_t), hipMemcpyHostToDevice))) return fail(e, "hipMemcpy");
	}
	*out = ref;
	return PVC3_OK;
}

int pvc3_split_match_ops(pvc3_context *ctx, pvc3_reference const *ref, pvc3_split_in const *in, pvc3_split_out *out, uint64_t *n_cigar, uint64_t *n_map_bytes)
{
	if (!ctx || !ref || !in || !out || !n_cigar || !n_map_bytes) return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: null argument");
	if (ref->ctx != ctx) return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: the reference belongs to another context");
	*n_cigar = 0; *n_map_bytes = 0;
	uint64_t const n = in->n;
	if (n >= (UINT64_C(1) << 31)) return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: more than 2^31 records in one call");
	if (!out->cigar_off || !out->n_map_off || (n && !out->status)) return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: an output array is null");
	if (0 == n) { out->cigar_off[0] = 0; out->n_map_off[0] = 0; return PVC3_OK; }
	if (!in->sequence || !in->pos || !in->cigar_off || !in->seq_off || !in->l_seq || (in->cigar_off[n] && !in->cigar) || (in->seq4_bytes && !in->seq4))
		return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: an input array is null");
	for (uint64_t i = 0; i < n; ++i)
	{
		if (in->cigar_off[i + 1] < in->cigar_off[i]) return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: cigar_off decreases at record %llu", (unsigned long long) i);
		if (in->seq_off[i] > in->seq4_bytes || ((uint64_t) in->l_seq[i] + 1) / 2 > in->seq4_bytes - in->seq_off[i])
			return set_error(PVC3_E_INVALID, "pvc3_split_match_ops: the bases of record %llu lie outside seq4", (unsigned long long) i);
	}
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t const s = ctx->stream;
	scoped_arrays A;
	split_work W{};
	W.n = n; W.n_sequences = ref->n; W.ref_bytes = ref->d_bytes; W.ref_len = ref->d_len;
	HIP_TRY(A.upload(&W.sequence, in->sequence, n, s));
	HIP_TRY(A.upload(&W.pos, in->pos, n, s));
	HIP_TRY(A.upload(&W.l_seq, in->l_seq, n, s));
	HIP_TRY(A.upload(&W.cigar_off, in->cigar_off, n + 1, s));
	HIP_TRY(A.upload(&W.seq_off, in->seq_off, n, s));
	HIP_TRY(A.upload(&W.cigar, in->cigar, in->cigar_off[n], s));
	HIP_TRY(A.upload(&W.seq4, in->seq4, in->seq4_bytes, s));
	uint64_t *d_cigar_off{}, *d_map_off{}, *tile_base{};
	uint32_t *tile_sums{};
	HIP_TRY(A.make(&W.status, n));
	HIP_TRY(A.make(&W.n_out, n));
	HIP_TRY(A.make(&W.n_map, n));
	HIP_TRY(A.make(&d_cigar_off, n + 1));
	HIP_TRY(A.make(&d_map_off, n + 1));
	HIP_TRY(A.make(&tile_sums, (uint64_t) rec_scan_tiles((uint32_t) n) + 1));
	HIP_TRY(A.make(&tile_base, (uint64_t) rec_scan_tiles((uint32_t) n) + 2));
	LAUNCH_TRY(launch_split_match_ops(false, W, s));
	LAUNCH_TRY(rec_launch_scan(W.n_out, nullptr, (uint32_t) n, d_cigar_off, tile_sums, tile_base, s));
	LAUNCH_TRY(rec_launch_scan(W.n_map, nullptr, (uint32_t) n, d_map_off, tile_sums, tile_base, s));
	uint64_t totals[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(&totals[0], d_cigar_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(&totals[1], d_map_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	*n_cigar = totals[0]; *n_map_bytes = totals[1];
	if (totals[0] > out->cigar_cap || totals[1] > out->n_map_cap || (totals[0] && !out->cigar) || (totals[1] && !out->n_map))
		return set_error(PVC3_E_CAPACITY, "pvc3_split_match_ops: %llu operations and %llu bytes of N positions are needed", (unsigned long long) totals[0], (unsigned long long) totals[1]);
	W.out_cigar_off = d_cigar_off; W.out_map_off = d_map_off;
	HIP_TRY(A.make(&W.out_cigar, totals[0]));
	HIP_TRY(A.make(&W.out_map, totals[1]));
	LAUNCH_TRY(launch_split_match_ops(true, W, s));
	HIP_TRY(hipMemcpyAsync(out->cigar_off, d_cigar_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(out->n_map_off, d_map_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(out->status, W.status, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
	if (totals[0]) HIP_TRY(hipMemcpyAsync(out->cigar, W.out_cigar, totals[0] * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	if (totals[1]) HIP_TRY(hipMemcpyAsync(out->n_map, W.out_map, totals[1], hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	return PVC3_OK;
}

}
