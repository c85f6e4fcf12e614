// split_ops.hip — the rewrite_cigar tool's sequence-match mode on the GPU (include/pvc3.h: pvc3_split_match_ops).
// rewrite-cigar/main.cc:191-284 (rewrite_cigar_sequence_match) and :126-148 (fill_ref_n_positions): lane = record, the walk over the
// record's operations done twice — once to count what comes out, once, after a scan of the counts, to write it.
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "split_internal.hpp"

namespace pvc3 {

namespace {

// seqan3::dna5::to_char of a BAM base code
__device__ __forceinline__ uint32_t dna5_char_of_code(uint32_t code) { return (1 == code) ? 'A' : (2 == code) ? 'C' : (4 == code) ? 'G' : (8 == code) ? 'T' : 'N'; }

template <bool t_emit>
__global__ __launch_bounds__(256) void k_split_match_ops(split_work const W)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < W.n; i += stride)
	{
		if (t_emit && W.status[i]) continue;
		uint32_t const *ops = W.cigar + W.cigar_off[i];
		uint32_t const n_ops = (uint32_t) (W.cigar_off[i + 1] - W.cigar_off[i]);
		uint8_t const *seq = W.seq4 + W.seq_off[i];
		uint32_t const l_seq = W.l_seq[i], which = W.sequence[i];
		uint64_t const base = W.pos[i];
		uint8_t const *ref = (which < W.n_sequences) ? W.ref_bytes[which] : nullptr;
		uint64_t const ref_len = (which < W.n_sequences) ? W.ref_len[which] : 0;
		uint32_t *out = t_emit ? (W.out_cigar + W.out_cigar_off[i]) : nullptr;
		uint32_t n_out = 0, status = ref ? PVC3_SPLIT_OK : PVC3_SPLIT_REF_RANGE;
		uint64_t q = 0, r = 0;
		for (uint32_t k = 0; k < n_ops && PVC3_SPLIT_OK == status; ++k)
		{
			uint32_t const c = ops[k], op = c & 0xf, cnt = c >> 4;
			bool const uses_ref = (0 == op || 2 == op || 3 == op || 7 == op || 8 == op);
			if (op > 8) { status = PVC3_SPLIT_BAD_OP; break; }
			if (uses_ref && base + r + cnt > ref_len) { status = PVC3_SPLIT_REF_RANGE; break; }		// (fill_ref_n_positions reads ref[base + r + i])
			if (0 != op)
			{
				if (t_emit) out[n_out] = c;
				++n_out;
				if (1 == op || 4 == op || 7 == op || 8 == op) q += cnt;		// I S = X consume the query
				if (uses_ref) r += cnt;
				continue;
			}
			if (0 == cnt) { status = PVC3_SPLIT_EMPTY_M; break; }
			if (q + cnt > l_seq) { status = PVC3_SPLIT_QUERY_RANGE; break; }
			uint32_t prev_cnt = 0, prev_match = 0;
			for (uint32_t x = 0; x < cnt; ++x)
			{
				uint64_t const qi = q + x;
				uint32_t const code = (seq[qi >> 1] >> ((~qi & 1) << 2)) & 0xf;
				uint32_t const match = dna5_char_of_code(code) == ref[base + r + x];
				if (x && match != prev_match)
				{
					if (t_emit) out[n_out] = (prev_cnt << 4) | (prev_match ? 7u : 8u);
					++n_out;
					prev_cnt = 0;
				}
				prev_match = match; ++prev_cnt;
			}
			if (t_emit) out[n_out] = (prev_cnt << 4) | (prev_match ? 7u : 8u);
			++n_out;
			q += cnt; r += cnt;
		}
		if (!t_emit)
		{
			W.status[i] = (int32_t) status;
			W.n_out[i] = (PVC3_SPLIT_OK == status) ? n_out : 0;
			W.n_map[i] = (PVC3_SPLIT_OK == status) ? (uint32_t) ((r + 7) / 8) : 0;
		}
		else
		{
			uint8_t *map = W.out_map + W.out_map_off[i];
			for (uint64_t at = 0; at < r; at += 8)
			{
				uint32_t bits = 0;
				for (uint32_t b = 0; b < 8 && at + b < r; ++b) bits |= ('N' == ref[base + at + b]) ? (1u << b) : 0u;
				map[at >> 3] = (uint8_t) bits;
			}
		}
	}
}

}

int launch_split_match_ops(bool emit, split_work const &W, void *stream)
{
	if (0 == W.n) return 0;
	uint32_t const grid = (uint32_t) ((W.n + 255) / 256 < 65536 ? (W.n + 255) / 256 : 65536);
	if (emit) hipLaunchKernelGGL(k_split_match_ops <true>, dim3(grid), dim3(256), 0, (hipStream_t) stream, W);
	else hipLaunchKernelGGL(k_split_match_ops <false>, dim3(grid), dim3(256), 0, (hipStream_t) stream, W);
	return (int) hipGetLastError();
}

}
