// mapq.hip — recalculate_mapq on the GPU (the step after project_alignments in both workflows).
//
//   k_rescore_alignments   lane = record: alignment score from the CIGAR (= / X form), the base qualities and the XN bit map
//                          of N positions in the reference (recalculate-mapq/main.cc:189-289, cigar_alignment_scorer)
//   k_mapq_groups          lane = QNAME group: pairwise scores through the original mate positions (XR / XP / YR / YP written by
//                          project_alignments), best total per projected position pair, the next best total at another
//                          position pair, and the Bowtie 2 (version 2) MAPQ table (main.cc:441-566, 962-1180)
//
// Groups are small (bowtie2 -k founders + 2), so a lane compares the records of its group pairwise instead of sorting them;
// the results are those of the reference's sort / binary-search formulation (see oracle/recalculate_mapq.py).
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

constexpr double SCORE_MIN = -1.7976931348623157e308;		// -DBL_MAX: ALIGNMENT_SCORE_MIN (main.cc:50)
constexpr uint32_t NO_POS = 0xFFFFFFFFu;

// std::min(i, 40) / 40.0 * (max - min) + min converted to int32 (main.cc:182-186)
__device__ __forceinline__ int32_t penalty_of(uint32_t rank, int32_t lo, int32_t hi)
{
	uint32_t const r = (rank < 40) ? rank : 40;
	return (int32_t) ((double) r / 40.0 * (double) (hi - lo) + (double) lo);
}

__global__ __launch_bounds__(256) void k_rescore_alignments(
	uint64_t n, uint64_t const *__restrict__ cigar_off, uint32_t const *__restrict__ cigar, uint64_t const *__restrict__ seq_off, uint8_t const *__restrict__ seq4,
	uint32_t const *__restrict__ l_seq, uint64_t const *__restrict__ qual_off, uint8_t const *__restrict__ qual, uint64_t const *__restrict__ xn_off, uint8_t const *__restrict__ xn,
	pvc3_rescore_params const prm, double *__restrict__ score, int32_t *__restrict__ status
)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t rec = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; rec < n; rec += stride)
	{
		uint32_t const *ops = cigar + cigar_off[rec];
		uint32_t const n_ops = (uint32_t) (cigar_off[rec + 1] - cigar_off[rec]);
		uint8_t const *s4 = seq4 + seq_off[rec];
		uint32_t const len = l_seq[rec];
		bool const has_qual = qual && (qual_off[rec + 1] - qual_off[rec] >= len);
		uint8_t const *q = has_qual ? (qual + qual_off[rec]) : nullptr;
		uint8_t const *bits = xn + xn_off[rec];
		uint64_t const n_bits_bytes = xn_off[rec + 1] - xn_off[rec];
		uint32_t query_pos = 0, ref_pos = 0, prev_indel = 0;
		int32_t st = PVC3_RESCORE_OK;
		double retval = 0.0;
		auto base_is_n = [&](uint32_t i) -> bool {
			uint32_t const b = s4[i >> 1], code = (i & 1) ? (b & 0xf) : (b >> 4);
			return !(1 == code || 2 == code || 4 == code || 8 == code);		// everything but A C G T reads as N (seqan3::dna5)
		};
		auto rank_of = [&](uint32_t i) -> uint32_t { uint32_t const v = q ? q[i] : 0; return (v > 41) ? 41 : v; };		// phred42 saturates
		for (uint32_t x = 0; x < n_ops && PVC3_RESCORE_OK == st; ++x)
		{
			uint32_t const c = ops[x], op = c & 0xf, count = c >> 4;
			if (OP_H == op || OP_P == op) continue;							// :213-216: prev_op is not set
			if (OP_N == op) { ref_pos += count; continue; }					// :218-221
			if (OP_I == op || OP_D == op)									// :223-239
			{
				if (!prev_indel) retval -= (double) prm.gap_opening_penalty;
				retval -= (double) count * (double) prm.gap_extension_penalty;
				if (OP_I == op) query_pos += count; else ref_pos += count;
				prev_indel = 1;
				continue;
			}
			prev_indel = 0;
			if (OP_S == op) { query_pos += count; continue; }					// :241-244
			if (OP_EQ == op || OP_X == op)
			{
				if ((uint64_t) query_pos + count > len) { st = PVC3_RESCORE_QUERY_RANGE; break; }
				for (uint32_t i = 0; i < count; ++i)
				{
					uint32_t const qp = query_pos + i, rp = ref_pos + i;
					if (OP_EQ == op)										// :246-255: a match against N costs the N penalty
					{
						if (base_is_n(qp)) retval -= (double) penalty_of(rank_of(qp), prm.n_penalty, prm.n_penalty);
					}
					else													// :257-273
					{
						if (rp / 8 >= n_bits_bytes) { st = PVC3_RESCORE_XN_RANGE; break; }		// the reference's libbio_always_assert_lt
						bool const ref_n = 0 != ((1u << (rp % 8)) & bits[rp / 8]);
						if (base_is_n(qp) || ref_n) retval -= (double) penalty_of(rank_of(qp), prm.n_penalty, prm.n_penalty);
						else retval -= (double) penalty_of(rank_of(qp), prm.min_mismatch_penalty, prm.max_mismatch_penalty);
					}
				}
				query_pos += count; ref_pos += count;
				continue;
			}
			st = PVC3_RESCORE_CIGAR_OP;										// :275-279: M (or an unknown code) is a failure
		}
		score[rec] = (PVC3_RESCORE_OK == st) ? retval : 0.0;
		status[rec] = st;
	}
}


// ---- MAPQ -------------------------------------------------------------------------------------------------------------
struct pos2 { uint32_t chr, pos; };
__device__ __forceinline__ bool operator==(pos2 a, pos2 b) { return a.chr == b.chr && a.pos == b.pos; }
__device__ __forceinline__ bool operator<(pos2 a, pos2 b) { return a.chr < b.chr || (a.chr == b.chr && a.pos < b.pos); }
__device__ __forceinline__ bool is_invalid(pos2 a) { return NO_POS == a.chr && NO_POS == a.pos; }
__device__ __forceinline__ pos2 make_pos(uint32_t chr, uint32_t pos) { return (NO_POS == chr || NO_POS == pos) ? pos2{NO_POS, NO_POS} : pos2{chr, pos}; }

__device__ __forceinline__ double read_min_score(uint32_t len) { return len ? (-0.6 + (-0.6 * (double) len)) : 0.0; }		// main.cc:455-460

// bowtie2_v2_score_calculator::calculate_mapq (main.cc:479-566) with its tables (:375-431)
__device__ uint32_t calculate_mapq(uint32_t read_length, uint32_t other_read_length, double score, double next_score_)
{
	double const min_score = read_min_score(read_length) + read_min_score(other_read_length);
	double const max_score = 0.0;
	double const range0 = max_score - min_score;
	double const score_range = (1.0 > range0) ? 1.0 : range0;
	if (score < min_score) return 0;
	double const next_score = (min_score < next_score_) ? next_score_ : min_score;
	double const nsq = (score - min_score) / score_range;
	double const diff_next = score - next_score;
	double const dnq = diff_next / score_range;
	if (SCORE_MIN == next_score)		// not reachable (next_score >= min_score), kept as in the source
	{
		double const th[7] = {0.0, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8};
		uint32_t const mq[7] = {0, 3, 8, 23, 24, 40, 42};
		uint32_t i = 0;
		while (i < 7 && !(nsq < th[i])) ++i;
		return mq[i ? i - 1 : 0];
	}
	if (0.0 != diff_next)
	{
		// non_unique_alignment_scores: (diff_next_threshold, normalised_score_threshold, mapq), sorted
		double const d[30] = {0.0, 0.0, 0.1, 0.1, 0.1, 0.1, 0.2, 0.2, 0.2, 0.2, 0.3, 0.3, 0.3, 0.3, 0.4, 0.4, 0.4, 0.4, 0.5, 0.5, 0.5, 0.5, 0.6, 0.6, 0.7, 0.7, 0.8, 0.8, 0.9, 0.9};
		double const s[30] = {0.0, 0.67, 0.0, 0.67, 0.88, 1.0, 0.0, 0.67, 0.88, 1.0, 0.0, 0.67, 0.88, 1.0, 0.0, 0.68, 0.84, 1.0, 0.0, 0.68, 0.84, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0};
		uint32_t const mq[30] = {2, 6, 0, 7, 12, 30, 0, 11, 17, 31, 3, 15, 18, 32, 4, 14, 21, 34, 5, 16, 25, 35, 22, 36, 26, 37, 27, 38, 33, 39};
		uint32_t it = 0;
		while (it < 30 && !(dnq < d[it])) ++it;			// upper_bound on diff_next_threshold
		double const threshold = d[it ? it - 1 : 0];
		uint32_t j = 0;
		while (j < it && !((threshold < d[j]) || (threshold == d[j] && nsq < s[j]))) ++j;		// upper_bound on (threshold, normalised score)
		return mq[j ? j - 1 : 0];
	}
	return (nsq >= 0.67) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_mapq_groups(pvc3_mapq_batch const in, double *__restrict__ pw, double *__restrict__ best_total, uint32_t *__restrict__ mate_len, uint8_t *__restrict__ mapq, uint8_t *__restrict__ emitted, unsigned long long *__restrict__ stats)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	unsigned long long st_unpaired = 0, st_both = 0, st_mate_missing = 0, st_no_position = 0;
	for (uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; g < in.n_groups; g += stride)
	{
		uint64_t const r0 = in.group_off[g], r1 = in.group_off[g + 1];
		if (r0 == r1) { emitted[g] = 0; continue; }
		auto pos_of = [&](uint64_t r) { return make_pos(in.ref_id[r], in.pos[r]); };
		auto mate_of = [&](uint64_t r) { return make_pos(in.mate_ref_id[r], in.mate_pos[r]); };
		auto orig_of = [&](uint64_t r) { return make_pos(in.orig_ref_id[r], in.orig_pos[r]); };
		// :1010-1044: record types of the group
		uint32_t seen = 0;
		bool any_valid_original = false;
		for (uint64_t r = r0; r < r1; ++r)
		{
			bool const has_mate = (NO_POS != in.mate_ref_id[r]) && (NO_POS != in.mate_pos[r]);
			seen |= 1u << (has_mate ? 1 : 0);
			st_unpaired += !has_mate;
			any_valid_original = any_valid_original || !is_invalid(orig_of(r));
		}
		if (3 == seen)
		{
			++st_both;
			emitted[g] = 0;
			for (uint64_t r = r0; r < r1; ++r) mapq[r] = 0;
			continue;
		}
		emitted[g] = 1;
		st_no_position += !any_valid_original;
		// :1066-1118: pairwise scores through the original positions
		for (uint64_t r = r0; r < r1; ++r)
		{
			pos2 a = pos_of(r), b = mate_of(r);
			if (!(a < b)) { pos2 const t = a; a = b; b = t; }		// position_pair::normalise
			bool const has_mate = !is_invalid(b);
			double other = 0.0;
			uint32_t mate_length = 0;
			if (!is_invalid(a) && has_mate)
			{
				pos2 const mo = make_pos(in.orig_mate_ref_id[r], in.orig_mate_pos[r]);
				// The reference sorts (original position, score, length) and takes the last entry at the mate's position: the best
				// score there, the later record on ties; its sentinel (invalid position, score 0, length 0) comes last of all and
				// takes part when the tags are missing.
				bool found = false;
				double best = 0.0;
				for (uint64_t s = r0; s < r1; ++s)
					if (orig_of(s) == mo && (!found || !(in.score[s] < best))) { best = in.score[s]; mate_length = in.l_seq[s]; found = true; }
				if (is_invalid(mo) && (!found || !(0.0 < best))) { best = 0.0; mate_length = 0; found = true; }
				if (found) other = best; else ++st_mate_missing;
			}
			mate_len[r] = mate_length;
			pw[r] = in.score[r] + other;
		}
		bool const paired = (2 == seen);
		// add_paired_segment_score (:893-917): one entry per position pair with the best total (paired records), one per record otherwise
		for (uint64_t r = r0; r < r1; ++r)
		{
			double bt = pw[r];
			if (paired)
			{
				pos2 a = pos_of(r), b = mate_of(r);
				if (!(a < b)) { pos2 const t = a; a = b; b = t; }
				for (uint64_t s = r0; s < r1; ++s)
				{
					pos2 c = pos_of(s), d = mate_of(s);
					if (!(c < d)) { pos2 const t = c; c = d; d = t; }
					if (c == a && d == b && bt < pw[s]) bt = pw[s];
				}
			}
			best_total[r] = bt;
		}
		// :1122-1170
		for (uint64_t r = r0; r < r1; ++r)
		{
			pos2 a = pos_of(r), b = mate_of(r);
			if (!(a < b)) { pos2 const t = a; a = b; b = t; }
			double const mine = pw[r];
			bool have_next = false, better_exists = false;
			double next = 0.0;
			for (uint64_t s = r0; s < r1; ++s)
			{
				double const total = best_total[s];
				if (mine < total) { better_exists = true; continue; }		// entries above the upper bound are not candidates
				if (!paired && in.seq_class[s] != in.seq_class[r]) continue;	// another sequence under the same name
				pos2 c = pos_of(s), d = mate_of(s);
				if (!(c < d)) { pos2 const t = c; c = d; d = t; }
				if (c == a && d == b) continue;									// the record's own position pair
				if (!have_next || next < total) next = total;
				have_next = true;
			}
			uint32_t const mate_length = mate_len[r];
			uint32_t q;
			if (have_next) q = calculate_mapq(in.l_seq[r], mate_length, mine, next);
			else if (!better_exists) q = calculate_mapq(in.l_seq[r], mate_length, mine, SCORE_MIN);
			else q = 255;		// MAPQ_NO_NEXT_RECORD
			mapq[r] = (uint8_t) q;
		}
	}
	if (st_unpaired) atomicAdd(stats + 0, st_unpaired);
	if (st_both) atomicAdd(stats + 1, st_both);
	if (st_mate_missing) atomicAdd(stats + 2, st_mate_missing);
	if (st_no_position) atomicAdd(stats + 3, st_no_position);
}

}	// anonymous namespace

int launch_rescore(
	uint64_t n, uint64_t const *cigar_off, uint32_t const *cigar, uint64_t const *seq_off, uint8_t const *seq4, uint32_t const *l_seq,
	uint64_t const *qual_off, uint8_t const *qual, uint64_t const *xn_off, uint8_t const *xn, pvc3_rescore_params const &prm, double *score, int32_t *status, void *stream
)
{
	if (n)
	{
		uint64_t const blocks = (n + 255) / 256;
		hipLaunchKernelGGL(k_rescore_alignments, dim3((uint32_t) (blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t) stream,
			n, cigar_off, cigar, seq_off, seq4, l_seq, qual_off, qual, xn_off, xn, prm, score, status);
	}
	return (int) hipGetLastError();
}

int launch_mapq_groups(pvc3_mapq_batch const &in, double *pw, double *best_total, uint32_t *mate_len, uint8_t *mapq, uint8_t *emitted, unsigned long long *stats, void *stream)
{
	if (in.n_groups)
	{
		uint64_t const blocks = (in.n_groups + 255) / 256;
		hipLaunchKernelGGL(k_mapq_groups, dim3((uint32_t) (blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t) stream, in, pw, best_total, mate_len, mapq, emitted, stats);
	}
	return (int) hipGetLastError();
}

}
