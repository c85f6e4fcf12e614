// realign.hip — local realignment of indel runs (the rare, compute-heavy leg of the hot path).
//
// One wavefront per flagged record.  Replays indel_run_checker (libpanvc3/indel_run_checker.cc:31-121) over the rewritten
// CIGAR, realigns every run that has both insertions and deletions with a global affine alignment of the run's reference
// slice against its query slice (include/panvc3/align.hh:59-112; seqan3::align_pairwise restated, see oracle/pvc3_oracle.c
// for the recurrence and the traceback priority), splices the result (libpanvc3/alignment_projector.cc:58-148) and
// merges adjacent equal operations (collapse_cigar_operations, libpanvc3/cigar.cc:13-51).
//
// DP: anti-diagonal wavefront, one lane per query row in stripes of 64 rows, fp64 adds / compares only (score tables are
// computed on the host with libm), one trace byte per cell in LDS (or in the wave's global scratch slot for large runs).
#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"
#include "device_common.hpp"

namespace pvc3 {

namespace {

constexpr uint32_t REALIGN_LDS_BYTES = 4096;		// per wave (= per block); larger alignments (beyond about 55 x 55) use the wave's global scratch slot
constexpr uint32_t TR_DIAG = 0, TR_UP = 1, TR_LEFT = 2, TR_DIR_MASK = 3, TR_V_OPEN = 4, TR_H_OPEN = 8, TR_MATCH = 16;
constexpr uint32_t MAX_PHRED = 41;

// seqan3::dna5 ranks (A C G N T); assign_char is case-insensitive, U -> T, anything else -> N.
__device__ __forceinline__ uint32_t dna5_rank_of_char(uint32_t c)
{
	c &= ~0x20u;	// upper case for letters
	return ('A' == c) ? 0 : ('C' == c) ? 1 : ('G' == c) ? 2 : ('T' == c || 'U' == c) ? 4 : 3;
}

__device__ __forceinline__ uint32_t dna5_rank_of_seq4(uint32_t code)
{
	return (1 == code) ? 0 : (2 == code) ? 1 : (4 == code) ? 2 : (8 == code) ? 4 : 3;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t lane)
{
	return ((uint64_t) readlane((uint32_t) (v >> 32), lane) << 32) | readlane((uint32_t) v, lane);
}

__device__ __forceinline__ uint32_t seq4_at(uint8_t const *seq4, uint32_t i)
{
	uint32_t const b = seq4[i >> 1];
	return (i & 1) ? (b & 0xf) : (b >> 4);
}

// Lane i receives lane i - 1's value (lane 0 keeps its own): DPP wave_shr:1, no LDS round trip.
__device__ __forceinline__ double shfl_up_f64(double v)
{
	union { double d; int i[2]; } a, r;
	a.d = v;
	r.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], 0x138, 0xf, 0xf, false);
	r.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], 0x138, 0xf, 0xf, false);
	return r.d;
}

struct dp_memory
{
	double *col_m;		// M(i, 0), i = 0..n
	double *bnd_m;		// M(i0, j), j = 0..m   (row above the current stripe)
	double *bnd_v;		// V(i0, j)
	uint8_t *trace;		// n * m bytes
	uint8_t *rev;		// m + n bytes
	uint8_t *ref_rank;	// m bytes: dna5 ranks of the reference slice
};

__device__ __forceinline__ uint64_t dp_bytes(uint32_t m, uint32_t n)
{
	return 8ull * (n + 1) + 16ull * (m + 1) + (uint64_t) m * n + (m + n) + m + 8;
}

__device__ __forceinline__ dp_memory dp_carve(uint8_t *base, uint32_t m, uint32_t n)
{
	dp_memory d;
	d.col_m = (double *) base;
	d.bnd_m = d.col_m + (n + 1);
	d.bnd_v = d.bnd_m + (m + 1);
	d.trace = (uint8_t *) (d.bnd_v + (m + 1));
	d.rev = d.trace + (uint64_t) m * n;
	d.ref_rank = d.rev + (m + n);
	return d;
}

// Global affine alignment of ref[0, m) (horizontal) against query bases [q0, q0 + n) (vertical); pushes the resulting
// operations into `out`.  All control flow is wave-uniform.
struct dp_scores
{
	double open, ext;			// pvc3_params gap costs
	double const *match;		// LDS copies of dev_params::s_match / s_mismatch
	double const *mismatch;
};

__device__ __forceinline__ void align_global(
	uint8_t const *ref, uint32_t m, uint8_t const *seq4, uint32_t q0, uint32_t n, uint8_t const *qual /* or nullptr */,
	dp_scores const &sc, dp_memory const mem, sink &out, uint32_t lane
)
{
	double const go = sc.open + sc.ext;	// seqan3's internal gap-open score = open + extension
	double const ge = sc.ext;
	double const neg_inf = -__builtin_inf();

	// Query base and scores of a lane's row in the stripe starting at i0.  Called for the first stripe before the reference
	// slice is fetched, so that the two global round trips overlap.
	auto load_row = [&](uint32_t i0, uint32_t &q_rank, double &s_match, double &s_mismatch) {
		q_rank = 3; s_match = 0.0; s_mismatch = -1.0;			// nucleotide_scoring_scheme<> default: 0 / -1
		if (i0 + lane < n)
		{
			q_rank = dna5_rank_of_seq4(seq4_at(seq4, q0 + i0 + lane));
			if (qual)
			{
				uint32_t qq = qual[q0 + i0 + lane];
				qq = (qq > MAX_PHRED) ? MAX_PHRED : qq;			// phred42 saturates
				s_match = sc.match[qq];
				s_mismatch = sc.mismatch[qq];
			}
		}
	};
	uint32_t q_rank0; double s_match0, s_mismatch0;
	load_row(0, q_rank0, s_match0, s_mismatch0);

	// Reference slice -> dna5 ranks, once (a global load per DP step would put an HBM round trip on the critical path).
	for (uint32_t j = lane; j < m; j += 64)
		mem.ref_rank[j] = (uint8_t) dna5_rank_of_char(ref[j]);

	// First column and first row (initialise_first_column_cell / initialise_first_row_cell).  The recurrence
	//     x_1 = go,  x_k = max(x_{k-1} + go [open], x_{k-1} + ge [extension, kept on ties])
	// has the closed forms k * (open + ext) when the opening score is positive and open + k * ext otherwise; all values are
	// small integers, so the products are exactly the iterated sums of the oracle.
	{
		double const open = sc.open;
		for (uint32_t k = lane; k <= (m > n ? m : n); k += 64)
		{
			double const x = (0 == k) ? 0.0 : ((open > 0.0) ? (double) k * go : open + (double) k * ge);
			if (k <= n) mem.col_m[k] = x;
			if (k <= m) { mem.bnd_m[k] = x; mem.bnd_v[k] = neg_inf; }
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

	for (uint32_t i0 = 0; i0 < n; i0 += 64)
	{
		uint32_t const nr = (n - i0 < 64) ? (n - i0) : 64;		// rows in this stripe
		bool const row_ok = lane < nr;
		uint32_t const i = i0 + lane + 1;						// this lane's row (1-based)
		uint32_t q_rank = q_rank0;
		double s_match = s_match0, s_mismatch = s_mismatch0;
		if (i0) load_row(i0, q_rank, s_match, s_mismatch);
		double m_left = 0.0, h_left = neg_inf, diag = 0.0;
		if (row_ok)
		{
			m_left = mem.col_m[i];
			diag = mem.col_m[i - 1];
		}
		double cur_m = 0.0, cur_v = neg_inf;
		uint32_t const steps = m + nr - 1;
		for (uint32_t t = 0; t < steps; ++t)
		{
			double up_m = shfl_up_f64(cur_m);
			double up_v = shfl_up_f64(cur_v);
			uint32_t const j = t - lane + 1;					// column handled by this lane in this step
			bool const active = row_ok && (t >= lane) && (j <= m);
			if (active)
			{
				if (0 == lane) { up_m = mem.bnd_m[j]; up_v = mem.bnd_v[j]; }
				uint32_t const r_rank = mem.ref_rank[j - 1];
				double const s = (r_rank == q_rank) ? s_match : s_mismatch;
				uint32_t tr = 0;
				double const v_open = up_m + go, v_ext = up_v + ge;
				double v;
				if (v_ext < v_open) { v = v_open; tr |= TR_V_OPEN; } else v = v_ext;
				double const h_open = m_left + go, h_ext = h_left + ge;
				double h;
				if (h_ext < h_open) { h = h_open; tr |= TR_H_OPEN; } else h = h_ext;
				double best = diag + s;
				uint32_t dir = TR_DIAG;
				if (best < v) { best = v; dir = TR_UP; }
				if (best < h) { best = h; dir = TR_LEFT; }
				mem.trace[(uint64_t) (i - 1) * m + (j - 1)] = (uint8_t) (tr | dir | ((r_rank == q_rank) ? TR_MATCH : 0));
				m_left = best; h_left = h; cur_m = best; cur_v = v; diag = up_m;
				if (lane == nr - 1) { mem.bnd_m[j] = best; mem.bnd_v[j] = v; }
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}

	// Traceback (trace_iterator_base): diagonal > up > left where the direction is decided; a gap continues until the
	// cell whose gap score was formed by opening.
	uint32_t k = 0;
	{
		uint32_t i = n, j = m;
		uint32_t st = 0;	// 0 = M, 1 = V (up), 2 = H (left)
		while (i || j)
		{
			uint32_t t;
			if (0 == i) t = TR_LEFT | ((1 == j) ? TR_H_OPEN : 0);		// first row: left, opened at column 1
			else if (0 == j) t = TR_UP | ((1 == i) ? TR_V_OPEN : 0);	// first column
			else t = mem.trace[(uint64_t) (i - 1) * m + (j - 1)];
			t = rfl(t);
			if (0 == st)
			{
				uint32_t const dir = t & TR_DIR_MASK;
				if (TR_DIAG == dir)
				{
					mem.rev[k] = (uint8_t) ((t & TR_MATCH) ? OP_EQ : OP_X);		// align.hh:103-106; wave-uniform stores
					++k; --i; --j;
					continue;
				}
				st = (TR_UP == dir) ? 1 : 2;
			}
			if (1 == st)
			{
				mem.rev[k] = (uint8_t) OP_I;
				++k;
				if (t & TR_V_OPEN) st = 0;
				--i;
			}
			else
			{
				mem.rev[k] = (uint8_t) OP_D;
				++k;
				if (t & TR_H_OPEN) st = 0;
				--j;
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	for (uint32_t x = k; x > 0; --x)
		out.push(rfl(mem.rev[x - 1]), 1, lane);		// align.hh:94-108 through cigar_buffer
}

// One record at a time per wave is bound by the latency of that record's dependent steps, so the kernel wants as many
// resident waves as the CU has slots: 64 VGPRs (a few spills) and 4 KB of LDS give 8 per SIMD; on C5 that is 17.6 ms
// against 22.5 ms with 86 VGPRs / 8 KB / 5 waves per SIMD.
__global__ __launch_bounds__(64, 8) void k_realign(
	dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W,
	dev_params const *__restrict__ prm, uint32_t lds_budget, uint32_t n_regions
)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[REALIGN_LDS_BYTES];
	__shared__ double lds_scores[2 * (MAX_PHRED + 1)];
	uint32_t const lane = lane_id();
	if (lane <= MAX_PHRED) { lds_scores[lane] = prm->s_match[lane]; lds_scores[MAX_PHRED + 1 + lane] = prm->s_mismatch[lane]; }
	dp_scores sc;
	sc.open = (double) prm->gap_open; sc.ext = (double) prm->gap_ext;
	sc.match = lds_scores; sc.mismatch = lds_scores + (MAX_PHRED + 1);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	uint32_t const wave = blockIdx.x;
	uint32_t const n_waves = gridDim.x;
	arena_alloc arena{0, 0}, runs_alloc{0, 0};
	uint32_t total_runs = 0, n_failed = 0, touched = 0;
	uint64_t qual_bytes = 0;

	// The records with indel runs come as per-region work lists (region = the record range of one wave of the projection
	// kernels, filled by k_project_mid and k_project_general); every wave takes a contiguous range of regions and walks each
	// list 64 entries at a time.  (Scanning the flags of all records instead cost more than half of this kernel on C3.)
	uint64_t const n_groups = (in.n + 63) / 64;
	uint32_t const r_begin = (uint32_t) ((uint64_t) n_regions * wave / n_waves), r_end = (uint32_t) ((uint64_t) n_regions * (wave + 1) / n_waves);
	for (uint32_t region = r_begin; region < r_end; ++region)
	for (uint32_t chunk = 0, cnt = rfl(W.run_count[region]); chunk < cnt; chunk += 64)
	{
		uint32_t const *list = W.run_list + 64 * (n_groups * region / n_regions);
		bool const mine = chunk + lane < cnt;
		uint32_t const v_rec = mine ? list[chunk + lane] : 0;
		uint64_t todo = ballot(mine);

		// Descriptors of the marked records, lane-parallel.
		uint64_t v_loc = 0, v_seq_off = 0, v_qual_off = 0, v_text = 0;
		uint32_t v_n = 0, v_dst_pos = 0, v_text_len = 0, v_l_seq = 0, v_has_qual = 0;
		if (mine)
		{
			uint64_t const r = v_rec;
			v_loc = W.cig_loc[r]; v_n = W.cig_n[r]; v_dst_pos = W.dst_pos[r];
			dev_entry const *de = entries + in.dst_entry[r];
			v_text = (uint64_t) de->text; v_text_len = de->text_len;
			v_seq_off = in.seq_off[r]; v_l_seq = in.l_seq[r];
			v_has_qual = ((in.flags[r] & PVC3_FLAG_HAS_QUAL) && in.qual) ? 1 : 0;
			if (v_has_qual) v_qual_off = in.qual_off[r];
		}

		// Touch what the records of this chunk will read (operations, text around the projected position, SEQ, QUAL), one
		// record per lane, all at once: the records are then walked one after the other with warm caches and TLBs instead of
		// paying a full miss at every dependent step.  The values only keep the loads alive.
		if (mine)
		{
			uint32_t const *o = W.arena + v_loc;
			touched ^= o[0];
			if (v_n > 16) touched ^= o[16];
			if (v_text)
			{
				uint8_t const *t = (uint8_t const *) v_text + v_dst_pos;		// texts carry TEXT_PAD readable bytes past their end
				touched ^= (uint32_t) t[0] ^ (uint32_t) t[64] ^ (uint32_t) t[128] ^ (uint32_t) t[192];
			}
			uint8_t const *q = in.seq4 + v_seq_off;
			touched ^= (uint32_t) q[0] ^ (uint32_t) q[(v_l_seq + 1) / 2 - 1];
			if (v_has_qual)
			{
				uint8_t const *qq = in.qual + v_qual_off;
				touched ^= (uint32_t) qq[0] ^ (uint32_t) qq[v_l_seq / 2] ^ (uint32_t) qq[v_l_seq - 1];
			}
		}

		// The first 64 operations of a record live in lanes; those of the next marked record are fetched while this one is processed.
		auto load_ops = [&](uint32_t r) -> uint32_t { return (lane < readlane(v_n, r)) ? (W.arena + readlane64(v_loc, r))[lane] : 0; };
		uint32_t opv_next = load_ops(ctz64(todo));
		while (todo)
		{
			uint32_t const ri = ctz64(todo);
			uint64_t const rec = readlane(v_rec, ri);
			todo &= todo - 1;
			uint32_t const opv = opv_next;
			if (todo) opv_next = load_ops(ctz64(todo));

			uint32_t const *ops = W.arena + readlane64(v_loc, ri);
			uint32_t const n_ops = readlane(v_n, ri);
			uint32_t const dst_pos = readlane(v_dst_pos, ri);
			uint8_t const *d_text = (uint8_t const *) readlane64(v_text, ri);
			uint32_t const d_text_len = readlane(v_text_len, ri);
			uint8_t const *seq4 = in.seq4 + readlane64(v_seq_off, ri);
			uint32_t const l_seq = readlane(v_l_seq, ri);
			uint8_t const *qual = readlane(v_has_qual, ri) ? (in.qual + readlane64(v_qual_off, ri)) : nullptr;

			// Upper bounds: every run of total length m + n yields at most m + n operations.
			uint32_t indel_total = 0, n_indel_ops = 0;
			for (uint32_t x = lane; x < n_ops; x += 64)
			{
				uint32_t const c = (x < 64) ? opv : ops[x];
				uint32_t const op = c & 0xf;
				if (OP_I == op || OP_D == op) { indel_total += c >> 4; ++n_indel_ops; }
			}
			for (int off = 32; off; off >>= 1)
			{
				indel_total += __shfl_xor(indel_total, off);
				n_indel_ops += __shfl_xor(n_indel_ops, off);
			}
			indel_total = rfl(indel_total); n_indel_ops = rfl(n_indel_ops);

			uint64_t out_loc = 0, runs_loc = 0;
			bool ok = arena_take(arena, n_ops + indel_total + 1, W, lane, out_loc);
			ok = ok && runs_take(runs_alloc, n_indel_ops / 2 + 1, W, lane, runs_loc);
			if (!ok)
				continue;	// the overflow flag is set; the host grows the arenas and repeats the batch

			sink out;
			out.init();
			out.spill = W.arena + out_loc;
			pvc3_run *runs = W.runs_arena + runs_loc;
			uint32_t n_runs = 0;
			uint32_t status = PVC3_ST_OK;

			// indel_run_checker state (indel_run_checker.cc:13-21)
			uint32_t ref_pos = dst_pos, query_pos = 0, run_type = 0;
			uint32_t run_first = 0, run_ref_loc = dst_pos, run_query_loc = 0;

			for (uint32_t x = 0; x <= n_ops && !status; ++x)
			{
				bool const at_end = (x == n_ops);
				uint32_t op = 0xf, cnt = 0;
				if (!at_end)
				{
					uint32_t const c = (x < 64) ? readlane(opv, x) : rfl(ops[x]);
					op = c & 0xf; cnt = c >> 4;
					if (0 == run_type) { run_ref_loc = ref_pos; run_query_loc = query_pos; run_first = x; }	// :45-50
					if (OP_I == op) { query_pos += cnt; run_type |= 1; continue; }							// :62-65
					if (OP_D == op) { ref_pos += cnt; run_type |= 2; continue; }							// :67-70
				}
				// A run-terminating operation (:55-60, :72-94) or the end of the CIGAR (:111-118).
				uint32_t const prev_type = run_type;
				uint32_t const run_ref_end = ref_pos, run_query_end = query_pos;
				run_type = 0;
				if (!at_end)
				{
					if (OP_S == op) query_pos += cnt;
					else if (OP_N == op) ref_pos += cnt;
					else if (OP_M == op || OP_EQ == op || OP_X == op) { ref_pos += cnt; query_pos += cnt; }
				}
				if (3 == prev_type)
				{
					uint32_t const m = run_ref_end - run_ref_loc, nq = run_query_end - run_query_loc;
					// alignment_projector.cc:69-73: reference_position() already includes the terminating operation.
					{
						pvc3_run rr;		// wave-uniform store
						rr.query_location = run_query_loc; rr.query_length = nq; rr.ref_position = ref_pos; rr.ref_length = m;
						runs[n_runs] = rr;
					}
					++n_runs;
					if (!d_text || (uint64_t) run_ref_loc + m > d_text_len) { status = PVC3_ST_REF_RANGE; break; }
					if ((uint64_t) run_query_loc + nq > l_seq) { status = PVC3_ST_QUERY_RANGE; break; }

					uint64_t const need = dp_bytes(m, nq);
					if (need <= lds_budget)
					{
						// Separate call site: after inlining every access is provably to LDS (ds_read / ds_write instead of flat_*).
						align_global(d_text + run_ref_loc, m, seq4, run_query_loc, nq, qual, sc, dp_carve(lds, m, nq), out, lane);
					}
					else
					{
						if (need > W.scratch_per_wave)
						{
							atomicMax(&W.counters->scratch_needed, (unsigned int) ((need > 0xFFFFFFFFull) ? 0xFFFFFFFFu : need));	// idempotent
							W.counters->scratch_overflow = 1;
							status = 0xff;	// not a record status: the host grows the scratch and repeats
							break;
						}
						align_global(d_text + run_ref_loc, m, seq4, run_query_loc, nq, qual, sc, dp_carve(W.scratch + (uint64_t) wave * W.scratch_per_wave, m, nq), out, lane);
					}
				}
				else if (prev_type)
				{
					for (uint32_t y = run_first; y < x; ++y)	// single-type run: copied as is (alignment_projector.cc:61-65)
					{
						uint32_t const c = (y < 64) ? readlane(opv, y) : rfl(ops[y]);
						out.push(c & 0xf, c >> 4, lane);
					}
				}
				if (!at_end)
					out.push(op, cnt, lane);
			}
			out.finish(lane);

			if (0xff == status)
				continue;
			// Wave-uniform stores (every lane writes the same values).
			if (status)
			{
				W.cig_n[rec] = 0; W.run_n[rec] = 0; W.dst_pos[rec] = 0;
				W.rec_flags[rec] = status;
				++n_failed;
			}
			else
			{
				W.cig_loc[rec] = out_loc; W.cig_n[rec] = out.n_out;
				W.run_loc[rec] = runs_loc; W.run_n[rec] = n_runs;
			}
			if (!status) { total_runs += n_runs; qual_bytes += qual ? l_seq : 0; }
		}
	}
	if (0xA5A5A5A5u == touched && 0xFFFFFFFFu == n_failed)		// never true: keeps the touch loads
		W.counters->scratch_needed = touched;
	if (0 == lane && n_failed)
		atomicAdd(&W.counters->n_failed, (unsigned long long) n_failed);
	if (0 == lane && total_runs)
	{
		atomicAdd(&W.counters->n_runs, (unsigned long long) total_runs);
		atomicAdd(&W.counters->qual_bytes, (unsigned long long) qual_bytes);
	}
}

}	// anonymous namespace

uint32_t realign_lds_bytes() { return REALIGN_LDS_BYTES; }

int launch_realign(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, dev_params const *params, uint32_t n_waves, uint32_t lds_budget, void *stream)
{
	if (0 == in.n) return 0;
	if (lds_budget > REALIGN_LDS_BYTES) lds_budget = REALIGN_LDS_BYTES;
	uint32_t const n_regions = project_wave_count(in.n);		// one work list per wave of the projection kernels
	if (n_waves > n_regions) n_waves = n_regions;
	hipLaunchKernelGGL(k_realign, dim3(n_waves), dim3(64), 0, (hipStream_t) stream, entries, n_entries, in, work, params, lds_budget, n_regions);
	return (int) hipGetLastError();
}

}
