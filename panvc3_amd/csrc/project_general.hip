// project_general.hip — k_project_general: the column walk (walk.hpp) for the records the fast kernel left behind.
//
// k_project_fast leaves their start column, projected position and reference-consuming length in the work arrays and their
// numbers in per-wave work lists.  Every wave takes 64 entries of its list at a time, gathers the descriptors lane-parallel
// and then walks the records one by one, with the loads of the next record in flight.
#include <hip/hip_runtime.h>

#include "walk.hpp"

namespace pvc3 {

namespace {

__global__ __launch_bounds__(256) void k_project_general(dev_entry const *__restrict__ entries, uint32_t n_entries, dev_batch_in const in, dev_batch_work const W)
{
	uint32_t const lane = lane_id();
	uint32_t const wave = rfl((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	uint32_t const n_waves = (gridDim.x * blockDim.x) >> 6;
	uint64_t const n_groups = (in.n + 63) / 64;
	uint64_t const g_begin = n_groups * wave / n_waves;
	arena_alloc arena{0, 0};
	uint32_t *run_list = W.run_list + 64 * g_begin;		// work list for k_realign, after the entries of k_project_mid
	uint32_t n_run = rfl(W.run_count[wave]);
	group_out G;
	G.n_failed = 0; G.n_with_runs = 0; G.bitvec_bytes = 0; G.ref_bytes = 0; G.bytes = 0;

	// The records to walk come as this wave's work list: what k_project_mid left of the list k_project_fast wrote (all of it
	// when k_project_mid is switched off).  Scanning the flags of all records instead cost 0.1 ms per 40 M records.
	uint32_t const *list = W.slow_list + 64 * g_begin;
	uint32_t const count = rfl(W.slow_count[wave]) & ~SLOW_ITEMS;		// (the flag is there when k_project_mid did not run)
	for (uint32_t base = 0; base < count; base += 64)
	{
		bool const mine = base + lane < count;
		uint64_t const rec = mine ? list[base + lane] : 0;
		uint64_t todo = ballot(mine);
		if (!todo)
			continue;

		lane_state A;
		A.status = PVC3_ST_OK; A.col = 0; A.r0 = 0; A.ref_consumed = 0; A.n_cigar = 0; A.l_seq = 0; A.src_e = 0; A.dst_e = 0; A.src_pos = 0;
		A.cig_begin = 0; A.seq_begin = 0;
		if (mine)
		{
			A.src_pos = in.src_pos[rec];
			A.src_e = in.src_entry[rec];
			A.dst_e = in.dst_entry[rec];
			A.cig_begin = in.cigar_off[rec];
			A.n_cigar = (uint32_t) (in.cigar_off[rec + 1] - A.cig_begin);
			A.seq_begin = in.seq_off[rec];
			A.l_seq = in.l_seq[rec];
			uint64_t const stash = W.cig_loc[rec];		// left by k_project_fast: start column | reference-consuming length << 32
			A.col = (uint32_t) stash;
			A.ref_consumed = (uint32_t) (stash >> 32);
			A.r0 = W.dst_pos[rec];
			// What k_project_fast's position window left open of the start (all of it when the pair of entries has no windows).
			uint32_t const skip = W.cig_n[rec];
			if (skip && A.src_e < n_entries && A.dst_e < n_entries)
			{
				if (resolve_start(entries[A.src_e], entries[A.dst_e], A.dst_e, A.src_pos, A.col, A.r0, skip))
				{
					W.dst_pos[rec] = A.r0;
					// The neighbouring record may be this record's mate, waiting for this position (k_project_fast).
					uint64_t const nb = rec ^ 1;
					if (nb < in.n && MATE_FROM_NEIGHBOUR == W.mate_dst_pos[nb]) W.mate_dst_pos[nb] = A.r0;
					W.cig_n[rec] = 0;
				}
			}
		}

		G.flags = 0; G.cig_n = 0; G.loc_lo = 0; G.loc_hi = 0;
		rec_ctx C0, C1;
		rec_loads L0, L1;
		rec_out R;
		uint32_t i0 = ctz64(todo), i1 = 0;
		todo &= todo - 1;
		stage_load(entries, in, A, i0, lane, C0, L0);
		for (;;)
		{
			bool const has1 = (0 != todo);
			if (has1) { i1 = ctz64(todo); todo &= todo - 1; stage_load(entries, in, A, i1, lane, C1, L1); }
			walk_record(entries, C0, L0, W, arena, lane, R);
			G.put(i0, R);
			if (!has1) break;
			bool const has0 = (0 != todo);
			if (has0) { i0 = ctz64(todo); todo &= todo - 1; stage_load(entries, in, A, i0, lane, C0, L0); }
			walk_record(entries, C1, L1, W, arena, lane, R);
			G.put(i1, R);
			if (!has0) break;
		}

		{
			uint64_t const m = ballot(mine && (0 == (G.flags & REC_STATUS_MASK)) && (0 != (G.flags & REC_HAS_RUN)));
			if ((m >> lane) & 1) run_list[n_run + mbcnt64(m)] = (uint32_t) rec;
			n_run += popc64(m);
		}
		if (mine)
		{
			bool const ok = (0 == (G.flags & REC_STATUS_MASK));
			if (!ok) W.dst_pos[rec] = 0;		// a rejected record has empty outputs
			W.rec_flags[rec] = G.flags & REC_STATUS_MASK;		// from here on this array is the status output (see dev_batch_packed::status)
			W.cig_n[rec] = G.cig_n;
			W.cig_loc[rec] = ((uint64_t) G.loc_hi << 32) | G.loc_lo;
		}
	}
	W.run_count[wave] = n_run;		// wave-uniform store
	if (0 == lane)
	{
		unsigned long long *ws = W.wave_stats + ((uint64_t) 2 * n_waves + wave) * WAVE_STATS;
		ws[WS_BITVEC] = G.bitvec_bytes;
		ws[WS_REF] = G.ref_bytes;
		ws[WS_FAILED] = G.n_failed;
		ws[WS_BYTES] = G.bytes;
		ws[WS_WITH_RUNS] = G.n_with_runs;
	}
}

}	// anonymous namespace

int launch_project_general(dev_entry const *entries, uint32_t n_entries, dev_batch_in const &in, dev_batch_work const &work, void *stream)
{
	if (0 == in.n) return 0;
	hipLaunchKernelGGL(k_project_general, dim3(project_wave_count(in.n) / 4), dim3(256), 0, (hipStream_t) stream, entries, n_entries, in, work);
	return (int) hipGetLastError();
}

}
