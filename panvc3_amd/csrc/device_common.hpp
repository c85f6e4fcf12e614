// device_common.hpp — device-side pieces shared by the projection and realignment kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"
#include "wave.hpp"

namespace pvc3 {

constexpr uint32_t OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8;
constexpr uint32_t ARENA_CHUNK = 2048;			// uint32 per wave-private arena chunk
constexpr uint32_t RUNS_CHUNK = 128;			// pvc3_run per wave-private chunk
constexpr uint32_t SINK_LANES = 192;			// runs buffered in three VGPRs before spilling to the arena
constexpr uint32_t SINK_SPILL_AT = SINK_LANES - 66;
constexpr uint32_t SEQ_WINDOW_BASES = 512;		// 64 lanes x 8 four-bit bases
constexpr uint32_t REF_WINDOW_BYTES = 256;

// ---- wave-private bump allocator over the global CIGAR arena -------------------------------------------------------
struct arena_alloc
{
	uint64_t ptr;
	uint32_t left;
};

// Takes n items; refills the wave's private chunk from the global head when it runs out (one atomic per chunk).
__device__ __forceinline__ bool arena_take_from(
	arena_alloc &a, uint32_t n, uint32_t chunk, unsigned long long *head, uint64_t cap, unsigned int *overflow, uint32_t lane, uint64_t &off
)
{
	a.ptr = rfl64(a.ptr); a.left = rfl(a.left); n = rfl(n);
	if (n <= a.left)
	{
		off = a.ptr; a.ptr += n; a.left -= n;
		return true;
	}
	uint32_t const grab = (n > chunk) ? n : chunk;
	unsigned long long off0 = 0;
	if (0 == lane)
		off0 = atomicAdd(head, (unsigned long long) grab);
	off0 = rfl64(off0);
	if (off0 + grab > cap)
	{
		*overflow = 1;		// every lane stores the same value: a single-lane branch here would make the join look divergent
		a.left = 0;
		off = 0;
		return false;
	}
	off = off0; a.ptr = off0 + n; a.left = grab - n;
	return true;
}

__device__ __forceinline__ bool arena_take(arena_alloc &a, uint32_t n, dev_batch_work const &W, uint32_t lane, uint64_t &off)
{
	return arena_take_from(a, n, ARENA_CHUNK, &W.counters->arena_head, W.arena_cap, &W.counters->arena_overflow, lane, off);
}

__device__ __forceinline__ bool runs_take(arena_alloc &a, uint32_t n, dev_batch_work const &W, uint32_t lane, uint64_t &off)
{
	return arena_take_from(a, n, RUNS_CHUNK, &W.counters->runs_head, W.runs_cap, &W.counters->runs_overflow, lane, off);
}

// ---- cigar_buffer (libpanvc3/cigar.cc:54-95) as wave-uniform state + lane-buffered output ---------------------------
struct sink
{
	uint32_t cur_op, cur_cnt, n_out;	// wave-uniform
	uint32_t run_type, has_run;			// indel_run_checker state: bit 0 = I seen, bit 1 = D seen
	uint32_t buf0, buf1, buf2;			// lane i of buf{0,1,2} = output operation i, 64 + i, 128 + i
	uint32_t *spill;					// once set, operations are stored directly (records with > ~126 runs)

	__device__ __forceinline__ void init()
	{
		cur_op = 0xf; cur_cnt = 0; n_out = 0; run_type = 0; has_run = 0; buf0 = buf1 = buf2 = 0; spill = nullptr;
	}

	__device__ __forceinline__ void make_uniform()
	{
		cur_op = rfl(cur_op); cur_cnt = rfl(cur_cnt); n_out = rfl(n_out); run_type = rfl(run_type); has_run = rfl(has_run);
		spill = rfl_ptr(spill);
	}

	__device__ __forceinline__ void flush(uint32_t lane)
	{
		make_uniform();
		uint32_t const enc = (cur_cnt << 4) | cur_op;
		if (spill)
		{
			spill[n_out] = enc;		// wave-uniform store (same address, same value in every lane)
		}
		else if (n_out < 64) buf0 = writelane(buf0, enc, n_out);
		else if (n_out < 128) buf1 = writelane(buf1, enc, n_out - 64);
		else buf2 = writelane(buf2, enc, n_out - 128);
		++n_out;
		// indel_run_checker.cc:55-94: I / D extend a run, everything else terminates it.
		if (OP_I == cur_op) run_type |= 1;
		else if (OP_D == cur_op) run_type |= 2;
		else { has_run |= (3 == run_type); run_type = 0; }
	}

	// cigar.cc:61-87 (the first push finds cur_cnt == 0 and an impossible cur_op).
	__device__ __forceinline__ void push(uint32_t op, uint32_t cnt, uint32_t lane)
	{
		op = rfl(op); cnt = rfl(cnt);
		if (0 == cnt) return;
		make_uniform();
		if (cur_op == op) cur_cnt += cnt;
		else
		{
			if (cur_cnt) flush(lane);
			cur_op = op; cur_cnt = cnt;
		}
	}

	// cigar.cc:90-95 + the final-run check of indel_run_checker.cc:111-118
	__device__ __forceinline__ void finish(uint32_t lane)
	{
		if (cur_cnt) flush(lane);
		has_run |= (3 == run_type);
	}
};


__device__ __forceinline__ uint32_t load_u32_unaligned(uint8_t const *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

}
