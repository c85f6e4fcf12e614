// ta_cost.hip — what one vector memory instruction costs the texture addresser / L1 of a CU on MI355X, by width, alignment
// and lane pattern, with the data resident in the L1 / L2 (no HBM in the way).  k_project_fast's compare issues twelve
// 16-byte loads per group of 64 records at byte granularity; the counters showed the addressers 70 % busy.
//
// Every wave of a full grid (256 CUs x 16 waves) runs ITER iterations of 8 independent loads of one kind from a 16 KB
// region of its own workgroup; cycles per instruction and CU = time x clock / (ITER x 8 x waves per CU).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;

template <typename T>
__device__ __forceinline__ T load_at(uint8_t const *p)
{
	T v;
	__builtin_memcpy(&v, p, sizeof(T));
	return v;
}

__device__ __forceinline__ uint32_t fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ uint32_t fold(u32x2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }

// pattern 0: lane i at i * sizeof(T) (coalesced); 1: lane i at its own 128-byte line; 2: 5 lanes per 128-byte line, 16 bytes apart
template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void k_loads(uint8_t const *buf, uint32_t misalign, uint32_t iters, uint32_t *out)
{
	uint32_t const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint8_t const *base = buf + (uint64_t) blockIdx.x * 65536 + wave * 16384;
	uint32_t off;
	if (0 == PATTERN) off = lane * sizeof(T);
	else if (1 == PATTERN) off = lane * 128;
	else off = (lane / 5) * 128 + (lane % 5) * 16;
	off += misalign;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; ++it)
	{
		T v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			v[k] = load_at <T>(base + ((off + k * 1024 + it * 64) & 8191));
#pragma unroll
		for (int k = 0; k < 8; ++k)
			acc ^= fold(v[k]);
	}
	if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <typename T, int PATTERN>
static void run(char const *name, uint8_t *buf, uint32_t *out, uint32_t misalign)
{
	uint32_t const iters = 2000, blocks = 256 * 4;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 3; ++rep)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((k_loads <T, PATTERN>), dim3(blocks), dim3(256), 0, 0, buf, misalign, iters, out);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	double const insts_per_cu = (double) iters * 8 * 16;		// 16 waves per CU
	printf("%-34s misalign %2u: %6.1f cycles per instruction and CU (at 2.4 GHz), %.3f ms\n", name, misalign, best * 1e-3 * 2.4e9 / insts_per_cu, best);
}

int main()
{
	uint8_t *buf; uint32_t *out;
	hipMalloc(&buf, 256 * 4 * 65536 + 65536); hipMemset(buf, 1, 256 * 4 * 65536 + 65536);
	hipMalloc(&out, 4096);
	for (uint32_t m : {0u, 4u, 1u})
	{
		run <uint32_t, 0>("b32 coalesced", buf, out, m);
		run <u32x2, 0>("b64 coalesced", buf, out, m);
		run <u32x4, 0>("b128 coalesced", buf, out, m);
		run <uint32_t, 1>("b32 one line per lane", buf, out, m);
		run <u32x4, 1>("b128 one line per lane", buf, out, m);
		run <u32x4, 2>("b128 five lanes per line", buf, out, m);
	}
	return 0;
}
