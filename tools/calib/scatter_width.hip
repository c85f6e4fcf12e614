// scatter_width.hip — cost of one lane-scattered load instruction on MI355X as a function of its width: every lane reads
// 4, 8 or 16 bytes (or a whole 64-byte line with four 16-byte loads) at its own random 64-byte aligned address of a 4 GiB
// table.  Reports loads per second: if narrow loads are not cheaper than wide ones, only the NUMBER of scattered load
// instructions matters for a lane-per-record kernel, not their bytes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;

__device__ inline uint64_t mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

template <int MODE>
__global__ void k_scatter(uint8_t const *buf, uint64_t n_lines, uint64_t per_thread, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0;
#pragma unroll 4
	for (uint64_t i = 0; i < per_thread; ++i)
	{
		uint8_t const *p = buf + (mix(tid * per_thread + i + MODE * 7919ull) % n_lines) * 64;
		if (0 == MODE) acc += *(uint32_t const *) p;
		if (1 == MODE) acc += *(uint64_t const *) p;
		if (2 == MODE) { u64x2 const v = *(u64x2 const *) p; acc += v.x ^ v.y; }
		if (3 == MODE) { u64x2 const *q = (u64x2 const *) p; u64x2 const a = q[0], b = q[1], c = q[2], d = q[3]; acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y; }
		if (4 == MODE) { acc += *(uint32_t const *) p + *(uint64_t const *) (p + 8); }		// two narrow loads, same line
	}
	if (acc == 0x1234567) sink[0] = acc;
}

int main()
{
	uint64_t const bytes = 4ull << 30;
	uint8_t *buf; uint64_t *sink;
	(void) hipMalloc(&buf, bytes); (void) hipMalloc(&sink, 8);
	(void) hipMemset(buf, 1, bytes);
	uint64_t const per_thread = 64;
	dim3 const grid(8192), block(256);
	uint64_t const n = 8192ull * 256 * per_thread;
	hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	char const *names[5] = {"4 B per lane", "8 B per lane", "16 B per lane", "64 B per lane (4 x 16 B, one line)", "4 B + 8 B per lane, one line"};
	for (int rep = 0; rep < 2; ++rep)
		for (int mode = 0; mode < 5; ++mode)
		{
			(void) hipEventRecord(e0);
			switch (mode)
			{
				case 0: hipLaunchKernelGGL(k_scatter <0>, grid, block, 0, 0, buf, bytes / 64, per_thread, sink); break;
				case 1: hipLaunchKernelGGL(k_scatter <1>, grid, block, 0, 0, buf, bytes / 64, per_thread, sink); break;
				case 2: hipLaunchKernelGGL(k_scatter <2>, grid, block, 0, 0, buf, bytes / 64, per_thread, sink); break;
				case 3: hipLaunchKernelGGL(k_scatter <3>, grid, block, 0, 0, buf, bytes / 64, per_thread, sink); break;
				default: hipLaunchKernelGGL(k_scatter <4>, grid, block, 0, 0, buf, bytes / 64, per_thread, sink); break;
			}
			(void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
			float ms; (void) hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("%-40s %7.3f ms  %6.2f G lines/s\n", names[mode], ms, n / ms / 1e6);
		}
	return 0;
}
