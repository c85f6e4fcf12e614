// random_access.hip — how fast can one lane-per-record kernel fetch small random pieces of a multi-GB table on MI355X?
// Informs the HBM layout of the index.  Every thread reads `per_thread` random records; a record is
//   mode 0: one 64-byte block                      mode 1: one 128-byte block (aligned)
//   mode 2: two 64-byte blocks 2 GiB apart         mode 3: 64 B + 64 B + 256 B inside one aligned 1 KiB region
//   mode 4: 64 B + 64 B (2 GiB apart) + 256 B at a third random place (what k_project_fast does today)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;

__device__ inline uint64_t mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

__device__ inline uint64_t rd(uint8_t const *p, int n16)
{
	uint64_t acc = 0;
	for (int k = 0; k < n16; ++k) { u64x2 const v = ((u64x2 const *) p)[k]; acc += v.x ^ v.y; }
	return acc;
}

template <int MODE>
__global__ void k_random(uint8_t const *buf, uint64_t bytes, uint64_t per_thread, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t const half = bytes / 2;
	uint64_t acc = 0;
	for (uint64_t i = 0; i < per_thread; ++i)
	{
		uint64_t const h = mix(tid * per_thread + i + MODE * 1000003ull);
		if (0 == MODE) acc += rd(buf + (h % (bytes / 64)) * 64, 4);
		if (1 == MODE) acc += rd(buf + (h % (bytes / 128)) * 128, 8);
		if (2 == MODE) { uint64_t const b = (h % (half / 64)) * 64; acc += rd(buf + b, 4) + rd(buf + half + b, 4); }
		if (3 == MODE) { uint64_t const b = (h % (bytes / 1024)) * 1024; acc += rd(buf + b, 4) + rd(buf + b + 64, 4) + rd(buf + b + 512 + 16 * (h >> 60), 16); }
		if (4 == MODE) { uint64_t const b = (h % (half / 64)) * 64; acc += rd(buf + b, 4) + rd(buf + half + b, 4) + rd(buf + (mix(h) % (bytes / 16 - 64)) * 16, 16); }
	}
	if (acc == 0x1234567) sink[0] = acc;
}

// Cooperative variants: LANES_PER_REC lanes read one record's contiguous bytes with one 16-byte load each, so that one
// wave instruction touches 64 / LANES_PER_REC records instead of 64.
//   mode 5: 64 B block by 4 lanes      mode 6: 256 B by 16 lanes      mode 7: 64 + 64 (far apart) by 4+4 lanes, 256 B by 16 lanes
//   mode 8: like 7 with the two 64-byte blocks next to each other in one aligned 128-byte line (the paired index layout)
//   mode 9: 64 B by 4 lanes at 128-byte aligned places (is the other half of the line fetched as well?)
template <int MODE>
__global__ void k_coop(uint8_t const *buf, uint64_t bytes, uint64_t per_thread, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t const wave_base = (tid & ~63ull) * per_thread;		// first record of this wave
	unsigned const lane = threadIdx.x & 63;
	uint64_t const half = bytes / 2;
	uint64_t acc = 0;
	uint64_t const n_wave = 64 * per_thread;
	if (5 == MODE)
		for (uint64_t i = 0; i < n_wave; i += 16)
		{
			uint64_t const h = mix(wave_base + i + lane / 4 + MODE * 1000003ull);
			acc += rd(buf + (h % (bytes / 64)) * 64 + 16 * (lane & 3), 1);
		}
	if (6 == MODE)
		for (uint64_t i = 0; i < n_wave; i += 4)
		{
			uint64_t const h = mix(wave_base + i + lane / 16 + MODE * 1000003ull);
			acc += rd(buf + (h % (bytes / 16 - 64)) * 16 + 16 * (lane & 15), 1);
		}
	if (7 == MODE)
	{
		for (uint64_t i = 0; i < n_wave; i += 8)
		{
			uint64_t const h = mix(wave_base + i + lane / 8 + MODE * 1000003ull);
			uint64_t const b = (h % (half / 64)) * 64;
			acc += rd(buf + ((lane & 4) ? half : 0) + b + 16 * (lane & 3), 1);
		}
		for (uint64_t i = 0; i < n_wave; i += 4)
		{
			uint64_t const h = mix(mix(wave_base + i + lane / 16 + MODE * 1000003ull));
			acc += rd(buf + (h % (bytes / 16 - 64)) * 16 + 16 * (lane & 15), 1);
		}
	}
	if (8 == MODE)
	{
		for (uint64_t i = 0; i < n_wave; i += 8)
		{
			uint64_t const h = mix(wave_base + i + lane / 8 + MODE * 1000003ull);
			uint64_t const b = (h % (bytes / 128)) * 128;
			acc += rd(buf + b + ((lane & 4) ? 64 : 0) + 16 * (lane & 3), 1);
		}
		for (uint64_t i = 0; i < n_wave; i += 4)
		{
			uint64_t const h = mix(mix(wave_base + i + lane / 16 + MODE * 1000003ull));
			acc += rd(buf + (h % (bytes / 16 - 64)) * 16 + 16 * (lane & 15), 1);
		}
	}
	if (9 == MODE)
		for (uint64_t i = 0; i < n_wave; i += 16)
		{
			uint64_t const h = mix(wave_base + i + lane / 4 + MODE * 1000003ull);
			acc += rd(buf + (h % (bytes / 128)) * 128 + 16 * (lane & 3), 1);
		}
	if (acc == 0x1234567) sink[0] = acc;
}

int main()
{
	uint64_t const bytes = 4ull << 30;
	uint8_t *buf; uint64_t *sink;
	hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
	hipMemset(buf, 1, bytes);
	uint64_t const per_thread = 32;
	dim3 const grid(8192), block(256);
	uint64_t const n = 8192ull * 256 * per_thread;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	char const *names[10] = {"64 B", "128 B aligned", "64 B + 64 B far apart", "64 + 64 + 256 B in one 1 KiB region", "64 + 64 far apart + 256 B elsewhere", "coop: 64 B by 4 lanes", "coop: 256 B by 16 lanes", "coop: 64 + 64 far apart + 256 B elsewhere", "coop: 64 + 64 in one line + 256 B elsewhere", "coop: 64 B by 4 lanes, 128-B aligned"};
	uint64_t const useful[10] = {64, 128, 128, 384, 384, 64, 256, 384, 384, 64};
	for (int rep = 0; rep < 2; ++rep)
		for (int mode = 0; mode < 10; ++mode)
		{
			hipEventRecord(e0);
			switch (mode)
			{
				case 0: hipLaunchKernelGGL(k_random <0>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 1: hipLaunchKernelGGL(k_random <1>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 2: hipLaunchKernelGGL(k_random <2>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 3: hipLaunchKernelGGL(k_random <3>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 4: hipLaunchKernelGGL(k_random <4>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 5: hipLaunchKernelGGL(k_coop <5>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 6: hipLaunchKernelGGL(k_coop <6>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 7: hipLaunchKernelGGL(k_coop <7>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				case 8: hipLaunchKernelGGL(k_coop <8>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
				default: hipLaunchKernelGGL(k_coop <9>, grid, block, 0, 0, buf, bytes, per_thread, sink); break;
			}
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("mode %d  %-40s  %7.3f ms  %6.2f G records/s  %6.2f TB/s useful\n", mode, names[mode], ms, n / ms / 1e6, n * useful[mode] / ms / 1e9);
		}
	return 0;
}
