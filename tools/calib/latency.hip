// latency.hip — dependent-load latency on MI355X: one lane chases pointers through tables of different sizes.
// (How long one step of a dependent chain takes bounds every lane-per-record kernel at low occupancy; DESIGN.md.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>

__global__ void k_chase(uint32_t const *next, uint32_t start, uint32_t steps, uint32_t *out)
{
	uint32_t i = start + threadIdx.x;
	for (uint32_t s = 0; s < steps; ++s) i = next[i];
	out[threadIdx.x] = i;
}

int main()
{
	uint32_t *out; hipMalloc(&out, 256);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (uint64_t mb : {1ull, 64ull, 1024ull, 8192ull})
	{
		uint64_t const n = mb * (1ull << 20) / 4;
		// one random cycle over all n slots with a stride of 16 slots (64 bytes): every step is a new cache line
		uint64_t const lines = n / 16;
		std::vector <uint32_t> perm(lines);
		std::iota(perm.begin(), perm.end(), 0u);
		std::mt19937_64 rng(5);
		std::shuffle(perm.begin(), perm.end(), rng);
		std::vector <uint32_t> host(n, 0);
		for (uint64_t k = 0; k < lines; ++k)
			host[(uint64_t) perm[k] * 16] = perm[(k + 1) % lines] * 16;
		uint32_t *dev; hipMalloc(&dev, n * 4);
		hipMemcpy(dev, host.data(), n * 4, hipMemcpyHostToDevice);
		uint32_t const steps = 20000;
		for (int rep = 0; rep < 2; ++rep)
		{
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, dev, perm[0] * 16, steps, out);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("table %5llu MiB: %.0f ns per dependent load\n", (unsigned long long) mb, ms * 1e6 / steps);
		}
		hipFree(dev);
	}
	return 0;
}
