// What a process that has used the GPU pays at exit, by kind and amount of memory it holds (see exit_cost.cpp).
// usage: exit_cost_hip PINNED_GB VRAM_GB [free|plain|thp] [PLAIN_GB]     "free": hipHostFree / hipFree before the exit, timed;
// plain / thp: PLAIN_GB of ordinary memory touched as well (thp: madvise(MADV_HUGEPAGE))
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <unistd.h>
#include <sys/mman.h>

int main(int argc, char **argv)
{
	double const pinned_gb = argc > 1 ? std::atof(argv[1]) : 0.0, vram_gb = argc > 2 ? std::atof(argv[2]) : 0.0;
	bool const do_free = argc > 3 && 0 == std::strcmp(argv[3], "free");
	bool const thp = argc > 3 && 0 == std::strcmp(argv[3], "thp");
	bool const dontneed = argc > 3 && 0 == std::strcmp(argv[3], "dontneed"), unmap = argc > 3 && 0 == std::strcmp(argv[3], "unmap");
	std::vector <char *> plain;
	double const plain_gb = argc > 4 ? std::atof(argv[4]) : 0.0;
	auto const t0 = std::chrono::steady_clock::now();
	if (hipSuccess != hipSetDevice(0) || hipSuccess != hipFree(nullptr)) return 1;
	std::vector <void *> pinned, vram;
	std::size_t const piece = std::size_t(128) << 20;
	for (std::size_t i = 0; i < (std::size_t) (pinned_gb * 8); ++i)
	{
		void *p{};
		if (hipSuccess != hipHostMalloc(&p, piece, hipHostMallocDefault)) return 2;
		std::memset(p, 1, piece);
		pinned.push_back(p);
	}
	for (std::size_t i = 0; i < (std::size_t) (vram_gb * 8); ++i)
	{
		void *p{};
		if (hipSuccess != hipMalloc(&p, piece) || hipSuccess != hipMemset(p, 1, piece)) return 3;
		vram.push_back(p);
	}
	for (std::size_t i = 0; i < (std::size_t) (plain_gb * 8); ++i)
	{
		char *p = (char *) mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (MAP_FAILED == p) return 5;
		if (thp) madvise(p, piece, MADV_HUGEPAGE);
		std::memset(p, 1, piece);
		plain.push_back(p);
	}
	if (hipSuccess != hipDeviceSynchronize()) return 4;
	double const setup = std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count();
	auto const t1 = std::chrono::steady_clock::now();
	if (do_free)
	{
		for (void *p : pinned) (void) hipHostFree(p);
		for (void *p : vram) (void) hipFree(p);
	}
	if (dontneed || unmap)
	{
		std::vector <std::thread> ts;
		for (unsigned t = 0; t < 16; ++t)
			ts.emplace_back([&, t]() { for (std::size_t i = t; i < plain.size(); i += 16) { if (unmap) munmap(plain[i], piece); else madvise(plain[i], piece, MADV_DONTNEED); } });
		for (auto &t : ts) t.join();
	}
	double const freeing = std::chrono::duration <double>(std::chrono::steady_clock::now() - t1).count();
	std::printf("%.6f setup %.3f s, freeing %.3f s\n", std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count(), setup, freeing);
	std::fflush(stdout);
	_Exit(0);
}
