// What a process pays at exit for the memory it has touched (the 0.2-0.3 s between panvc3_project_alignments' last line and the
// parent's wait() returning).  usage: exit_cost GB MODE THREADS    MODE: plain | thp | free (munmap by THREADS threads before exit)
// | dontneed (madvise(MADV_DONTNEED) by THREADS threads before exit).  Prints the wall clock right before _Exit.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <unistd.h>

int main(int argc, char **argv)
{
	double const gb = argc > 1 ? std::atof(argv[1]) : 3.0;
	char const *mode = argc > 2 ? argv[2] : "plain";
	unsigned const threads = argc > 3 ? (unsigned) std::atoi(argv[3]) : 16;
	std::size_t const piece = std::size_t(64) << 20, n = (std::size_t) (gb * 1024.0 / 64.0);
	std::vector <char *> pieces(n);
	for (auto &p : pieces)
	{
		p = (char *) mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (MAP_FAILED == p) return 1;
		if (0 == std::strcmp(mode, "thp")) madvise(p, piece, MADV_HUGEPAGE);
	}
	auto const t0 = std::chrono::steady_clock::now();
	{
		std::vector <std::thread> ts;
		for (unsigned t = 0; t < threads; ++t)
			ts.emplace_back([&, t]() { for (std::size_t i = t; i < n; i += threads) std::memset(pieces[i], 1, piece); });
		for (auto &t : ts) t.join();
	}
	double const touch = std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count();
	auto const t1 = std::chrono::steady_clock::now();
	if (0 == std::strcmp(mode, "free") || 0 == std::strcmp(mode, "dontneed"))
	{
		bool const unmap = 'f' == mode[0];
		std::vector <std::thread> ts;
		for (unsigned t = 0; t < threads; ++t)
			ts.emplace_back([&, t]() { for (std::size_t i = t; i < n; i += threads) { if (unmap) munmap(pieces[i], piece); else madvise(pieces[i], piece, MADV_DONTNEED); } });
		for (auto &t : ts) t.join();
	}
	double const freeing = std::chrono::duration <double>(std::chrono::steady_clock::now() - t1).count();
	std::printf("%.6f touch %.3f s, freeing %.3f s\n", std::chrono::duration <double>(std::chrono::system_clock::now().time_since_epoch()).count(), touch, freeing);
	std::fflush(stdout);
	_Exit(0);
}
