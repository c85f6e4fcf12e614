// How fast 2 GB reach a new file in /tmp, 35 MB at a time (the writer thread of the record path on the GPU): pwrite from 1 / 8 / 16
// threads at their own offsets, a mapping of the file filled by 8 / 16 threads (ftruncate + mmap per piece), and pwrite after
// fallocate.  usage: file_write [dir]
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration <double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	std::string const dir = argc > 1 ? argv[1] : "/tmp";
	size_t const piece = 35u << 20, pieces = 58;
	std::vector <char> src(piece);
	for (size_t i = 0; i < piece; ++i) src[i] = (char) (i * 2654435761u >> 13);
	auto run = [&](char const *name, int threads, int mode) {
		std::string const path = dir + "/pvc3_file_write.tmp";
		unlink(path.c_str());
		int const fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
		if (fd < 0) { perror("open"); return; }
		double const t0 = now();
		if (2 == mode && fallocate(fd, 0, 0, (off_t) (piece * pieces))) perror("fallocate");
		for (size_t p = 0; p < pieces; ++p)
		{
			off_t const at = (off_t) (p * piece);
			char *map = nullptr;
			if (1 == mode)
			{
				if (ftruncate(fd, at + (off_t) piece)) perror("ftruncate");
				map = (char *) mmap(nullptr, piece, PROT_READ | PROT_WRITE, MAP_SHARED, fd, at);
				if (MAP_FAILED == map) { perror("mmap"); return; }
			}
			std::vector <std::thread> ts;
			for (int k = 0; k < threads; ++k)
				ts.emplace_back([&, k]() {
					size_t const lo = piece * k / threads, hi = piece * (k + 1) / threads;
					if (1 == mode) std::memcpy(map + lo, src.data() + lo, hi - lo);
					else
						for (size_t o = lo; o < hi; )
						{
							ssize_t const w = pwrite(fd, src.data() + o, hi - o, at + (off_t) o);
							if (w <= 0) { perror("pwrite"); return; }
							o += (size_t) w;
						}
				});
			for (auto &t : ts) t.join();
			if (map) munmap(map, piece);
		}
		double const t1 = now();
		close(fd);
		double const t2 = now();
		std::printf("%-44s %2d threads: %.3f s = %.1f GB/s (close %.3f s)\n", name, threads, t1 - t0, piece * pieces / (t1 - t0) / 1e9, t2 - t1);
		unlink(path.c_str());
	};
	run("pwrite", 1, 0); run("pwrite", 8, 0); run("pwrite", 16, 0);
	run("mmap + memcpy", 8, 1); run("mmap + memcpy", 16, 1);
	run("fallocate, then pwrite", 8, 2);
	return 0;
}
