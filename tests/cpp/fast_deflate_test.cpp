// Test driver for panvc3_amd/host/fast_deflate.hpp: stdin -> one raw DEFLATE stream per 65 280-byte piece on stdout, each
// preceded by its length as a 4-byte little-endian integer (tests/test_fast_deflate.py inflates them with zlib).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#include "../../panvc3_amd/host/fast_deflate.hpp"

int main()
{
	std::vector <uint8_t> in;
	uint8_t buf[1 << 16];
	for (std::size_t got; (got = std::fread(buf, 1, sizeof(buf), stdin)) > 0; )
		in.insert(in.end(), buf, buf + got);
	pvc3host::fast_deflater enc;
	std::size_t const piece = 0xff00;
	std::size_t off = 0;
	if (std::getenv("PVC3_DEFLATE_TIMING"))		// compress only, to stderr
	{
		auto const t0 = std::chrono::steady_clock::now();
		std::size_t total = 0;
		std::vector <uint8_t> out(pvc3host::fast_deflater::bound(piece));
		for (std::size_t o = 0; o < in.size(); o += piece)
			total += enc.compress(in.data() + o, std::min(piece, in.size() - o), out.data(), out.size());
		double const dt = std::chrono::duration <double>(std::chrono::steady_clock::now() - t0).count();
		std::fprintf(stderr, "%.1f MB/s, ratio %.3f\n", in.size() / 1e6 / dt, (double) total / in.size());
	}
	do
	{
		std::size_t const n = std::min(piece, in.size() - off);
		std::vector <uint8_t> out(pvc3host::fast_deflater::bound(n));
		std::size_t const c = enc.compress(in.data() + off, n, out.data(), out.size());
		if (0 == c) return 1;
		uint32_t const len = (uint32_t) c;
		std::fwrite(&len, 4, 1, stdout);
		std::fwrite(out.data(), 1, c, stdout);
		off += n;
	}
	while (off < in.size());
	return 0;
}
