// Test driver for panvc3_amd/host/fast_inflate.hpp (tests/test_host_io_cpu.py): raw DEFLATE streams made by zlib at levels 0, 1,
// 6 and 9 with the default, fixed-Huffman, Huffman-only and run-length strategies (stored, fixed and dynamic blocks; matches at
// every distance up to 32 K, lengths up to 258, overlapping copies) over empty, constant, random, four-letter, mixed and
// text inputs of sizes around every boundary of the decoder.  Each must inflate to its input; with an output one byte too
// short, or an input three bytes too short, the decoder must refuse.  Exit code 0 iff all of it holds.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../panvc3_amd/host/fast_inflate.hpp"

static std::vector <uint8_t> deflate_raw(std::vector <uint8_t> const &in, int level, int strategy)
{
	z_stream zs{};
	deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
	std::vector <uint8_t> out(compressBound((uLong) in.size()) + 64);
	zs.next_in = const_cast <uint8_t *>(in.data()); zs.avail_in = (uInt) in.size();
	zs.next_out = out.data(); zs.avail_out = (uInt) out.size();
	deflate(&zs, Z_FINISH);
	out.resize(zs.total_out);
	deflateEnd(&zs);
	return out;
}

int main()
{
	std::mt19937_64 rng(3);
	pvc3host::fast_inflater inf;
	int bad = 0, n = 0;
	auto check = [&](std::vector <uint8_t> const &data, char const *what) {
		for (int level : {0, 1, 6, 9})
			for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE})
			{
				auto const comp = deflate_raw(data, level, strategy);
				std::vector <uint8_t> out(data.size() + 1, 0xAB);
				bool const ok = inf.inflate(comp.data(), comp.size(), out.data(), data.size());
				++n;
				if (!ok || 0 != std::memcmp(out.data(), data.data(), data.size()) || 0xAB != out[data.size()])
					if (bad++ < 10) std::printf("FAIL %s level %d strategy %d size %zu\n", what, level, strategy, data.size());
				if (!data.empty())
				{
					std::vector <uint8_t> o2(data.size());
					if (inf.inflate(comp.data(), comp.size(), o2.data(), data.size() - 1) && bad++ < 10) std::printf("accepted a short output: %s\n", what);
				}
				if (comp.size() > 4)
				{
					std::vector <uint8_t> o3(data.size());
					if (inf.inflate(comp.data(), comp.size() - 3, o3.data(), data.size()) && bad++ < 10) std::printf("accepted a truncated input: %s\n", what);
				}
			}
	};
	std::vector <uint8_t> d;
	check(d, "empty");
	for (std::size_t sz : {1u, 2u, 7u, 8u, 9u, 100u, 257u, 258u, 259u, 273u, 274u, 275u, 1000u, 32768u, 32769u, 65280u})
	{
		d.assign(sz, 0); check(d, "zeros");
		for (auto &b : d) b = (uint8_t) rng();
		check(d, "random");
		for (std::size_t i = 0; i < sz; ++i) d[i] = (uint8_t) ("ACGT"[rng() & 3]);
		check(d, "acgt");
		for (std::size_t i = 0; i < sz; ++i) d[i] = (uint8_t) (i % 7 < 3 ? 'a' + i % 5 : rng() & 0x0f);
		check(d, "mixed");
	}
	{
		std::string t;
		for (int i = 0; i < 3000; ++i) t += "read_" + std::to_string(rng() % 100000) + "/1\tXS:i:" + std::to_string(rng() % 50) + "\n";
		d.assign(t.begin(), t.begin() + (std::ptrdiff_t) std::min <std::size_t>(t.size(), 65280));
		check(d, "text");
	}
	std::printf("%d streams, %d failures\n", n, bad);
	return bad ? 1 : 0;
}
