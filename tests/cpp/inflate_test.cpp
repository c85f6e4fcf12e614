// Test driver for panvc3_amd/host/fast_inflate.hpp (tests/test_host_io_cpu.py): raw DEFLATE streams made by zlib at levels 0, 1,
// 6 and 9 with the default, fixed-Huffman, Huffman-only and run-length strategies (stored, fixed and dynamic blocks; matches at
// every distance up to 32 K, lengths up to 258, overlapping copies) over empty, constant, random, four-letter, mixed and
// text inputs of sizes around every boundary of the decoder.  Each must inflate to its input; with an output one byte too
// short, or an input three bytes too short, the decoder must refuse.  Exit code 0 iff all of it holds.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
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

// `fuzz`: streams with flipped bits, random bytes, random cuts and wrong output sizes, from and into heap buffers of exactly
// the sizes given (built with -fsanitize=address,undefined by tests/test_host_io_cpu.py: a read or write outside them, or a
// pointer that leaves its array, ends the program).  Whatever the decoder accepts must be what zlib makes of the same bytes.
static int fuzz(unsigned rounds)
{
	std::mt19937_64 rng(11);
	pvc3host::fast_inflater inf;
	std::vector <std::vector <uint8_t>> plain;
	for (std::size_t sz : {0u, 1u, 5u, 300u, 4000u, 40000u, 65280u})
	{
		std::vector <uint8_t> d(sz);
		for (std::size_t i = 0; i < sz; ++i) d[i] = (uint8_t) ((i % 11 < 4) ? ('a' + i % 3) : (rng() & ((i & 64) ? 0xff : 0x03)));
		plain.push_back(d);
	}
	int bad = 0;
	unsigned accepted = 0;
	for (unsigned r = 0; r < rounds; ++r)
	{
		auto const &data = plain[rng() % plain.size()];
		static int const levels[4] = {0, 1, 6, 9};
		static int const strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
		std::vector <uint8_t> comp = deflate_raw(data, levels[rng() % 4], strategies[rng() % 4]);
		std::size_t out_len = data.size();
		switch (rng() % 6)
		{
			case 0: for (int k = 0; k < 1 + (int) (rng() % 4) && !comp.empty(); ++k) comp[rng() % comp.size()] ^= (uint8_t) (1u << (rng() % 8)); break;
			case 1: for (auto &b : comp) if (0 == rng() % 50) b = (uint8_t) rng(); break;
			case 2: comp.resize(rng() % (comp.size() + 1)); break;
			case 3: comp.resize(rng() % 600); for (auto &b : comp) b = (uint8_t) rng(); break;
			case 4: out_len = rng() % (data.size() + 300); break;
			default: break;		// as it is
		}
		// exact-size heap buffers (a vector's capacity may be larger than its size)
		uint8_t *in = new uint8_t[comp.size() ? comp.size() : 1], *out = new uint8_t[out_len ? out_len : 1];
		if (!comp.empty()) std::memcpy(in, comp.data(), comp.size());
		bool const ok = inf.inflate(comp.empty() ? in + 1 : in, comp.size(), out_len ? out : out + 1, out_len);
		if (ok)
		{
			++accepted;
			// zlib must make the same bytes of it
			std::vector <uint8_t> ref(out_len + 1);
			z_stream zs{};
			inflateInit2(&zs, -15);
			zs.next_in = in; zs.avail_in = (uInt) comp.size(); zs.next_out = ref.data(); zs.avail_out = (uInt) ref.size();
			int const rc = inflate(&zs, Z_FINISH);
			bool const same = Z_STREAM_END == rc && zs.total_out == out_len && 0 == std::memcmp(ref.data(), out, out_len);
			inflateEnd(&zs);
			if (!same && bad++ < 10) std::printf("FAIL round %u: accepted what zlib does not make the same bytes of (rc %d, %lu of %zu)\n", r, rc, (unsigned long) zs.total_out, out_len);
		}
		delete[] in; delete[] out;
	}
	std::printf("%u rounds, %u accepted, %d failures\n", rounds, accepted, bad);
	return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
	if (argc > 1 && 0 == std::strcmp(argv[1], "fuzz")) return fuzz(argc > 2 ? (unsigned) std::atoi(argv[2]) : 20000u);
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
