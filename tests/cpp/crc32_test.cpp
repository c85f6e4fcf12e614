// Test driver for panvc3_amd/host/fast_crc32.hpp (tests/test_host_io_cpu.py): fast_crc32 against zlib's crc32 on every length
// up to 1200 bytes at three alignments with running values, and on BGZF-block-sized inputs.  Exit code 0 iff all agree.
#include <cstdio>
#include <random>
#include <vector>

#include "../../panvc3_amd/host/fast_crc32.hpp"

int main()
{
	std::mt19937_64 rng(7);
	std::vector <uint8_t> buf(70000);
	for (auto &b : buf) b = (uint8_t) rng();
	int bad = 0;
	for (std::size_t len = 0; len <= 1200; ++len)
		for (std::size_t off = 0; off < 3; ++off)
		{
			uint32_t const init = (len & 1) ? 0 : (uint32_t) rng();
			uint32_t const a = (uint32_t) crc32(init, buf.data() + off, (uInt) len), b = pvc3host::fast_crc32(init, buf.data() + off, len);
			if (a != b && bad++ < 5) std::printf("len %zu off %zu: zlib %08x, fast %08x\n", len, off, a, b);
		}
	for (std::size_t len : {65279u, 65280u, 69999u})
		if ((uint32_t) crc32(0, buf.data(), (uInt) len) != pvc3host::fast_crc32(0, buf.data(), len)) { ++bad; std::printf("len %zu differs\n", len); }
	std::printf("%d mismatches\n", bad);
	return bad ? 1 : 0;
}
