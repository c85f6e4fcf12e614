// crc32_device.hpp — CRC-32 (the gzip polynomial) of many byte ranges on the GPU, one wave per range; used by both files of
// libpvc3_bgzf.so for the device-resident forms of the two codecs (include/pvc3_bgzf.h): the inflated blocks of the input are
// checked against the checksums in the file, and the blocks of the output get theirs, without the bytes visiting the host.
//
// Lane l takes the l-th 64th of the range (a multiple of four bytes) through the byte table in LDS; the lanes' checksums are
// put together the way zlib's crc32_combine does it: crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B), so the checksum of
// the whole is the xor over the lanes of crc(chunk) * x^(8 * bytes behind the chunk).  x^(2^k) mod P for k < 32 come from a
// table; a product modulo P is 32 shift-and-xor steps.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pvc3_crc {

constexpr uint32_t POLY = 0xedb88320u;

struct tables
{
	uint32_t byte_table[256];
	uint32_t x2n[32];		// x^(2^k) mod P, reflected
};

inline uint32_t host_multmodp(uint32_t a, uint32_t b)
{
	uint32_t m = 1u << 31, p = 0;
	for (;;)
	{
		if (a & m) { p ^= b; if (0 == (a & (m - 1))) break; }
		m >>= 1;
		b = (b & 1) ? ((b >> 1) ^ POLY) : (b >> 1);
	}
	return p;
}

inline tables make_tables()
{
	tables t{};
	for (uint32_t i = 0; i < 256; ++i)
	{
		uint32_t c = i;
		for (int k = 0; k < 8; ++k) c = (c & 1) ? ((c >> 1) ^ POLY) : (c >> 1);
		t.byte_table[i] = c;
	}
	uint32_t p = 1u << 30;		// x^1
	t.x2n[0] = p;
	for (int k = 1; k < 32; ++k) { p = host_multmodp(p, p); t.x2n[k] = p; }
	return t;
}

__device__ inline uint32_t multmodp(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (int k = 31; k >= 0; --k)		// bit 31 of the reflected form is x^0
	{
		if ((a >> k) & 1u) p ^= b;
		b = (b & 1) ? ((b >> 1) ^ POLY) : (b >> 1);
	}
	return p;
}

// x^(8 n) mod P
__device__ inline uint32_t x8nmodp(uint32_t n, uint32_t const *x2n)
{
	uint32_t p = 1u << 31;		// x^0
	uint32_t k = 3;
	while (n)
	{
		if (n & 1) p = multmodp(x2n[k & 31], p);
		n >>= 1; ++k;
	}
	return p;
}

// One wave (64 threads) per range: range i = base + off[i] .. + len[i]; with `stride` non-zero range i begins at base + i * stride + pad instead.
static __global__ __launch_bounds__(64) void k_crc32_ranges(uint8_t const *__restrict__ base, uint64_t const *__restrict__ off, uint32_t const *__restrict__ len, uint32_t stride, uint32_t pad,
	uint32_t n_ranges, tables const *__restrict__ T, uint32_t *__restrict__ crc_out)
{
	__shared__ uint32_t tab[256];
	__shared__ uint32_t x2n[32];
	unsigned const lane = threadIdx.x;
	for (unsigned i = lane; i < 256; i += 64) tab[i] = T->byte_table[i];
	if (lane < 32) x2n[lane] = T->x2n[lane];
	__syncthreads();
	for (uint32_t r = blockIdx.x; r < n_ranges; r += gridDim.x)
	{
		uint8_t const *p = stride ? (base + (size_t) r * stride + pad) : (base + off[r]);
		uint32_t const n = len[r];
		uint32_t const chunk = ((n + 63) / 64 + 3) & ~3u;
		uint32_t const lo = (lane * chunk < n) ? lane * chunk : n, hi = (lo + chunk < n) ? (lo + chunk) : n;
		uint32_t c = 0;
		if (hi > lo)
		{
			c = 0xffffffffu;
			uint32_t x = lo;
			// bytes up to a dword boundary of the address, whole dwords, the rest
			while (x < hi && (((uintptr_t) (p + x)) & 3)) { c = tab[(c ^ p[x]) & 0xff] ^ (c >> 8); ++x; }
			for (; x + 4 <= hi; x += 4)
			{
				uint32_t const w = *(uint32_t const *) (p + x);
				c ^= w;
				c = tab[c & 0xff] ^ (c >> 8); c = tab[c & 0xff] ^ (c >> 8); c = tab[c & 0xff] ^ (c >> 8); c = tab[c & 0xff] ^ (c >> 8);
			}
			for (; x < hi; ++x) c = tab[(c ^ p[x]) & 0xff] ^ (c >> 8);
			c ^= 0xffffffffu;
			if (n > hi) c = multmodp(x8nmodp(n - hi, x2n), c);
		}
		for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t) __shfl_xor((int) c, d);
		if (0 == lane) crc_out[r] = c;
	}
}

}
