// fast_crc32.hpp — CRC-32 (the gzip / BGZF polynomial, reflected 0xEDB88320) by carry-less multiplication.
// Every BGZF block is checksummed once when it is read and once when it is written; zlib's table-driven crc32 was a tenth of
// the CPU time of panvc3_project_alignments (2.5 GB per four million records at ~1.5 GB/s per core).  Folding 64 bytes per
// iteration with PCLMULQDQ (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
// Intel 2009; constants for this polynomial as published there) runs at several times that.  Checked against zlib's crc32
// on every length in tests/test_host_io_cpu.py; CPUs without the instruction take zlib's.
#pragma once

#include <zlib.h>

#include <cstddef>
#include <cstdint>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace pvc3host {

#if defined(__x86_64__)
// State update (no pre- / post-inversion) over len bytes, len >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1")))
inline uint32_t crc32_fold_pclmul(uint32_t state, uint8_t const *buf, std::size_t len)
{
	// x^(4*128+32) mod P, x^(4*128-32) mod P | x^(128+32), x^(128-32) | x^64 | P and its Barrett constant (bit-reflected forms)
	__m128i const k1k2 = _mm_set_epi64x(0x00000001c6e41596, 0x0000000154442bd4);
	__m128i const k3k4 = _mm_set_epi64x(0x00000000ccaa009e, 0x00000001751997d0);
	__m128i const k5 = _mm_set_epi64x(0, 0x0000000163cd6124);
	__m128i const poly_mu = _mm_set_epi64x(0x00000001F7011641, 0x00000001DB710641);
	__m128i const mask32 = _mm_set_epi32(0, 0, 0, -1);
	__m128i const *p = (__m128i const *) buf;
	__m128i x1 = _mm_loadu_si128(p), x2 = _mm_loadu_si128(p + 1), x3 = _mm_loadu_si128(p + 2), x4 = _mm_loadu_si128(p + 3);
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int) state));
	p += 4; len -= 64;
	while (len >= 64)
	{
		__m128i const a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
		__m128i const a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
		__m128i const b1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), b2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
		__m128i const b3 = _mm_clmulepi64_si128(x3, k1k2, 0x11), b4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(a1, b1), _mm_loadu_si128(p));
		x2 = _mm_xor_si128(_mm_xor_si128(a2, b2), _mm_loadu_si128(p + 1));
		x3 = _mm_xor_si128(_mm_xor_si128(a3, b3), _mm_loadu_si128(p + 2));
		x4 = _mm_xor_si128(_mm_xor_si128(a4, b4), _mm_loadu_si128(p + 3));
		p += 4; len -= 64;
	}
	// four accumulators -> one (a lambda would not inherit the target attribute)
#define PVC3_CRC_FOLD(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x00), _mm_clmulepi64_si128(acc, k3k4, 0x11)), next)
	x1 = PVC3_CRC_FOLD(x1, x2); x1 = PVC3_CRC_FOLD(x1, x3); x1 = PVC3_CRC_FOLD(x1, x4);
	while (len >= 16)
	{
		__m128i const next = _mm_loadu_si128(p);
		x1 = PVC3_CRC_FOLD(x1, next);
		++p; len -= 16;
	}
#undef PVC3_CRC_FOLD
	// 128 -> 64 bits (this also appends the 32 zero bits of the CRC's definition), 64 -> 32, Barrett reduction
	__m128i t = _mm_clmulepi64_si128(k3k4, x1, 0x01);		// high half of the constants x low half of the accumulator
	x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
	__m128i x2b = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, mask32);
	x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), x2b);
	__m128i const keep = x1;
	x1 = _mm_and_si128(x1, mask32);
	x1 = _mm_clmulepi64_si128(x1, poly_mu, 0x10);
	x1 = _mm_and_si128(x1, mask32);
	x1 = _mm_clmulepi64_si128(x1, poly_mu, 0x00);
	x1 = _mm_xor_si128(x1, keep);
	return (uint32_t) _mm_extract_epi32(x1, 1);
}
#endif

// zlib's crc32(crc, buf, len), same arguments and result.
inline uint32_t fast_crc32(uint32_t crc, uint8_t const *buf, std::size_t len)
{
#if defined(__x86_64__)
	static bool const have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (have && len >= 64)
	{
		std::size_t const body = len & ~(std::size_t) 15;
		crc = ~crc32_fold_pclmul(~crc, buf, body);
		buf += body; len -= body;
	}
#endif
	while (len)
	{
		uInt const piece = (uInt) ((len > (1u << 30)) ? (1u << 30) : len);
		crc = (uint32_t) crc32(crc, buf, piece);
		buf += piece; len -= piece;
	}
	return crc;
}

}
