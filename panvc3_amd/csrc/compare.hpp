// compare.hpp — the base comparison of rewrite_cigar (include/panvc3/rewrite_cigar.hh:96-104) on 4-bit codes.
//
// The query is BAM SEQ (two bases per byte, high nibble first); the destination text is the packed copy made by
// k_pack_text4 (kernels.hip): A C G T N -> 1 2 4 8 15, any other byte -> 0, low nibble first.
#pragma once

#include <hip/hip_runtime.h>

#include "pvc3_internal.hpp"

namespace pvc3 {

typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;

__device__ __forceinline__ u64x2 load_u128_unaligned(uint8_t const *p)
{
	u64x2 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}

// A load from global memory through a pointer whose address space the compiler cannot see (it came out of a structure in
// memory): as a flat load it would also count against the LDS counter and wait with the LDS traffic around it.
template <typename t_value>
__device__ __forceinline__ t_value global_load(t_value const *p)
{
	typedef t_value __attribute__((address_space(1))) const *gptr;
	return *(gptr) (uintptr_t) p;
}

// 40 consecutive bytes from any address in global memory as five 64-bit words, fetched as 44 bytes from the dword-aligned
// address below it and shifted into place.  A 16-byte load from an address that is not a multiple of four costs the texture
// addresser three to four times as much (tools/calib/ta_cost.hip: 258 against 71 cycles with one line per lane).  Reads up
// to 3 bytes before p and 4 after p + 40: the buffers this is used on are padded.
__device__ __forceinline__ void load_bytes40(uint8_t const *p, uint64_t (&w)[5])
{
	typedef uint32_t __attribute__((address_space(1))) const *gptr;
	uintptr_t const a = (uintptr_t) p;
	gptr const q = (gptr) (a & ~(uintptr_t) 3);
	uint32_t const sh = 8 * (uint32_t) (a & 3);
	uint32_t d[11];
#pragma unroll
	for (uint32_t i = 0; i < 11; ++i) d[i] = q[i];
#pragma unroll
	for (uint32_t k = 0; k < 5; ++k)
		w[k] = (uint64_t) __builtin_amdgcn_alignbit(d[2 * k + 1], d[2 * k], sh) | ((uint64_t) __builtin_amdgcn_alignbit(d[2 * k + 2], d[2 * k + 1], sh) << 32);
}

// Packed BAM bases (high nibble first) -> nibble-linear: nibble n of the byte string at bits [4n, 4n + 4).
__device__ __forceinline__ uint64_t nibble_linear(uint64_t w)
{
	return ((w & UINT64_C(0x0F0F0F0F0F0F0F0F)) << 4) | ((w >> 4) & UINT64_C(0x0F0F0F0F0F0F0F0F));
}

// Bit k set iff base k of q (8 BAM codes, 4 bits each, read through dna5::to_char) differs from text base k of t (8 codes of
// the packed text, kernels.hip k_pack_text4).  to_char maps 1 2 4 8 to A C G T and every other code to N, so the query
// code is first canonicalised (exactly one bit set ? itself : 15) and then compared nibble-wise, all eight at once.
__device__ __forceinline__ uint32_t mismatch8(uint32_t q, uint32_t t)
{
	uint32_t p = q - ((q >> 1) & 0x55555555u);
	p = (p & 0x33333333u) + ((p >> 2) & 0x33333333u);		// bits set per nibble
	uint32_t const z = p ^ 0x11111111u;						// zero nibble <=> a base
	uint32_t nb = z | (z >> 1);
	nb = (nb | (nb >> 2)) & 0x11111111u;					// 1 per nibble that is not a base
	uint32_t const d = (q | (nb * 15u)) ^ t;
	uint32_t nz = d | (d >> 1);
	nz = (nz | (nz >> 2)) & 0x11111111u;					// 1 per differing nibble, at bit 4k
	uint32_t const u = (nz | (nz >> 3)) & 0x03030303u;		// two flags per byte
	return (u * 0x01041040u) >> 24;
}

// 16 consecutive query base codes starting at base qi of SEQ, nibble-linear (reads 16 bytes at seq + qi / 2).
__device__ __forceinline__ uint64_t load_query16(uint8_t const *seq, uint32_t qi)
{
	u64x2 const v = load_u128_unaligned(seq + (qi >> 1));
	uint64_t const x = nibble_linear(v.x), y = nibble_linear(v.y);
	return (qi & 1) ? ((x >> 4) | (y << 60)) : x;
}

// 16 consecutive text codes starting at text base ri (reads 16 bytes at text4 + (ri + TEXT4_FRONT) / 2).
__device__ __forceinline__ uint64_t load_text16(uint8_t const *text4, uint32_t ri)
{
	uint64_t const ti = (uint64_t) ri + TEXT4_FRONT;
	u64x2 const v = load_u128_unaligned(text4 + (ti >> 1));
	return (ti & 1) ? ((v.x >> 4) | (v.y << 60)) : v.x;
}

// Mismatch bits of 16 bases.
__device__ __forceinline__ uint32_t mismatch16(uint64_t q, uint64_t t)
{
	return mismatch8((uint32_t) q, (uint32_t) t) | (mismatch8((uint32_t) (q >> 32), (uint32_t) (t >> 32)) << 8);
}

}
