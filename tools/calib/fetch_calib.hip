// fetch_calib.hip — calibrates rocprofv3 FETCH_SIZE on this code base's access patterns (MI355X_MICROARCH.md §HBM:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   k_calib_block64   every lane reads one random, 64-byte aligned block with four 16-byte loads (k_project_fast's index reads)
//   k_calib_stream    wide coalesced 16 B / lane stream (the guide's reference pattern, expected ratio 0.5)
//   k_calib_row160    every lane reads 160 contiguous bytes at a random 1-byte-aligned offset with ten 16-byte loads (byte text reads)
//   k_calib_row80coop five neighbouring lanes read 80 contiguous bytes at a random 1-byte-aligned offset, 16 bytes each (the packed
//                     text reads of k_project_fast's cooperative compare; 12 rows per wave instruction)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;

__device__ inline uint64_t mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

__global__ void k_calib_block64(uint8_t const *buf, uint64_t n_blocks, uint64_t per_thread, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0;
	for (uint64_t i = 0; i < per_thread; ++i)
	{
		uint64_t const b = mix(tid * per_thread + i) % n_blocks;
		u64x2 const *p = (u64x2 const *) (buf + b * 64);
		u64x2 const a = p[0], c = p[1], d = p[2], e = p[3];
		acc += a.x ^ a.y ^ c.x ^ c.y ^ d.x ^ d.y ^ e.x ^ e.y;
	}
	if (acc == 0x1234567) sink[0] = acc;
}

__global__ void k_calib_stream(u64x2 const *buf, uint64_t n_vec, uint64_t *sink)
{
	uint64_t const stride = (uint64_t) gridDim.x * blockDim.x;
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride)
	{
		u64x2 const v = buf[i];
		acc += v.x ^ v.y;
	}
	if (acc == 0x1234567) sink[0] = acc;
}

__global__ void k_calib_row160(uint8_t const *buf, uint64_t n_bytes, uint64_t per_thread, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0;
	for (uint64_t i = 0; i < per_thread; ++i)
	{
		uint64_t const off = mix(tid * per_thread + i + 77) % (n_bytes - 256);
#pragma unroll
		for (int k = 0; k < 10; ++k)
		{
			u64x2 v;
			__builtin_memcpy(&v, buf + off + 16 * k, 16);
			acc += v.x ^ v.y;
		}
	}
	if (acc == 0x1234567) sink[0] = acc;
}

__global__ void k_calib_row80coop(uint8_t const *buf, uint64_t n_bytes, uint64_t per_wave, uint64_t *sink)
{
	uint64_t const tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t const wave = tid >> 6;
	uint32_t const lane = threadIdx.x & 63;
	uint32_t const sub = lane / 5, piece = lane % 5;		// 12 rows per instruction, lanes 60..63 idle
	uint64_t acc = 0;
	for (uint64_t i = 0; i < per_wave; i += 12)
	{
		uint64_t const off = mix(wave * per_wave + i + sub + 991) % (n_bytes - 256);
		if (sub < 12)
		{
			u64x2 v;
			__builtin_memcpy(&v, buf + off + 16 * piece, 16);
			acc += v.x ^ v.y;
		}
	}
	if (acc == 0x1234567) sink[0] = acc;
}

int main()
{
	uint64_t const bytes = 4ull << 30;		// 4 GiB: far beyond L2 and the 256 MiB Infinity Cache
	uint8_t *buf; uint64_t *sink;
	hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
	hipMemset(buf, 1, bytes);
	uint64_t const threads = 256ull * 4096;
	uint64_t const per_thread = 64;
	hipLaunchKernelGGL(k_calib_block64, dim3(4096), dim3(256), 0, 0, buf, bytes / 64, per_thread, sink);
	hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(256), 0, 0, (u64x2 const *) buf, bytes / 16, sink);
	hipLaunchKernelGGL(k_calib_row160, dim3(4096), dim3(256), 0, 0, buf, bytes, per_thread / 4, sink);
	hipLaunchKernelGGL(k_calib_row80coop, dim3(4096), dim3(256), 0, 0, buf, bytes, (uint64_t) 12 * 128, sink);
	hipDeviceSynchronize();
	printf("known_bytes k_calib_block64 %llu\n", (unsigned long long) (threads * per_thread * 64));
	printf("known_bytes k_calib_stream %llu\n", (unsigned long long) bytes);
	printf("known_bytes k_calib_row160 %llu (useful; touches 3-4 64-byte lines = 192-256 bytes per row)\n", (unsigned long long) (threads * (per_thread / 4) * 160));
	printf("known_bytes k_calib_row80coop %llu (useful; an 80-byte row touches 2.23 64-byte lines on average = 143 bytes)\n", (unsigned long long) (threads / 64 * 12 * 128 * 80));
	return 0;
}
