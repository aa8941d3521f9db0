// pamd_encode.hip -- log-probabilities as table indices, for the trip to the host.
//
// Every per-base log-p the assemble kernels write is an entry of the assembler's table image (qual_score, the scorer's
// match / mismatch matrices, qual_nn; panda_result.p as algo_*.c match_probability and assembler.c:162-243 produce it), so
// a row of f64 can cross PCIe as u16 indices -- 2 bytes per base instead of 8 -- and be rebuilt bit for bit from the
// host's copy of the same image (pamd_table_image).  One pass over the arena: 8 bytes read, 2 written per base.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pamd_internal.h"

namespace {

constexpr unsigned SLOTS = PAMD_ENCODE_SLOTS;

__device__ __forceinline__ unsigned slot_of(unsigned long long bits) {
	return (unsigned) ((bits * 0x9E3779B97F4A7C15ull) >> 50) & (SLOTS - 1u);
}

__global__ __launch_bounds__(256) void pamd_encode_kernel(const double *p, uint16_t *idx, size_t n, const double *image, unsigned n_image, const uint16_t *slots) {
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	double *tab = reinterpret_cast<double *>(lds);
	uint16_t *sl = reinterpret_cast<uint16_t *>(lds + (size_t) ((n_image + 1u) & ~1u) * 8u);
	for (unsigned i = threadIdx.x; i < n_image; i += blockDim.x)
		tab[i] = image[i];
	for (unsigned i = threadIdx.x; i < SLOTS; i += blockDim.x)
		sl[i] = slots[i];
	__syncthreads();
	const size_t pairs = n / 2;	// two values per thread: one 16-byte load, one 4-byte store
	for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < pairs; j += (size_t) gridDim.x * blockDim.x) {
		const double2 v = reinterpret_cast<const double2 *>(p)[j];
		uint16_t out[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const double x = k == 0 ? v.x : v.y;
			const unsigned long long bits = (unsigned long long) __double_as_longlong(x);
			unsigned h = slot_of(bits);
			uint16_t found = 0xFFFFu;	// what lies between rows (never written) is no table entry
			for (int tries = 0; tries < PAMD_ENCODE_PROBES; tries++) {
				const uint16_t s = sl[h];
				if (s == 0xFFFFu)
					break;
				if ((unsigned long long) __double_as_longlong(tab[s]) == bits) {
					found = s;
					break;
				}
				h = (h + 1u) & (SLOTS - 1u);
			}
			out[k] = found;
		}
		reinterpret_cast<uint32_t *>(idx)[j] = (uint32_t) out[0] | ((uint32_t) out[1] << 16);
	}
}

}	// namespace

int pamd_launch_encode(const double *p, uint16_t *idx, size_t n_bases, const double *d_image, unsigned n_image, const uint16_t *d_slots, void *stream) {
	if (n_bases == 0)
		return PAMD_OK;
	const size_t lds = (size_t) ((n_image + 1u) & ~1u) * 8u + (size_t) SLOTS * 2u;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pamd_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	if (e != hipSuccess) {
		pamd_set_error("hipFuncSetAttribute(encode kernel): %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	size_t blocks = (n_bases / 2 + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	if (blocks == 0)
		blocks = 1;
	hipLaunchKernelGGL(pamd_encode_kernel, dim3((unsigned) blocks), dim3(256), lds, static_cast<hipStream_t>(stream), p, idx, (n_bases + 1) & ~(size_t) 1, d_image, n_image, d_slots);
	e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("encode kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
