// wg_placement.hip -- where do the waves of co-resident workgroups land?  (development aid, GPU box only)
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe/wg_placement.hip -o tools/probe/wg_placement
//   tools/probe/wg_placement <waves per workgroup> <LDS bytes per workgroup> <workgroups per CU wanted>
//
// Every wave records its hardware id (XCC, SE, CU, SIMD: s_getreg HW_ID / XCC_ID) and the clock when it starts and ends, and
// holds its slot for a while so that everything that CAN be resident at once is.  The host then counts, per CU, how many
// waves were in flight at the time the first workgroup of the CU was half done, and how they were spread over the four
// SIMDs.  VGPRS: the kernel's vector register allocation (compile-time, -DVLAST=95).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#ifndef VLAST	/* the last vector register of the wanted allocation: -DVLAST=95 for 96 registers */
#define VLAST 95
#endif
#define VGPRS (VLAST + 1)
#define STR2(x) #x
#define STR(x) STR2(x)

struct Rec {
	unsigned hw_id, xcc_id;
	unsigned long long t0, t1;
};

__global__ __launch_bounds__(1024) void probe(Rec *out, unsigned long long hold) {
	extern __shared__ unsigned char lds[];
	// touch the last register of the wanted allocation so that the kernel is given VGPRS registers
	asm volatile("v_mov_b32 v" STR(VLAST) ", 0" ::: "v" STR(VLAST));
	const unsigned long long t0 = __builtin_readcyclecounter();
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	lds[threadIdx.x] = (unsigned char) hw;
	while (__builtin_readcyclecounter() - t0 < hold)
		__builtin_amdgcn_s_sleep(8);
	const unsigned long long t1 = __builtin_readcyclecounter();
	if ((threadIdx.x & 63) == 0) {
		Rec r = { hw, xcc, t0, t1 };
		out[(size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
	}
}

int main(int argc, char **argv) {
	const int waves = argc > 1 ? atoi(argv[1]) : 10;
	const int ldsb = argc > 2 ? atoi(argv[2]) : 78576;
	const int per_cu = argc > 3 ? atoi(argv[3]) : 2;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const int wgs = cus * per_cu;
	Rec *d;
	hipMalloc(&d, sizeof(Rec) * (size_t) wgs * waves);
	hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
	hipLaunchKernelGGL(probe, dim3(wgs), dim3(waves * 64), ldsb, 0, d, 20000000ull);	// 20 M clocks of s_memtime/100 MHz-ish counter... whatever the unit, long against launch skew
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) {
		printf("launch failed: %s\n", hipGetErrorString(e));
		return 1;
	}
	std::vector<Rec> h((size_t) wgs * waves);
	hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
	// HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: se_id 3 bits)
	std::map<unsigned, std::vector<const Rec *>> by_cu;
	for (const Rec &r : h) {
		const unsigned cu = (r.hw_id >> 8) & 0xF, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 0xF;
		by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(&r);
	}
	std::map<std::vector<int>, int> patterns;
	std::map<int, int> resident_hist;
	for (auto &kv : by_cu) {
		// the instant at which the earliest wave of this CU is half done
		unsigned long long first = ~0ull, its_end = 0;
		for (const Rec *r : kv.second)
			if (r->t0 < first) {
				first = r->t0;
				its_end = r->t1;
			}
		const unsigned long long at = first + (its_end - first) / 2;
		std::vector<int> simd(4, 0);
		int n = 0;
		for (const Rec *r : kv.second)
			if (r->t0 <= at && at < r->t1) {
				simd[(r->hw_id >> 4) & 3]++;
				n++;
			}
		std::sort(simd.begin(), simd.end());
		patterns[simd]++;
		resident_hist[n]++;
	}
	printf("device: %d CUs seen as %zu distinct (xcc, se, sh, cu); %d workgroups of %d waves, %d B of LDS each, %d VGPRs\n", cus, by_cu.size(), wgs, waves, ldsb, VGPRS);
	for (auto &kv : resident_hist)
		printf("  %3d waves resident at once: %d CUs\n", kv.first, kv.second);
	for (auto &kv : patterns)
		printf("  waves per SIMD (sorted) %d %d %d %d: %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
	return 0;
}
