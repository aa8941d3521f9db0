// development probe: what the DPP wave shifts do on this GPU
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
	const int lane = threadIdx.x;
	int v = lane + 100;
	out[lane] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);        // wave_shr:1
	out[64 + lane] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);   // wave_shl:1
	out[128 + lane] = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
	out[192 + lane] = __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);  // row_shl:1
	out[256 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);        // wave_shr:1 no bound ctrl
	out[320 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xF, 0xF, false);        // wave_ror:1
	out[384 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xF, 0xF, false);        // wave_rol:1
}
int main() {
	int *d, h[448];
	hipMalloc(&d, sizeof h);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *names[7] = {"wave_shr:1 bc", "wave_shl:1 bc", "row_shr:1 bc", "row_shl:1 bc", "wave_shr:1 old=-1", "wave_ror:1", "wave_rol:1"};
	for (int t = 0; t < 7; t++) {
		printf("%s:", names[t]);
		for (int i = 0; i < 64; i++)
			printf(" %d", h[64 * t + i]);
		printf("\n");
	}
	return 0;
}
