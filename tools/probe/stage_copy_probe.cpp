// How fast can one host thread copy reads into pinned staging rows?  memcpy against AVX2 streaming stores.
//   hipcc -O2 -mavx2 tools/probe/stage_copy_probe.cpp -o /tmp/scp && /tmp/scp
#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void nt_copy(char *dst, const char *src, size_t n) {
	size_t i = 0;
	for (; i + 64 <= n; i += 64) {
		const __m256i a = _mm256_loadu_si256((const __m256i *) (src + i)), b = _mm256_loadu_si256((const __m256i *) (src + i + 32));
		_mm256_stream_si256((__m256i *) (dst + i), a);
		_mm256_stream_si256((__m256i *) (dst + i + 32), b);
	}
	memcpy(dst + i, src + i, n - i);
}
int main() {
	const size_t rows = 4 * 65536, stride = 512, len = 500, pool = 8192;
	char *src = (char *) malloc(pool * 2 * len);
	memset(src, 7, pool * 2 * len);
	for (int kind = 0; kind < 3; kind++) {
		char *f = nullptr, *r = nullptr;
		if (kind == 0) { f = (char *) aligned_alloc(4096, rows * stride); r = (char *) aligned_alloc(4096, rows * stride); }
		else { unsigned fl = kind == 1 ? hipHostMallocNonCoherent : hipHostMallocDefault; if (hipHostMalloc((void **) &f, rows * stride, fl) != hipSuccess || hipHostMalloc((void **) &r, rows * stride, fl) != hipSuccess) return 1; }
		memset(f, 0, rows * stride); memset(r, 0, rows * stride);
		for (int how = 0; how < 2; how++) {
			double best = 1e9;
			for (int rep = 0; rep < 3; rep++) {
				const double t0 = now();
				for (size_t k = 0; k < rows; k++) {
					const char *s = src + (k % pool) * 2 * len;
					if (how == 0) { memcpy(f + k * stride, s, len); memcpy(r + k * stride, s + len, len); }
					else { nt_copy(f + k * stride, s, len); nt_copy(r + k * stride, s + len, len); }
				}
				_mm_sfence();
				const double t = now() - t0;
				best = t < best ? t : best;
			}
			printf("%s  %s: %.1f ns per pair\n", kind == 0 ? "malloc        " : kind == 1 ? "pinned noncoh " : "pinned default", how == 0 ? "memcpy" : "avx2 nt", best / rows * 1e9);
		}
	}
	return 0;
}
