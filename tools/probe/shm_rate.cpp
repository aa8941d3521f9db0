// shm_rate.cpp -- how fast can N threads fill / read ONE file on a memory-backed file system?  (development aid for pamd_pipe.cpp)
//   g++ -O2 -pthread tools/probe/shm_rate.cpp -o tools/probe/shm_rate && tools/probe/shm_rate /dev/shm/x 8
// write methods: pwrite; one shared mapping + memcpy; the same with MADV_POPULATE_WRITE per piece; fallocate first + mapping.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
	const char *path = argc > 1 ? argv[1] : "/dev/shm/shm_rate.tmp";
	const size_t gb = argc > 2 ? atoi(argv[2]) : 8, piece = 32u << 20, total = gb << 30, npieces = total / piece;
	char *src = (char *) malloc(piece);
	memset(src, 'A', piece);
	for (int method = 0; method < 4; method++) {
		for (int nt : { 1, 2, 4, 6, 8, 12 }) {
			int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
			char *map = nullptr;
			double t0 = now();
			if (method == 3 && fallocate(fd, 0, 0, total) != 0)
				perror("fallocate");
			const double t_fa = now() - t0;
			if (method >= 1) {
				if (method != 3 && ftruncate(fd, total) != 0)
					perror("ftruncate");
				map = (char *) mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			}
			std::atomic<size_t> next{0};
			std::vector<std::thread> th;
			for (int t = 0; t < nt; t++)
				th.emplace_back([&] {
					for (;;) {
						const size_t i = next++;
						if (i >= npieces)
							return;
						if (method == 0) {
							if (pwrite(fd, src, piece, i * piece) != (ssize_t) piece)
								perror("pwrite");
						} else {
							if (method == 2)
								madvise(map + i * piece, piece, MADV_POPULATE_WRITE);
							memcpy(map + i * piece, src, piece);
						}
					}
				});
			for (auto &t : th)
				t.join();
			const double dt = now() - t0;
			if (map)
				munmap(map, total);
			close(fd);
			const char *names[] = { "pwrite", "mapping + memcpy", "mapping + populate + memcpy", "fallocate, mapping + memcpy" };
			printf("write %-30s %2d threads: %6.2f GB/s%s\n", names[method], nt, total / dt / 1e9, method == 3 ? (t_fa > 0 ? "  (fallocate included)" : "") : "");
			fflush(stdout);
		}
	}
	// read the file back (it exists now): pread into a private buffer per thread
	for (int nt : { 1, 2, 4, 6, 8, 12 }) {
		int fd = open(path, O_RDONLY);
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		const double t0 = now();
		for (int t = 0; t < nt; t++)
			th.emplace_back([&] {
				char *dst = (char *) malloc(piece);
				memset(dst, 0, piece);
				for (;;) {
					const size_t i = next++;
					if (i >= npieces)
						break;
					if (pread(fd, dst, piece, i * piece) != (ssize_t) piece)
						perror("pread");
				}
				free(dst);
			});
		for (auto &t : th)
			t.join();
		printf("read  pread                          %2d threads: %6.2f GB/s\n", nt, total / (now() - t0) / 1e9);
		close(fd);
	}
	unlink(path);
	return 0;
}
