// pamd_pipe.cpp -- pamd_pipe_run(): two FASTQ files in, one FASTA / FASTQ file out, at the rate the PCIe link allows.
//
// The reference moves its text with one reader thread per input behind a queue (async.c:103-149), worker threads that pull
// pairs from a mux (pool.c:71-108, mux.c:132-161) and a writer thread per output (writer.c).  Here the workers are the three
// device stages (pamd_fastq_parse_device -> pamd_assemble_device -> pamd_format_device), and the host's job is to keep both
// directions of the link busy while they run:
//
//   reader threads   pread() blocks of each input file into a ring of PINNED blocks           (async.c's reader, per block)
//                    -- a gzip or bzip2 file, or an input that is not a regular file (a FIFO), has ONE reader instead, which
//                    reads it in order through zlib / libbz2, as the reference opens every input (fileio.c:70-96); its size is
//                    known when it ends
//   copier threads   one per input: hipMemcpyAsync of each block into a device ring, on a stream of its own
//   the pump         (the calling thread) gathers a chunk's text from the two device rings behind the previous chunk's
//                    unconsumed tail (device-to-device: records straddle blocks and the two files' records differ in size),
//                    runs the three stages on one stream, sends the text back in pieces on a third stream
//   writer threads   pwrite() each piece at its offset of the output file                      (writer.c's thread)
//
// Nothing here scores, merges, parses or formats: bytes are moved, stages are called.  No CPU fallback: without the device the
// stages fail and so does this.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "pamd_internal.h"
#include "pamd_bz2.h"
#include "../../include/pandaseq_amd_io.h"

namespace {

double now_s() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Shared {	// what every thread of one run looks at
	std::atomic<bool> failed{false};
	std::mutex msg_lock;
	std::string msg;
	int code = PAMD_OK;	// what the run returns: PAMD_EIO for a file that could not be read or written, PAMD_ENODEVICE for the HIP runtime
	int device = 0;
	std::function<void()> wake;	// notifies every condition variable of the run, each under its mutex: a waiter that has just looked at
					// `failed` and not yet gone to sleep holds that mutex, so it cannot miss the notification
	void fail(int rc, const char *what, const char *why) {
		{
			std::lock_guard<std::mutex> g(msg_lock);
			if (!failed.exchange(true)) {
				msg = std::string(what) + ": " + why;
				code = rc;
			}
		}
		if (wake)
			wake();
	}
};

// ---- one input file: pinned blocks filled by reader threads, copied into a device ring by a copier thread ---------------
struct Input {
	Shared *sh = nullptr;
	int fd = -1;
	uint64_t size = 0, block = 0, n_blocks = 0;
	uint64_t limit = 0;		// blocks that will ever be read: n_blocks, or fewer once the pump has stopped early (under `m`)
	static constexpr int NB = 6;	// pinned blocks
	int nr = 8;			// device ring slots
	char *pin[NB] = { nullptr };
	char *ring = nullptr;		// device, nr * block bytes
	hipStream_t s_in = nullptr;
	hipEvent_t ev[2] = { nullptr, nullptr };
	std::mutex m;
	std::condition_variable cv;
	uint64_t next_read = 0;		// next block a reader thread takes
	uint64_t filled[NB];		// which block each pinned slot holds (~0: none)
	uint64_t pin_free_upto = 0;	// blocks below this have left their pinned slot
	uint64_t on_device = 0;		// blocks below this are in the device ring, their copies complete
	uint64_t consumed = 0;		// bytes of the file the pump no longer needs in the ring (published after the gathers that read them are done)
	std::vector<std::thread> readers;
	std::thread copier;

	gzFile gz = nullptr;		// a streamed input (gzip text, or not a regular file): read in order by one thread; `size` is ~0 until it ends
	void *bz = nullptr;		// ... or bzip2 text (fileio.c:75-85)
	bool streamed() const {
		return gz != nullptr || bz != nullptr;
	}
	// up to `want` bytes of the stream; 0 = it has ended; < 0 = failed (message set)
	long stream_read(char *dst, uint64_t want) {
		const unsigned ask = (unsigned) std::min<uint64_t>(want, 1u << 30);
		if (bz != nullptr) {
			if (bz_ended)
				return 0;	// (fileio.c:48-58 asks once more and takes the BZ_SEQUENCE_ERROR for the end: further streams of the file are not read there either)
			int e = 0;
			const int r = pamd_bz2()->bzRead(&e, bz, dst, (int) ask);
			bz_ended = e == PAMD_BZ_STREAM_END;
			if (e != PAMD_BZ_OK && e != PAMD_BZ_STREAM_END) {
				sh->fail(PAMD_EIO, "BZ2_bzRead", ("libbz2 error " + std::to_string(e)).c_str());
				return -1;
			}
			return r;
		}
		const int r = gzread(gz, dst, ask);
		if (r < 0) {
			int e = 0;
			const char *why = gzerror(gz, &e);
			sh->fail(PAMD_EIO, "gzread", e == Z_ERRNO ? strerror(errno) : why);
		}
		return r;
	}
	bool bz_ended = false;

	uint64_t block_len(uint64_t b) const {
		const uint64_t at = b * block;
		return at >= size ? 0 : std::min<uint64_t>(block, size - at);
	}
	uint64_t size_now() {
		std::lock_guard<std::mutex> g(m);
		return size;
	}
	void stream_main() {	// fileio.c:31-46's gzread(), a block at a time
		for (uint64_t b = 0;; b++) {
			{
				std::unique_lock<std::mutex> g(m);
				if (b >= limit)
					return;
				next_read = b + 1;
				cv.wait(g, [&] { return sh->failed.load() || pin_free_upto + NB > b; });
				if (sh->failed.load())
					return;
			}
			char *dst = pin[b % NB];
			uint64_t got = 0;
			while (got < block) {
				const long r = stream_read(dst + got, block - got);
				if (r < 0) {
					cv.notify_all();
					return;
				}
				if (r == 0)
					break;
				got += (uint64_t) r;
			}
			{
				std::lock_guard<std::mutex> g(m);
				if (got < block) {	// the input ends in this block
					size = b * block + got;
					n_blocks = got > 0 ? b + 1 : b;
					limit = std::min(limit, n_blocks);
				}
				if (got > 0)
					filled[b % NB] = b;
			}
			cv.notify_all();
			if (got < block)
				return;
		}
	}
	void reader_main() {
		for (;;) {
			uint64_t b;
			{
				std::unique_lock<std::mutex> g(m);
				if (next_read >= limit)
					return;
				b = next_read++;
				cv.wait(g, [&] { return sh->failed.load() || pin_free_upto + NB > b; });	// its pinned slot's previous block has gone up
				if (sh->failed.load())
					return;
			}
			const uint64_t len = block_len(b);
			char *dst = pin[b % NB];
			uint64_t got = 0;
			while (got < len) {
				const ssize_t r = pread(fd, dst + got, len - got, (off_t) (b * block + got));
				if (r <= 0) {
					sh->fail(PAMD_EIO, "pread", r == 0 ? "the file shrank" : strerror(errno));
					cv.notify_all();
					return;
				}
				got += (uint64_t) r;
			}
			{
				std::lock_guard<std::mutex> g(m);
				filled[b % NB] = b;
			}
			cv.notify_all();
		}
	}
	void copier_main() {
		if (hipSetDevice(sh->device) != hipSuccess) {
			sh->fail(PAMD_ENODEVICE, "hipSetDevice", "copier thread");
			cv.notify_all();
			return;
		}
		// block b is issued, then block b-1 is waited for: one copy is always queued behind the one in flight
		for (uint64_t b = 0;; b++) {
			bool issue;
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&] { return sh->failed.load() || b >= limit || (filled[b % NB] == b && consumed >= (b + 1) * block - std::min<uint64_t>((b + 1) * block, (uint64_t) nr * block)); });	// read, and its ring slot's previous bytes gathered
				if (sh->failed.load())
					return;
				issue = b < limit;
			}
			if (issue && (hipMemcpyAsync(ring + (b % (uint64_t) nr) * block, pin[b % NB], block_len(b), hipMemcpyHostToDevice, s_in) != hipSuccess || hipEventRecord(ev[b & 1], s_in) != hipSuccess)) {
				sh->fail(PAMD_ENODEVICE, "hipMemcpyAsync", "host to device");
				cv.notify_all();
				return;
			}
			if (b > 0) {
				if (hipEventSynchronize(ev[(b - 1) & 1]) != hipSuccess) {
					sh->fail(PAMD_ENODEVICE, "hipEventSynchronize", "host to device");
					cv.notify_all();
					return;
				}
				{
					std::lock_guard<std::mutex> g(m);
					pin_free_upto = b;
					on_device = b;
				}
				cv.notify_all();
			}
			if (!issue)
				return;
		}
	}
	// the pump: wait until `want` bytes from file position `pos` are in the ring (or the file ends); returns how many are
	uint64_t wait_available(uint64_t pos, uint64_t want) {
		std::unique_lock<std::mutex> g(m);
		uint64_t have = 0;
		cv.wait(g, [&] {
			const uint64_t upto = std::min<uint64_t>(on_device * block, size);
			have = upto > pos ? upto - pos : 0;
			return sh->failed.load() || have >= want || on_device >= limit;
		});
		return std::min(have, want);
	}
	void publish_consumed(uint64_t pos) {
		{
			std::lock_guard<std::mutex> g(m);
			consumed = pos;
		}
		cv.notify_all();
	}
};

// ---- the output file: pinned pieces written by writer threads ---------------------------------------------------------------
struct Output {
	Shared *sh = nullptr;
	int fd = -1;
	static constexpr int NP = 6;
	bool seekable = true;
	uint64_t piece = 0;
	char *pin[NP] = { nullptr };
	hipEvent_t ev[NP] = { nullptr };
	struct Job {
		int slot;
		uint64_t off, len;
	};
	std::mutex m;
	std::condition_variable cv;
	std::deque<Job> jobs;
	bool busy[NP] = { false };
	bool closing = false;
	std::vector<std::thread> writers;

	int take_slot() {	// the pump: a pinned piece nobody is writing from
		std::unique_lock<std::mutex> g(m);
		int s = -1;
		cv.wait(g, [&] {
			if (sh->failed.load())
				return true;
			for (int k = 0; k < NP; k++)
				if (!busy[k]) {
					s = k;
					return true;
				}
			return false;
		});
		if (s >= 0)
			busy[s] = true;
		return s;
	}
	void submit(int slot, uint64_t off, uint64_t len) {
		{
			std::lock_guard<std::mutex> g(m);
			jobs.push_back({ slot, off, len });
		}
		cv.notify_all();
	}
	void writer_main() {
		if (hipSetDevice(sh->device) != hipSuccess) {
			sh->fail(PAMD_ENODEVICE, "hipSetDevice", "writer thread");
			return;
		}
		for (;;) {
			Job j;
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&] { return closing || !jobs.empty() || sh->failed.load(); });
				if (jobs.empty())
					return;
				j = jobs.front();
				jobs.pop_front();
			}
			if (hipEventSynchronize(ev[j.slot]) != hipSuccess)
				sh->fail(PAMD_ENODEVICE, "hipEventSynchronize", "device to host");
			// (pwrite, and by default from ONE thread: write() holds the inode's lock for the whole copy, so threads writing one file
			// take turns, and on a memory-backed file system -- where making the file's pages is the cost -- one thread's 6.5 GB/s is
			// what the file takes however it is written: shared mappings filled by 1-12 threads, with or without MADV_POPULATE_WRITE,
			// 3.4-5.4 GB/s; fallocate first, then mappings, 7 GB/s; tools/probe/shm_rate.cpp, profiles/r05_pipe_shm_rate.txt)
			uint64_t put = 0;
			while (put < j.len && !sh->failed.load()) {
				// (an output that cannot seek -- a pipe, a FIFO: `| bzip2 > merged.fasta.bz2` -- has one writer, which takes the
				// pieces in the order they were queued: plain write())
				const ssize_t w = seekable ? pwrite(fd, pin[j.slot] + put, j.len - put, (off_t) (j.off + put)) : write(fd, pin[j.slot] + put, j.len - put);
				if (w <= 0) {
					sh->fail(PAMD_EIO, seekable ? "pwrite" : "write", strerror(errno));
					break;
				}
				put += (uint64_t) w;
			}
			{
				std::lock_guard<std::mutex> g(m);
				busy[j.slot] = false;
			}
			cv.notify_all();
		}
	}
	void drain_and_stop() {
		{
			std::unique_lock<std::mutex> g(m);
			closing = true;
		}
		cv.notify_all();
		for (std::thread &t : writers)
			t.join();
		writers.clear();
	}
};

struct DeviceBuffers {	// everything hipMalloc'ed / hipHostMalloc'ed by a run, freed in one place
	std::vector<void *> dev, host;
	~DeviceBuffers() {
		for (void *p : dev)
			(void) hipFree(p);
		for (void *p : host)
			(void) hipHostFree(p);
	}
	template <class T>
	bool dmalloc(T **p, size_t bytes) {
		if (hipMalloc(reinterpret_cast<void **>(p), bytes ? bytes : 16) != hipSuccess)
			return false;
		dev.push_back(*p);
		return true;
	}
	template <class T>
	bool hmalloc(T **p, size_t bytes) {
		if (hipHostMalloc(reinterpret_cast<void **>(p), bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess)
			return false;
		host.push_back(*p);
		return true;
	}
};

// gather `len` bytes of file position `pos` from the device ring (which wraps) to dst
bool gather(const Input &in, uint64_t pos, uint64_t len, char *dst, hipStream_t st) {
	const uint64_t cap = (uint64_t) in.nr * in.block;
	while (len > 0) {
		const uint64_t at = pos % cap, run = std::min<uint64_t>(len, cap - at);
		if (hipMemcpyAsync(dst, in.ring + at, run, hipMemcpyDeviceToDevice, st) != hipSuccess)
			return false;
		dst += run;
		pos += run;
		len -= run;
	}
	return true;
}

}	// namespace

extern "C" int pamd_pipe_run(pamd_assembler *a, int device, const pamd_pipe_options *o, pamd_pipe_stats *stats) {
	if (a == nullptr || o == nullptr || stats == nullptr || o->forward_path == nullptr || o->reverse_path == nullptr || o->out_path == nullptr) {
		pamd_set_error("pamd_pipe_run: null argument");
		return PAMD_EINVAL;
	}
	memset(stats, 0, sizeof *stats);
	const double t_start = now_s();
	const size_t chunk_pairs = o->pairs_per_chunk ? o->pairs_per_chunk : 1000000;
	const uint64_t text_cap = (uint64_t) chunk_pairs * 1400 + 65536;	// room for chunk_pairs records of up to ~1.4 KB
	if (text_cap >= (1ull << 32)) {
		pamd_set_error("pamd_pipe_run: at most 3 000 000 pairs per chunk (a text chunk must stay below 4 GiB)");
		return PAMD_EINVAL;
	}
	const uint64_t out_cap = (uint64_t) chunk_pairs * 2400 + 4096;
	const size_t stride = 456;
	Shared sh;
	sh.device = device;
	if (hipSetDevice(device) != hipSuccess) {
		pamd_set_error("pamd_pipe_run: no such device");
		return PAMD_ENODEVICE;
	}
	Input in[2];
	Output out;
	out.sh = &sh;
	sh.wake = [&in, &out]() {
		for (Input &s : in) {
			std::lock_guard<std::mutex> g(s.m);
			s.cv.notify_all();
		}
		std::lock_guard<std::mutex> g(out.m);
		out.cv.notify_all();
	};
	DeviceBuffers mem;
	const char *paths[2] = { o->forward_path, o->reverse_path };
	auto close_all = [&]() {
		for (Input &s : in) {
			if (s.gz)
				gzclose(s.gz);	// (closes the descriptor it was opened on)
			else if (s.bz)
				pamd_bz2()->bzclose(s.bz);	// (likewise)
			else if (s.fd >= 0)
				close(s.fd);
			s.gz = nullptr;
			s.bz = nullptr;
			s.fd = -1;
		}
		if (out.fd >= 0)
			close(out.fd);
	};
	for (int s = 0; s < 2; s++) {
		in[s].sh = &sh;
		in[s].fd = open(paths[s], O_RDONLY);
		struct stat sb;
		if (in[s].fd < 0 || fstat(in[s].fd, &sb) != 0) {
			pamd_set_error("pamd_pipe_run: cannot open %s", paths[s]);
			close_all();
			return PAMD_EINVAL;
		}
		in[s].size = (uint64_t) sb.st_size;
		unsigned char magic[2] = { 0, 0 };
		const bool regular = S_ISREG(sb.st_mode);
		if (regular && pread(in[s].fd, magic, 2, 0) == 2 && magic[0] == 'B' && magic[1] == 'Z') {	// fileio.c:75-85
			const PamdBz2 *bz = pamd_bz2();
			in[s].bz = bz ? bz->bzdopen(in[s].fd, "r") : nullptr;
			if (in[s].bz == nullptr) {
				pamd_set_error(bz ? "pamd_pipe_run: libbz2 cannot open %s" : "pamd_pipe_run: %s is bzip2 text and libbz2.so.1 cannot be loaded here", paths[s]);
				close_all();
				return PAMD_EINVAL;
			}
			in[s].size = ~0ull;
		} else if (!regular || (magic[0] == 0x1f && magic[1] == 0x8b)) {
			in[s].gz = gzdopen(in[s].fd, "rb");	// (reads plain text as it is: a FIFO need not carry gzip)
			if (in[s].gz == nullptr) {
				pamd_set_error("pamd_pipe_run: zlib cannot open %s", paths[s]);
				close_all();
				return PAMD_EINVAL;
			}
			gzbuffer(in[s].gz, 1u << 20);
			in[s].size = ~0ull;
		}
		// blocks of a quarter of a chunk's text (64 KiB .. 32 MiB); a device ring of eight: twice what a chunk may ask for
		uint64_t b = (text_cap / 4 + 65535) & ~65535ull;
		b = std::min<uint64_t>(std::max<uint64_t>(b, 65536), 32u << 20);
		in[s].block = b;
		in[s].nr = (int) ((2 * text_cap + b - 1) / b) + 1;
		in[s].n_blocks = in[s].limit = in[s].streamed() ? ~0ull : (in[s].size + b - 1) / b;
		for (uint64_t &f : in[s].filled)
			f = ~0ull;
	}
	out.fd = open(o->out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (out.fd < 0) {
		pamd_set_error("pamd_pipe_run: cannot create %s", o->out_path);
		close_all();
		return PAMD_EINVAL;
	}
	out.seekable = lseek(out.fd, 0, SEEK_CUR) != (off_t) -1;
	out.piece = std::min<uint64_t>(std::max<uint64_t>((out_cap / 4 + 65535) & ~65535ull, 65536), 32u << 20);

	pamd_fastq_parser *fq = pamd_fastq_parser_create(device, o->qualmin ? o->qualmin : 33, o->policy);
	pamd_formatter *fmt = pamd_formatter_create(device);
	hipStream_t s_cmp = nullptr, s_out = nullptr;
	char *d_text[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } }, *d_out[2] = { nullptr, nullptr };
	pamd_qual *d_fwd = nullptr, *d_rev = nullptr;
	uint16_t *d_flen = nullptr, *d_rlen = nullptr;
	pamd_seqid *d_ids = nullptr;
	pamd_pair_result *d_res = nullptr;
	uint8_t *d_nt = nullptr;
	double *d_p = nullptr;
	unsigned long long *d_cursor = nullptr;
	long long *d_ctr = nullptr;
	hipEvent_t ev_out_done[2] = { nullptr, nullptr };
	const size_t seq_cap = chunk_pairs * 912 + (size_t) 16384 * 4096;
	bool ok = fq != nullptr && fmt != nullptr && hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking) == hipSuccess;
	for (int s = 0; s < 2 && ok; s++) {
		ok = ok && mem.dmalloc(&in[s].ring, (size_t) in[s].nr * in[s].block) && mem.dmalloc(&d_text[s][0], text_cap + 16) && mem.dmalloc(&d_text[s][1], text_cap + 16);
		for (int k = 0; k < Input::NB && ok; k++)
			ok = mem.hmalloc(&in[s].pin[k], in[s].block);
		ok = ok && hipStreamCreateWithFlags(&in[s].s_in, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&in[s].ev[0], hipEventDisableTiming) == hipSuccess
		     && hipEventCreateWithFlags(&in[s].ev[1], hipEventDisableTiming) == hipSuccess;
	}
	for (int k = 0; k < Output::NP && ok; k++)
		ok = mem.hmalloc(&out.pin[k], out.piece) && hipEventCreateWithFlags(&out.ev[k], hipEventDisableTiming) == hipSuccess;
	ok = ok && mem.dmalloc(&d_out[0], out_cap) && mem.dmalloc(&d_out[1], out_cap) && mem.dmalloc(&d_fwd, chunk_pairs * stride * 2) && mem.dmalloc(&d_rev, chunk_pairs * stride * 2)
	     && mem.dmalloc(&d_flen, chunk_pairs * 2) && mem.dmalloc(&d_rlen, chunk_pairs * 2) && mem.dmalloc(&d_ids, chunk_pairs * sizeof(pamd_seqid)) && mem.dmalloc(&d_res, chunk_pairs * sizeof(pamd_pair_result))
	     && mem.dmalloc(&d_nt, seq_cap) && mem.dmalloc(&d_p, seq_cap * sizeof(double)) && mem.dmalloc(&d_cursor, 8) && mem.dmalloc(&d_ctr, PAMD_NUM_COUNTERS * 8)
	     && hipEventCreateWithFlags(&ev_out_done[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_out_done[1], hipEventDisableTiming) == hipSuccess
	     && hipMemset(d_ctr, 0, PAMD_NUM_COUNTERS * 8) == hipSuccess;
	auto destroy_all = [&]() {
		for (Input &s : in) {
			if (s.s_in)
				(void) hipStreamDestroy(s.s_in);
			for (hipEvent_t e : s.ev)
				if (e)
					(void) hipEventDestroy(e);
		}
		for (hipEvent_t e : out.ev)
			if (e)
				(void) hipEventDestroy(e);
		for (hipEvent_t e : ev_out_done)
			if (e)
				(void) hipEventDestroy(e);
		if (s_cmp)
			(void) hipStreamDestroy(s_cmp);
		if (s_out)
			(void) hipStreamDestroy(s_out);
		if (fq)
			pamd_fastq_parser_destroy(fq);
		if (fmt)
			pamd_formatter_destroy(fmt);
		close_all();
	};
	if (!ok) {
		if (fq != nullptr && fmt != nullptr)
			pamd_set_error("pamd_pipe_run: out of device or pinned memory for %zu pairs per chunk", chunk_pairs);
		destroy_all();
		return PAMD_ENODEVICE;
	}
	const int n_readers = o->reader_threads > 0 ? o->reader_threads : 3, n_writers = !out.seekable ? 1 : (o->writer_threads > 0 ? o->writer_threads : 1);
	for (Input &s : in) {
		if (s.streamed())
			s.readers.emplace_back(&Input::stream_main, &s);
		else
			for (int t = 0; t < n_readers; t++)
				s.readers.emplace_back(&Input::reader_main, &s);
		s.copier = std::thread(&Input::copier_main, &s);
	}
	for (int t = 0; t < n_writers; t++)
		out.writers.emplace_back(&Output::writer_main, &out);
	const double t_ready = now_s();

	// ---- the pump ----
	uint64_t pos[2] = { 0, 0 };		// file bytes gathered out of each ring so far
	uint64_t tail_off[2] = { 0, 0 }, tail_len[2] = { 0, 0 };	// the previous chunk's unconsumed text, in d_text[s][(k - 1) & 1]
	double bytes_per_record[2] = { 600.0, 600.0 };
	uint64_t out_off = 0, records = 0, dropped = 0, chunk = 0, parsed = 0;
	int stop = PAMD_FQ_END, rc = PAMD_OK;
	uint64_t stop_record = 0;
	bool grow = false;
	const bool trace = getenv("PAMD_PIPE_TRACE") != nullptr;	// per chunk on stderr: seconds waiting for input, in parse, in assemble + format, handing pieces to the writers
	while (!sh.failed.load()) {
		const int cur = (int) (chunk & 1), prev = cur ^ 1;
		const double c0 = now_s();
		double c1 = c0, c2 = c0, c3 = c0;
		uint64_t have[2];
		bool more_input = false;
		for (int s = 0; s < 2; s++) {
			uint64_t target = grow ? text_cap : std::min<uint64_t>(text_cap, (uint64_t) (bytes_per_record[s] * 1.03 * (double) chunk_pairs) + 65536);
			target &= ~15ull;
			const uint64_t want = target > tail_len[s] ? target - tail_len[s] : 0;
			const uint64_t take = in[s].wait_available(pos[s], want);
			if (sh.failed.load())
				break;
			// this chunk's text: the tail of the previous one, then `take` new bytes of the file
			if (tail_len[s] > 0 && hipMemcpyAsync(d_text[s][cur], d_text[s][prev] + tail_off[s], tail_len[s], hipMemcpyDeviceToDevice, s_cmp) != hipSuccess)
				sh.fail(PAMD_ENODEVICE, "hipMemcpyAsync", "tail");
			if (!gather(in[s], pos[s], take, d_text[s][cur] + tail_len[s], s_cmp))
				sh.fail(PAMD_ENODEVICE, "hipMemcpyAsync", "gather");
			pos[s] += take;
			have[s] = tail_len[s] + take;
			more_input |= pos[s] < in[s].size_now();
		}
		if (sh.failed.load())
			break;
		if (have[0] == 0 || have[1] == 0)
			break;	// one of the files is used up: what is left of the other has no mate (fastq.c:123,130)
		pamd_fastq_status st;
		c1 = now_s();
		rc = pamd_fastq_parse_device(fq, d_text[0][cur], have[0], d_text[1][cur], have[1], chunk_pairs, d_fwd, stride, d_rev, stride, d_flen, d_rlen, d_ids, &st, s_cmp);
		if (rc != PAMD_OK)
			break;
		parsed++;
		// (the parse synchronised its stream: the gathers are done, their ring bytes may be overwritten)
		for (int s = 0; s < 2; s++)
			in[s].publish_consumed(pos[s]);
		c2 = c3 = now_s();
		if (st.n_good > 0) {
			pamd_batch b;
			memset(&b, 0, sizeof b);
			b.n_pairs = st.n_good;
			b.fwd = d_fwd;
			b.fwd_stride = stride;
			b.rev = d_rev;
			b.rev_stride = stride;
			b.flen = d_flen;
			b.rlen = d_rlen;
			b.results = d_res;
			b.seq_nt = d_nt;
			b.seq_p = d_p;
			b.seq_stride = 0;
			b.seq_capacity = seq_cap;
			b.seq_cursor = reinterpret_cast<uint64_t *>(d_cursor);
			b.counters = reinterpret_cast<int64_t *>(d_ctr);
			if (hipMemsetAsync(d_cursor, 0, 8, s_cmp) != hipSuccess) {
				sh.fail(PAMD_ENODEVICE, "hipMemsetAsync", "cursor");
				break;
			}
			if ((rc = pamd_assemble_device(a, &b, s_cmp)) != PAMD_OK)
				break;
			// this text buffer's previous contents (two chunks ago) have left for the host
			if (hipStreamWaitEvent(s_cmp, ev_out_done[cur], 0) != hipSuccess) {
				sh.fail(PAMD_ENODEVICE, "hipStreamWaitEvent", "output buffer");
				break;
			}
			size_t text_bytes = 0, n_fix = 0;
			uint64_t fix[2 * 64];
			if ((rc = pamd_format_device(fmt, o->format, st.n_good, d_res, d_nt, d_p, d_ids, d_text[0][cur], d_out[cur], out_cap, &text_bytes, fix, 64, &n_fix, s_cmp)) != PAMD_OK)
				break;
			for (size_t k = 0; k < n_fix; k++) {	// the "%f" fields the device would not decide: re-printed as output.c:93 prints them (about one pair in 10^9)
				pamd_pair_result one;
				char tmp[32];
				if (hipMemcpy(&one, d_res + fix[2 * k + 1], sizeof one, hipMemcpyDeviceToHost) != hipSuccess) {
					sh.fail(PAMD_ENODEVICE, "hipMemcpy", "fixup");
					break;
				}
				snprintf(tmp, sizeof tmp, "%f", exp(one.quality));
				if (hipMemcpy(d_out[cur] + fix[2 * k], tmp, 8, hipMemcpyHostToDevice) != hipSuccess)
					sh.fail(PAMD_ENODEVICE, "hipMemcpy", "fixup");
			}
			c3 = now_s();
			// (the format synchronised its stream: the text is there) -- back in pieces, each written as soon as it has landed
			for (uint64_t at = 0; at < text_bytes && !sh.failed.load(); at += out.piece) {
				const uint64_t len = std::min<uint64_t>(out.piece, text_bytes - at);
				const int slot = out.take_slot();
				if (slot < 0)
					break;
				if (hipMemcpyAsync(out.pin[slot], d_out[cur] + at, len, hipMemcpyDeviceToHost, s_out) != hipSuccess || hipEventRecord(out.ev[slot], s_out) != hipSuccess) {
					sh.fail(PAMD_ENODEVICE, "hipMemcpyAsync", "device to host");
					break;
				}
				out.submit(slot, out_off + at, len);
			}
			if (hipEventRecord(ev_out_done[cur], s_out) != hipSuccess)
				sh.fail(PAMD_ENODEVICE, "hipEventRecord", "output buffer");
			out_off += text_bytes;
		}
		if (trace)
			fprintf(stderr, "pamd_pipe: chunk %llu: %llu pairs; input %.4f s, parse %.4f s, assemble + format %.4f s, output hand-off %.4f s\n", (unsigned long long) chunk, (unsigned long long) st.n_good, c1 - c0, c2 - c1, c3 - c2, now_s() - c3);
		records += st.n_good;
		dropped += st.n_dropped;
		if (st.stop_code != PAMD_FQ_END) {	// the reader gives up at this record, as stream_next_seq() does
			stop = st.stop_code;
			stop_record = records;
			break;
		}
		if (st.n_records == 0) {
			// not one whole record pair in this text: more of the files if there is more (a chunk's first guess may be short of a
			// long record); else both files end inside a record -- the headers are read and read_seq() runs out of lines:
			// PANDA_CODE_PREMATURE_EOF (fastq.c:58-60,69-71,84-86)
			if (!more_input) {
				stop = PAMD_FQ_EOF;
				stop_record = records;
				break;
			}
			if (grow) {
				sh.fail(PAMD_EINVAL, "input", "a record of more text than a chunk holds");
				break;
			}
			grow = true;
		} else {
			grow = false;
			bytes_per_record[0] = (double) st.fwd_consumed / (double) st.n_records;
			bytes_per_record[1] = (double) st.rev_consumed / (double) st.n_records;
		}
		tail_off[0] = st.fwd_consumed;
		tail_off[1] = st.rev_consumed;
		tail_len[0] = have[0] - st.fwd_consumed;
		tail_len[1] = have[1] - st.rev_consumed;
		chunk++;
		if (!more_input && tail_len[0] == 0 && tail_len[1] == 0)
			break;
	}
	// ---- wind down: nothing may be left in flight ----
	// readers and copiers may still be at work on input nobody will parse (the reader gave up at a record, a stage failed):
	// no further blocks are taken, every ring slot counts as free, and what is already read goes up and is dropped
	for (Input &s : in) {
		{
			std::lock_guard<std::mutex> g(s.m);
			s.limit = std::min(s.limit, s.next_read);
		}
		s.publish_consumed(~0ull);
		for (std::thread &t : s.readers)
			t.join();
		s.copier.join();
	}
	(void) hipStreamSynchronize(s_cmp);
	(void) hipStreamSynchronize(s_out);
	out.drain_and_stop();
	long long ctr[PAMD_NUM_COUNTERS];
	memset(ctr, 0, sizeof ctr);
	if (hipMemcpy(ctr, d_ctr, sizeof ctr, hipMemcpyDeviceToHost) != hipSuccess)
		sh.fail(PAMD_ENODEVICE, "hipMemcpy", "counters");
	const double t_end = now_s();
	for (unsigned k = 0; k < PAMD_NUM_COUNTERS; k++)
		stats->counters[k] = ctr[k];
	stats->pairs = records;
	stats->dropped = dropped;
	stats->chunks = parsed;
	stats->in_bytes = pos[0] + pos[1];
	stats->out_bytes = out_off;
	stats->stop_code = stop;
	stats->stop_record = stop_record;
	stats->seconds_setup = t_ready - t_start;
	stats->seconds_pump = t_end - t_ready;
	destroy_all();
	if (sh.failed.load()) {
		pamd_set_error("pamd_pipe_run: %s", sh.msg.c_str());
		return sh.code != PAMD_OK ? sh.code : PAMD_ENODEVICE;
	}
	return rc;
}
