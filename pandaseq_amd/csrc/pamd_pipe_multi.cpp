// pamd_pipe_multi.cpp -- pamd_pipe_run_multi(): two FASTQ files in, one FASTA / FASTQ file out, over N devices.
//
// The reference spreads a run over its workers with panda_run_pool (pool.c:110-181): every pthread has an assembler of its own,
// configured alike (panda_assembler_copy_configuration, pool.c:150), pulls pairs from the one mux (mux.c:132-161), hands its
// results to the one writer and prints a STAT block of its own at the end (pool.c:86-104).  Here a worker is a DEVICE with the
// three stages of pamd_pipe_run() on it, and what is dealt out is not a pair but a chunk of `pairs_per_chunk` records:
//
//   indexer threads   count the line ends of both files, a megabyte at a time (a FASTQ record is four lines of either file,
//                     fastq.c:46-104,107-197: whatever breaks that ends the stream there)
//   the cutter        turns the counts into chunk boundaries -- record k * R starts at this byte of the forward file and at that
//                     byte of the reverse file -- and deals chunk k to device k mod N: its byte ranges become the blocks that
//                     device's readers fetch
//   per device        reader threads pread() those blocks into pinned memory, a copier thread per input sends them up on a stream
//                     of its own, straight to where the parser wants the chunk's text (chunks are whole records: nothing is
//                     carried from one to the next, nothing is gathered); a pump thread runs parse -> assemble -> format and
//                     sends the text back in pieces.  Each device has its own PCIe link: this is where N devices buy throughput.
//   the ledger        chunk k's text lies behind the text of chunks 0 .. k-1 in the output: a pump publishes its chunk's size
//                     when the formatter is done and waits for the sizes before it.  A chunk in which the reader gives up
//                     (stream_next_seq() returning false, fastq.c:118-197) ends the run: later chunks, already dealt, are dropped.
//   writer threads    pwrite() the pieces at their offsets; one writer, pieces in offset order, when the output cannot seek
//
// The counters of every chunk come back with it and are summed on the host -- what the all-reduce of the STAT counters is when
// the N devices belong to one process (the one-process-per-GPU form sums them over RCCL: pandaseq_amd/stats.py, bench.py).
// pamd_stat_block() prints the sum as one worker's STAT block (pool.c:86-104, proxy.c:175-190).
//
// The bytes written and the counters are those of pamd_pipe_run() on the same files (tests/test_pipe_gpu.py holds the two
// to each other, devices = {0, 0} on a one-GPU box).  Inputs that have to be read in order through zlib / libbz2 or from a FIFO
// are one reader's work whatever the number of devices (0.7 M pairs/s): such a run goes to the first device's
// pamd_pipe_run().  No CPU fallback: without the devices the stages fail and so does this.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "pamd_internal.h"
#include "../../include/pandaseq_amd_io.h"

namespace {

double now_s() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Run;

// what every thread of one run looks at.  A failure is recorded once, under every lock a waiter may hold while it decides to
// sleep, and every condition variable is notified: no waiter misses it (wake_all is filled in by the run)
struct Shared {
	std::atomic<bool> failed{false};
	std::mutex msg_lock;
	std::string msg;
	int code = PAMD_OK;
	Run *run = nullptr;
	void fail(int rc, const char *what, const char *why);
};

// ---- line ends of one file, counted a segment at a time ------------------------------------------------------------------------
uint64_t count_nl_plain(const char *p, size_t n) {
	uint64_t c = 0;
	for (size_t i = 0; i < n; i++)
		c += p[i] == '\n';
	return c;
}
__attribute__((target("avx2"))) uint64_t count_nl_avx2(const char *p, size_t n) {
	uint64_t c = 0;
	for (size_t i = 0; i < n; i++)
		c += p[i] == '\n';
	return c;
}
uint64_t count_nl(const char *p, size_t n) {
	static const bool avx2 = __builtin_cpu_supports("avx2");
	return avx2 ? count_nl_avx2(p, n) : count_nl_plain(p, n);
}

struct FileIndex {
	Shared *sh = nullptr;
	int fd = -1;
	uint64_t size = 0;
	static constexpr uint64_t SEG = 1u << 20, BATCH = 8;	// counts per MiB; a counting thread reads eight at a time
	uint64_t n_seg = 0;
	std::vector<uint32_t> count;	// line ends in segment s
	std::vector<uint8_t> done;
	std::mutex m;
	std::condition_variable cv;
	uint64_t next_batch = 0;
	uint64_t ahead_of = 0;		// the cutter's segment: counting runs at most WINDOW segments ahead of it (bounded work for a run that stops early)
	static constexpr uint64_t WINDOW = 4096;
	std::vector<std::thread> threads;
	// the cutter's own state: lines before segment `cum_seg`
	uint64_t cum_seg = 0, cum_lines = 0;
	std::vector<char> scratch;

	void counter_main() {
		std::vector<char> buf(SEG * BATCH);
		for (;;) {
			uint64_t b;
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&] { return sh->failed.load() || next_batch * BATCH >= n_seg || next_batch * BATCH < ahead_of + WINDOW; });
				if (sh->failed.load() || next_batch * BATCH >= n_seg)
					return;
				b = next_batch++;
			}
			const uint64_t at = b * BATCH * SEG, len = std::min<uint64_t>(BATCH * SEG, size - at);
			uint64_t got = 0;
			while (got < len) {
				const ssize_t r = pread(fd, buf.data() + got, len - got, (off_t) (at + got));
				if (r <= 0) {
					sh->fail(PAMD_EIO, "pread", r == 0 ? "the file shrank" : strerror(errno));
					return;
				}
				got += (uint64_t) r;
			}
			uint32_t c[BATCH];
			const uint64_t segs = (len + SEG - 1) / SEG;
			for (uint64_t s = 0; s < segs; s++)
				c[s] = (uint32_t) count_nl(buf.data() + s * SEG, (size_t) std::min<uint64_t>(SEG, len - s * SEG));
			{
				std::lock_guard<std::mutex> g(m);
				for (uint64_t s = 0; s < segs; s++) {
					count[b * BATCH + s] = c[s];
					done[b * BATCH + s] = 1;
				}
			}
			cv.notify_all();
		}
	}
	// (the cutter) the byte at which line `line` starts (0-based: the byte behind the line-th line end); the file's size if it has
	// fewer lines.  Lines are asked for in ascending order (the walk over the segments only moves forwards).  false: the run failed
	bool locate(uint64_t line, uint64_t *at) {
		if (line == 0) {
			*at = 0;
			return true;
		}
		for (;;) {
			if (cum_seg >= n_seg) {
				*at = size;
				return true;
			}
			uint32_t c;
			{
				std::unique_lock<std::mutex> g(m);
				ahead_of = cum_seg;
				cv.notify_all();
				cv.wait(g, [&] { return sh->failed.load() || done[cum_seg]; });
				if (sh->failed.load())
					return false;
				c = count[cum_seg];
			}
			if (cum_lines + c < line) {	// the line end wanted lies behind this segment
				cum_lines += c;
				cum_seg++;
				continue;
			}
			// it is the (line - cum_lines)-th line end of this segment
			const uint64_t seg_at = cum_seg * SEG, len = std::min<uint64_t>(SEG, size - seg_at);
			scratch.resize(SEG);
			uint64_t got = 0;
			while (got < len) {
				const ssize_t r = pread(fd, scratch.data() + got, len - got, (off_t) (seg_at + got));
				if (r <= 0) {
					sh->fail(PAMD_EIO, "pread", r == 0 ? "the file shrank" : strerror(errno));
					return false;
				}
				got += (uint64_t) r;
			}
			uint64_t need = line - cum_lines;
			const char *p = scratch.data(), *end = p + len;
			while (need > 0) {
				const char *q = static_cast<const char *>(memchr(p, '\n', (size_t) (end - p)));
				if (q == nullptr) {	// (the count said it is here)
					sh->fail(PAMD_EIO, "index", "the file changed while it was read");
					return false;
				}
				p = q + 1;
				need--;
			}
			*at = seg_at + (uint64_t) (p - scratch.data());
			return true;	// (cum_* stay at this segment's start: the next line wanted may lie in it as well)
		}
	}
};

// ---- one input of one device: the blocks of its chunks, read into pinned memory and sent up ----------------------------------
struct Block {
	uint64_t file_off, len;
	uint64_t chunk;		// local chunk number (0, 1, .. on this device)
	uint64_t dst_off;	// where in the chunk's text
	bool last;		// the chunk's last block
};

struct DevInput {
	Shared *sh = nullptr;
	int fd = -1, device = 0;
	uint64_t block = 0;
	static constexpr int NB = 6;
	char *pin[NB] = { nullptr };
	char *text[2] = { nullptr, nullptr };	// device: the text of chunk i goes to text[i & 1]
	hipStream_t s_in = nullptr;
	hipEvent_t ev[2] = { nullptr, nullptr };
	std::mutex m;
	std::condition_variable cv;
	std::vector<Block> plan;	// appended by the cutter
	bool plan_final = false;	// nothing will be appended any more
	uint64_t next_read = 0;
	uint64_t filled[NB];
	uint64_t pin_free_upto = 0;
	uint64_t chunks_ready = 0;	// local chunks whose text is complete on the device
	uint64_t released = 0;		// local chunks the pump is done with (their text slot may be overwritten)
	std::vector<std::thread> readers;
	std::thread copier;

	void reader_main() {
		for (;;) {
			uint64_t b;
			Block bl;
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&] { return sh->failed.load() || next_read < plan.size() || plan_final; });
				if (sh->failed.load() || next_read >= plan.size())
					return;
				b = next_read++;
				bl = plan[b];
				cv.wait(g, [&] { return sh->failed.load() || pin_free_upto + NB > b; });
				if (sh->failed.load())
					return;
			}
			char *dst = pin[b % NB];
			uint64_t got = 0;
			while (got < bl.len) {
				const ssize_t r = pread(fd, dst + got, bl.len - got, (off_t) (bl.file_off + got));
				if (r <= 0) {
					sh->fail(PAMD_EIO, "pread", r == 0 ? "the file shrank" : strerror(errno));
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
		if (hipSetDevice(device) != hipSuccess) {
			sh->fail(PAMD_ENODEVICE, "hipSetDevice", "copier thread");
			return;
		}
		// block b is issued, then block b-1 is waited for: one copy is always queued behind the one in flight
		Block prev{};
		for (uint64_t b = 0;; b++) {
			bool issue;
			Block bl{};
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&] {
					if (sh->failed.load())
						return true;
					if (b < plan.size())
						return filled[b % NB] == b && released + 2 > plan[b].chunk;	// read, and the chunk's text slot is free
					return plan_final;
				});
				if (sh->failed.load())
					return;
				issue = b < plan.size();
				if (issue)
					bl = plan[b];
			}
			if (issue && bl.len > 0 && hipMemcpyAsync(text[bl.chunk & 1] + bl.dst_off, pin[b % NB], bl.len, hipMemcpyHostToDevice, s_in) != hipSuccess) {
				sh->fail(PAMD_ENODEVICE, "hipMemcpyAsync", "host to device");
				return;
			}
			if (issue && hipEventRecord(ev[b & 1], s_in) != hipSuccess) {
				sh->fail(PAMD_ENODEVICE, "hipEventRecord", "host to device");
				return;
			}
			if (b > 0) {
				if (hipEventSynchronize(ev[(b - 1) & 1]) != hipSuccess) {
					sh->fail(PAMD_ENODEVICE, "hipEventSynchronize", "host to device");
					return;
				}
				{
					std::lock_guard<std::mutex> g(m);
					pin_free_upto = b;
					if (prev.last)
						chunks_ready = prev.chunk + 1;
				}
				cv.notify_all();
			}
			if (!issue)
				return;
			prev = bl;
		}
	}
	// the pump: local chunk i is on the device (true), or there is no such chunk (false)
	bool wait_chunk(uint64_t i) {
		std::unique_lock<std::mutex> g(m);
		cv.wait(g, [&] { return sh->failed.load() || chunks_ready > i || (plan_final && (plan.empty() || plan.back().chunk < i)); });
		return !sh->failed.load() && chunks_ready > i;
	}
	void release(uint64_t n) {
		{
			std::lock_guard<std::mutex> g(m);
			released = n;
		}
		cv.notify_all();
	}
};

// ---- the output file: pinned pieces of every device, written by writer threads ---------------------------------------------------
struct Piece {
	char *pin = nullptr;
	hipEvent_t ev = nullptr;
	bool busy = false;
};

struct Output {
	Shared *sh = nullptr;
	int fd = -1;
	bool seekable = true;
	uint64_t piece = 0;
	struct Job {
		Piece *p;
		uint64_t off, len;
	};
	std::mutex m;
	std::condition_variable cv;
	std::deque<Job> jobs;
	uint64_t written_upto = 0;	// (an output that cannot seek: the next byte it takes)
	bool closing = false;
	std::vector<std::thread> writers;

	Piece *take(std::vector<Piece> &mine) {	// a pump: one of its pinned pieces nobody is writing from
		std::unique_lock<std::mutex> g(m);
		Piece *s = nullptr;
		cv.wait(g, [&] {
			if (sh->failed.load())
				return true;
			for (Piece &p : mine)
				if (!p.busy) {
					s = &p;
					return true;
				}
			return false;
		});
		if (s != nullptr)
			s->busy = true;
		return s;
	}
	void submit(Piece *p, uint64_t off, uint64_t len) {
		{
			std::lock_guard<std::mutex> g(m);
			jobs.push_back({ p, off, len });
		}
		cv.notify_all();
	}
	void writer_main() {
		for (;;) {
			Job j{};
			{
				std::unique_lock<std::mutex> g(m);
				auto pick = [&]() -> std::deque<Job>::iterator {
					if (seekable)
						return jobs.begin();
					for (auto it = jobs.begin(); it != jobs.end(); ++it)
						if (it->off == written_upto)
							return it;
					return jobs.end();
				};
				cv.wait(g, [&] { return sh->failed.load() || pick() != jobs.end() || (closing && jobs.empty()); });
				if (sh->failed.load())
					return;
				auto it = pick();
				if (it == jobs.end())
					return;	// closing, nothing left
				j = *it;
				jobs.erase(it);
			}
			if (hipEventSynchronize(j.p->ev) != hipSuccess)
				sh->fail(PAMD_ENODEVICE, "hipEventSynchronize", "device to host");
			uint64_t put = 0;
			while (put < j.len && !sh->failed.load()) {
				const ssize_t w = seekable ? pwrite(fd, j.p->pin + put, j.len - put, (off_t) (j.off + put)) : write(fd, j.p->pin + put, j.len - put);
				if (w <= 0) {
					sh->fail(PAMD_EIO, seekable ? "pwrite" : "write", strerror(errno));
					break;
				}
				put += (uint64_t) w;
			}
			{
				std::lock_guard<std::mutex> g(m);
				j.p->busy = false;
				written_upto = j.off + j.len;
			}
			cv.notify_all();
		}
	}
	void drain_and_stop() {
		{
			std::lock_guard<std::mutex> g(m);
			closing = true;
		}
		cv.notify_all();
		for (std::thread &t : writers)
			t.join();
		writers.clear();
	}
};

// ---- what the chunks came to, in order ------------------------------------------------------------------------------------------
struct ChunkResult {
	bool known = false;
	uint64_t records = 0, good = 0, dropped = 0, text_bytes = 0, in_bytes = 0;
	int stop_code = PAMD_FQ_END;
	bool left_fwd = false, left_rev = false;	// bytes were left over at the end of the chunk's forward / reverse text
	std::vector<long long> counters;
};

struct Ledger {
	Shared *sh = nullptr;
	std::mutex m;
	std::condition_variable cv;
	std::deque<ChunkResult> res;	// by global chunk number
	uint64_t prefix_upto = 0, prefix_bytes = 0;	// chunks below prefix_upto are known; their text sizes add up to prefix_bytes
	bool stopped = false;		// a chunk below prefix_upto ended the run
	uint64_t stop_chunk = 0;

	void publish(uint64_t k, ChunkResult &&r) {
		{
			std::lock_guard<std::mutex> g(m);
			if (res.size() <= k)
				res.resize(k + 1);
			res[k] = std::move(r);
			res[k].known = true;
			while (!stopped && prefix_upto < res.size() && res[prefix_upto].known) {
				const ChunkResult &c = res[prefix_upto];
				prefix_bytes += c.text_bytes;
				if (c.stop_code != PAMD_FQ_END) {
					stopped = true;
					stop_chunk = prefix_upto;
				}
				prefix_upto++;
			}
		}
		cv.notify_all();
	}
	// where chunk k's text goes; false: the run ended in an earlier chunk (or failed): chunk k is dropped
	bool wait_offset(uint64_t k, uint64_t *off) {
		std::unique_lock<std::mutex> g(m);
		cv.wait(g, [&] { return sh->failed.load() || stopped || prefix_upto >= k; });
		if (sh->failed.load() || (stopped && stop_chunk < k))
			return false;
		uint64_t o = 0;
		for (uint64_t c = 0; c < k; c++)
			o += res[c].text_bytes;
		*off = o;
		return true;
	}
	bool run_over(uint64_t k) {	// (a pump, before it spends time on chunk k)
		std::lock_guard<std::mutex> g(m);
		return stopped && stop_chunk < k;
	}
};

struct DeviceBuffers {	// everything hipMalloc'ed / hipHostMalloc'ed on one device by a run, freed in one place
	int device = 0;
	std::vector<void *> dev, host;
	~DeviceBuffers() {
		(void) hipSetDevice(device);
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
		if (hipHostMalloc(reinterpret_cast<void **>(p), bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess)
			return false;
		host.push_back(*p);
		return true;
	}
};

// ---- one device's three stages ---------------------------------------------------------------------------------------------------
struct Pump {
	Shared *sh = nullptr;
	Run *run = nullptr;
	int index = 0, device = 0, n_dev = 1;
	pamd_assembler *a = nullptr;
	const pamd_pipe_options *o = nullptr;
	DevInput in[2];
	DeviceBuffers mem;
	std::vector<Piece> pieces;
	pamd_fastq_parser *fq = nullptr;
	pamd_formatter *fmt = nullptr;
	hipStream_t s_cmp = nullptr, s_out = nullptr;
	char *d_out[2] = { nullptr, nullptr };
	pamd_qual *d_fwd = nullptr, *d_rev = nullptr;
	uint16_t *d_flen = nullptr, *d_rlen = nullptr;
	pamd_seqid *d_ids = nullptr;
	pamd_pair_result *d_res = nullptr;
	uint8_t *d_nt = nullptr;
	double *d_p = nullptr;
	unsigned long long *d_cursor = nullptr;
	long long *d_ctr = nullptr, *h_ctr = nullptr;
	hipEvent_t ev_out_done[2] = { nullptr, nullptr };
	size_t chunk_pairs = 0, seq_cap = 0;
	uint64_t text_cap = 0, out_cap = 0;
	std::vector<std::pair<uint64_t, uint64_t>> lens;	// (forward, reverse) text bytes of this device's chunks, appended by the cutter under in[0].m
	std::thread thread;
	uint64_t chunks_done = 0;

	bool setup(size_t n_readers);
	void destroy();
	void main();
};

struct Run {
	Shared sh;
	FileIndex idx[2];
	Output out;
	Ledger ledger;
	std::vector<std::unique_ptr<Pump>> pumps;
	void wake_all() {
		for (FileIndex &f : idx) {
			std::lock_guard<std::mutex> g(f.m);
			f.cv.notify_all();
		}
		for (auto &p : pumps)
			for (DevInput &s : p->in) {
				std::lock_guard<std::mutex> g(s.m);
				s.cv.notify_all();
			}
		{
			std::lock_guard<std::mutex> g(out.m);
			out.cv.notify_all();
		}
		{
			std::lock_guard<std::mutex> g(ledger.m);
			ledger.cv.notify_all();
		}
	}
};

void Shared::fail(int rc, const char *what, const char *why) {
	{
		std::lock_guard<std::mutex> g(msg_lock);
		if (!failed.exchange(true)) {
			msg = std::string(what) + ": " + why;
			code = rc;
		}
	}
	if (run != nullptr)
		run->wake_all();
}

bool Pump::setup(size_t n_readers) {
	mem.device = device;
	if (hipSetDevice(device) != hipSuccess)
		return false;
	const size_t stride = 456;
	fq = pamd_fastq_parser_create(device, o->qualmin ? o->qualmin : 33, o->policy);
	fmt = pamd_formatter_create(device);
	bool ok = fq != nullptr && fmt != nullptr && hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking) == hipSuccess;
	for (int s = 0; s < 2 && ok; s++) {
		DevInput &d = in[s];
		d.sh = sh;
		d.device = device;
		for (uint64_t &f : d.filled)
			f = ~0ull;
		ok = ok && mem.dmalloc(&d.text[0], text_cap + 16) && mem.dmalloc(&d.text[1], text_cap + 16);
		for (int k = 0; k < DevInput::NB && ok; k++)
			ok = mem.hmalloc(&d.pin[k], d.block);
		ok = ok && hipStreamCreateWithFlags(&d.s_in, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&d.ev[0], hipEventDisableTiming) == hipSuccess
		     && hipEventCreateWithFlags(&d.ev[1], hipEventDisableTiming) == hipSuccess;
	}
	pieces.resize(6);
	for (Piece &p : pieces)
		ok = ok && mem.hmalloc(&p.pin, run->out.piece) && hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) == hipSuccess;
	ok = ok && mem.dmalloc(&d_out[0], out_cap) && mem.dmalloc(&d_out[1], out_cap) && mem.dmalloc(&d_fwd, chunk_pairs * stride * 2) && mem.dmalloc(&d_rev, chunk_pairs * stride * 2)
	     && mem.dmalloc(&d_flen, chunk_pairs * 2) && mem.dmalloc(&d_rlen, chunk_pairs * 2) && mem.dmalloc(&d_ids, chunk_pairs * sizeof(pamd_seqid)) && mem.dmalloc(&d_res, chunk_pairs * sizeof(pamd_pair_result))
	     && mem.dmalloc(&d_nt, seq_cap) && mem.dmalloc(&d_p, seq_cap * sizeof(double)) && mem.dmalloc(&d_cursor, 8) && mem.dmalloc(&d_ctr, PAMD_NUM_COUNTERS * 8) && mem.hmalloc(&h_ctr, PAMD_NUM_COUNTERS * 8)
	     && hipEventCreateWithFlags(&ev_out_done[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_out_done[1], hipEventDisableTiming) == hipSuccess;
	if (!ok)
		return false;
	for (DevInput &d : in) {
		for (size_t t = 0; t < n_readers; t++)
			d.readers.emplace_back(&DevInput::reader_main, &d);
		d.copier = std::thread(&DevInput::copier_main, &d);
	}
	return true;
}

void Pump::destroy() {
	(void) hipSetDevice(device);
	for (DevInput &d : in) {
		if (d.s_in)
			(void) hipStreamDestroy(d.s_in);
		for (hipEvent_t e : d.ev)
			if (e)
				(void) hipEventDestroy(e);
	}
	for (Piece &p : pieces)
		if (p.ev)
			(void) hipEventDestroy(p.ev);
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
}

void Pump::main() {
	if (hipSetDevice(device) != hipSuccess) {
		sh->fail(PAMD_ENODEVICE, "hipSetDevice", "pump thread");
		return;
	}
	const size_t stride = 456;
	const bool trace = getenv("PAMD_PIPE_TRACE") != nullptr;
	for (uint64_t i = 0; !sh->failed.load(); i++) {
		const uint64_t k = i * (uint64_t) n_dev + (uint64_t) index;	// the global chunk number
		const double c0 = now_s();
		if (!in[0].wait_chunk(i) || !in[1].wait_chunk(i))
			break;	// no such chunk (or the run failed)
		uint64_t have[2];
		{
			std::lock_guard<std::mutex> g(in[0].m);
			have[0] = lens[i].first;
			have[1] = lens[i].second;
		}
		const int cur = (int) (i & 1);
		ChunkResult r;
		r.in_bytes = have[0] + have[1];
		const double c1 = now_s();
		double c2 = c1, c3 = c1;
		if (!run->ledger.run_over(k)) {
			pamd_fastq_status st;
			int rc = pamd_fastq_parse_device(fq, in[0].text[cur], have[0], in[1].text[cur], have[1], chunk_pairs, d_fwd, stride, d_rev, stride, d_flen, d_rlen, d_ids, &st, s_cmp);
			if (rc != PAMD_OK) {
				sh->fail(rc, "pamd_fastq_parse_device", pamd_last_error());
				break;
			}
			c2 = c3 = now_s();
			r.records = st.n_records;
			r.good = st.n_good;
			r.dropped = st.n_dropped;
			r.stop_code = st.stop_code;
			r.left_fwd = st.stop_code == PAMD_FQ_END && st.fwd_consumed < have[0];
			r.left_rev = st.stop_code == PAMD_FQ_END && st.rev_consumed < have[1];
			size_t text_bytes = 0;
			r.counters.assign(PAMD_NUM_COUNTERS, 0);
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
				if (hipMemsetAsync(d_cursor, 0, 8, s_cmp) != hipSuccess || hipMemsetAsync(d_ctr, 0, PAMD_NUM_COUNTERS * 8, s_cmp) != hipSuccess) {
					sh->fail(PAMD_ENODEVICE, "hipMemsetAsync", "cursor");
					break;
				}
				if ((rc = pamd_assemble_device(a, &b, s_cmp)) != PAMD_OK) {
					sh->fail(rc, "pamd_assemble_device", pamd_last_error());
					break;
				}
				// the chunk's own counters come back with it (the formatter below synchronises the stream); this text buffer's previous
				// contents (two chunks ago) have left for the host
				if (hipMemcpyAsync(h_ctr, d_ctr, PAMD_NUM_COUNTERS * 8, hipMemcpyDeviceToHost, s_cmp) != hipSuccess || hipStreamWaitEvent(s_cmp, ev_out_done[cur], 0) != hipSuccess) {
					sh->fail(PAMD_ENODEVICE, "hipMemcpyAsync", "counters");
					break;
				}
				size_t n_fix = 0;
				uint64_t fix[2 * 64];
				if ((rc = pamd_format_device(fmt, o->format, st.n_good, d_res, d_nt, d_p, d_ids, in[0].text[cur], d_out[cur], out_cap, &text_bytes, fix, 64, &n_fix, s_cmp)) != PAMD_OK) {
					sh->fail(rc, "pamd_format_device", pamd_last_error());
					break;
				}
				for (size_t f = 0; f < n_fix && !sh->failed.load(); f++) {	// the "%f" fields the device would not decide: re-printed as output.c:93 prints them
					pamd_pair_result one;
					char tmp[32];
					if (hipMemcpy(&one, d_res + fix[2 * f + 1], sizeof one, hipMemcpyDeviceToHost) != hipSuccess) {
						sh->fail(PAMD_ENODEVICE, "hipMemcpy", "fixup");
						break;
					}
					snprintf(tmp, sizeof tmp, "%f", exp(one.quality));
					if (hipMemcpy(d_out[cur] + fix[2 * f], tmp, 8, hipMemcpyHostToDevice) != hipSuccess)
						sh->fail(PAMD_ENODEVICE, "hipMemcpy", "fixup");
				}
				for (unsigned c = 0; c < PAMD_NUM_COUNTERS; c++)
					r.counters[c] = h_ctr[c];
				c3 = now_s();
			}
			r.text_bytes = text_bytes;
			const uint64_t n_good = st.n_good;
			run->ledger.publish(k, std::move(r));
			uint64_t out_off = 0;
			if (text_bytes > 0 && run->ledger.wait_offset(k, &out_off)) {
				for (uint64_t at = 0; at < text_bytes && !sh->failed.load(); at += run->out.piece) {
					const uint64_t len = std::min<uint64_t>(run->out.piece, text_bytes - at);
					Piece *p = run->out.take(pieces);
					if (p == nullptr)
						break;
					if (hipMemcpyAsync(p->pin, d_out[cur] + at, len, hipMemcpyDeviceToHost, s_out) != hipSuccess || hipEventRecord(p->ev, s_out) != hipSuccess) {
						sh->fail(PAMD_ENODEVICE, "hipMemcpyAsync", "device to host");
						break;
					}
					run->out.submit(p, out_off + at, len);
				}
				if (hipEventRecord(ev_out_done[cur], s_out) != hipSuccess)
					sh->fail(PAMD_ENODEVICE, "hipEventRecord", "output buffer");
			}
			if (trace)
				fprintf(stderr, "pamd_pipe[%d]: chunk %llu: %llu pairs; input %.4f s, parse %.4f s, assemble + format %.4f s, output hand-off %.4f s\n", index, (unsigned long long) k, (unsigned long long) n_good, c1 - c0, c2 - c1, c3 - c2, now_s() - c3);
		} else {
			run->ledger.publish(k, std::move(r));	// (dealt before the run ended in an earlier chunk: nothing of it counts)
		}
		chunks_done = i + 1;
		in[0].release(i + 1);
		in[1].release(i + 1);
	}
	(void) hipStreamSynchronize(s_cmp);
	(void) hipStreamSynchronize(s_out);
}

}	// namespace

// pool.c:86-104 (do_assembly's last lines) and proxy.c:175-190 (panda_log_proxy_write_overlap): the STAT block one worker prints at
// the end of its input, from a counter block -- a device's own, or the sum over devices.  READS is pool.c:66-67's last line
// (printtime: the assembler's count); TIME / ELAPSED are the caller's clock and are left to it.
extern "C" int pamd_stat_block(const int64_t *counters, int has_forward_primer, int has_reverse_primer, const char *const *module_names, int n_modules, const char *name, char *buf, size_t cap) {
	if (counters == nullptr || buf == nullptr || cap == 0 || n_modules < 0 || n_modules > PAMD_MAX_MODULES || (n_modules > 0 && module_names == nullptr)) {
		pamd_set_error("pamd_stat_block: bad argument");
		return PAMD_EINVAL;
	}
	std::string s;
	const std::string pre = name != nullptr ? std::string(name) + "\t" : std::string();
	auto line = [&](const char *key, long long v) {
		s += pre + "STAT\t" + key + "\t" + std::to_string(v) + "\n";
	};
	line("READS", counters[PAMD_CTR_COUNT]);
	if (has_forward_primer)
		line("NOFP", counters[PAMD_CTR_NOFP]);
	if (has_reverse_primer)
		line("NORP", counters[PAMD_CTR_NORP]);
	line("NOALGN", counters[PAMD_CTR_NOALGN]);
	line("LOWQ", counters[PAMD_CTR_LOWQ]);
	line("BADR", counters[PAMD_CTR_BADR]);
	line("SLOW", counters[PAMD_CTR_SLOW]);
	for (int m = 0; m < n_modules; m++)	// module.c:208-216: a module that rejected something has a line
		if (counters[PAMD_CTR_REJECTED0 + m] > 0)
			line(module_names[m], counters[PAMD_CTR_REJECTED0 + m]);
	line("OK", counters[PAMD_CTR_OK]);
	int longest = 0;	// a running maximum in the reference (assembler.c:342-344): the highest overlap counted
	for (int i = 2 * PAMD_MAX_LEN - 1; i > 0; i--)
		if (counters[PAMD_CTR_OVERLAP0 + i] != 0) {
			longest = i;
			break;
		}
	s += pre + "STAT\tOVERLAPS\t" + std::to_string((long long) counters[PAMD_CTR_OVERLAP0]);
	for (int i = 1; i <= longest; i++)
		s += " " + std::to_string((long long) counters[PAMD_CTR_OVERLAP0 + i]);
	s += "\n";
	if (s.size() + 1 > cap) {
		pamd_set_error("pamd_stat_block: the block needs %zu bytes", s.size() + 1);
		return PAMD_EINVAL;
	}
	memcpy(buf, s.c_str(), s.size() + 1);
	return (int) s.size();
}

extern "C" int pamd_pipe_run_multi(pamd_assembler *const *assemblers, const int *devices, int n_devices, const pamd_pipe_options *o, pamd_pipe_stats *stats) {
	if (assemblers == nullptr || devices == nullptr || n_devices < 1 || n_devices > 64 || o == nullptr || stats == nullptr || o->forward_path == nullptr || o->reverse_path == nullptr || o->out_path == nullptr) {
		pamd_set_error("pamd_pipe_run_multi: bad argument");
		return PAMD_EINVAL;
	}
	for (int d = 0; d < n_devices; d++)
		if (assemblers[d] == nullptr) {
			pamd_set_error("pamd_pipe_run_multi: an assembler per device");
			return PAMD_EINVAL;
		}
	memset(stats, 0, sizeof *stats);
	const double t_start = now_s();
	const size_t chunk_pairs = o->pairs_per_chunk ? o->pairs_per_chunk : 1000000;
	const uint64_t text_cap = (uint64_t) chunk_pairs * 1400 + 65536;
	if (text_cap >= (1ull << 32)) {
		pamd_set_error("pamd_pipe_run_multi: at most 3 000 000 pairs per chunk (a text chunk must stay below 4 GiB)");
		return PAMD_EINVAL;
	}
	const char *paths[2] = { o->forward_path, o->reverse_path };
	Run run;
	run.sh.run = &run;
	int fds[2] = { -1, -1 };
	bool streamed = false;
	for (int s = 0; s < 2; s++) {
		fds[s] = open(paths[s], O_RDONLY);
		struct stat sb;
		if (fds[s] < 0 || fstat(fds[s], &sb) != 0) {
			pamd_set_error("pamd_pipe_run_multi: cannot open %s", paths[s]);
			for (int f : fds)
				if (f >= 0)
					close(f);
			return PAMD_EINVAL;
		}
		unsigned char magic[2] = { 0, 0 };
		const bool regular = S_ISREG(sb.st_mode);
		if (!regular || (pread(fds[s], magic, 2, 0) == 2 && ((magic[0] == 'B' && magic[1] == 'Z') || (magic[0] == 0x1f && magic[1] == 0x8b))))
			streamed = true;
		run.idx[s].size = regular ? (uint64_t) sb.st_size : 0;
	}
	if (streamed) {	// read in order through zlib / libbz2, or from a FIFO: one reader's work whatever the number of devices
		for (int f : fds)
			close(f);
		return pamd_pipe_run(assemblers[0], devices[0], o, stats);
	}
	auto close_all = [&]() {
		for (int &f : fds) {
			if (f >= 0)
				close(f);
			f = -1;
		}
		if (run.out.fd >= 0)
			close(run.out.fd);
		run.out.fd = -1;
	};
	run.out.sh = &run.sh;
	run.out.fd = open(o->out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (run.out.fd < 0) {
		pamd_set_error("pamd_pipe_run_multi: cannot create %s", o->out_path);
		close_all();
		return PAMD_EINVAL;
	}
	const uint64_t out_cap = (uint64_t) chunk_pairs * 2400 + 4096;
	run.out.seekable = lseek(run.out.fd, 0, SEEK_CUR) != (off_t) -1;
	run.out.piece = std::min<uint64_t>(std::max<uint64_t>((out_cap / 4 + 65535) & ~65535ull, 65536), 32u << 20);
	run.ledger.sh = &run.sh;
	for (int s = 0; s < 2; s++) {
		FileIndex &f = run.idx[s];
		f.sh = &run.sh;
		f.fd = fds[s];
		f.n_seg = (f.size + FileIndex::SEG - 1) / FileIndex::SEG;
		f.count.assign(f.n_seg, 0);
		f.done.assign(f.n_seg, 0);
	}
	const int n_readers = o->reader_threads > 0 ? o->reader_threads : 3, n_writers = !run.out.seekable ? 1 : (o->writer_threads > 0 ? o->writer_threads : 1);
	uint64_t block = (text_cap / 4 + 65535) & ~65535ull;
	block = std::min<uint64_t>(std::max<uint64_t>(block, 65536), 32u << 20);
	bool ok = true;
	for (int d = 0; d < n_devices && ok; d++) {
		run.pumps.emplace_back(new Pump);
		Pump &p = *run.pumps.back();
		p.sh = &run.sh;
		p.run = &run;
		p.index = d;
		p.device = devices[d];
		p.n_dev = n_devices;
		p.a = assemblers[d];
		p.o = o;
		p.chunk_pairs = chunk_pairs;
		p.text_cap = text_cap;
		p.out_cap = out_cap;
		p.seq_cap = chunk_pairs * 912 + (size_t) 16384 * 4096;
		for (int s = 0; s < 2; s++) {
			p.in[s].fd = fds[s];
			p.in[s].block = block;
		}
		ok = p.setup((size_t) n_readers);
	}
	auto stop_inputs = [&]() {	// no further blocks are planned; what is planned is read, sent up and dropped
		for (auto &p : run.pumps)
			for (DevInput &s : p->in) {
				{
					std::lock_guard<std::mutex> g(s.m);
					s.plan_final = true;
					s.released = ~0ull - 4;
				}
				s.cv.notify_all();
			}
	};
	auto join_inputs = [&]() {
		for (auto &p : run.pumps)
			for (DevInput &s : p->in) {
				for (std::thread &t : s.readers)
					t.join();
				s.readers.clear();
				if (s.copier.joinable())
					s.copier.join();
			}
	};
	if (!ok) {
		run.sh.fail(PAMD_ENODEVICE, "setup", "out of device or pinned memory, or no such device");
		stop_inputs();
		join_inputs();
		for (auto &p : run.pumps)
			p->destroy();
		close_all();
		pamd_set_error("pamd_pipe_run_multi: out of device or pinned memory for %zu pairs per chunk on %d devices (%s)", chunk_pairs, n_devices, pamd_last_error());
		return PAMD_ENODEVICE;
	}
	const int n_counters = std::max(2, std::min(4, (int) std::thread::hardware_concurrency() / 4));
	for (FileIndex &f : run.idx)
		for (int t = 0; t < n_counters; t++)
			f.threads.emplace_back(&FileIndex::counter_main, &f);
	for (int t = 0; t < n_writers; t++)
		run.out.writers.emplace_back(&Output::writer_main, &run.out);
	for (auto &p : run.pumps)
		p->thread = std::thread(&Pump::main, p.get());
	const double t_ready = now_s();

	// ---- the cutter (this thread): chunk k = records [k R, (k + 1) R) of both files, to device k mod N ----
	uint64_t n_chunks = 0;
	{
		uint64_t at[2] = { 0, 0 };
		uint64_t rec = 0;
		while (!run.sh.failed.load()) {
			if (at[0] >= run.idx[0].size || at[1] >= run.idx[1].size)
				break;	// one of the files is used up: what is left of the other has no mate (fastq.c:123,130)
			{
				std::lock_guard<std::mutex> g(run.ledger.m);
				if (run.ledger.stopped)
					break;	// the reader gave up in a chunk already parsed
			}
			uint64_t take = chunk_pairs, end[2] = { 0, 0 };
			bool fits = false;
			while (!fits) {	// (a chunk's text has to fit its buffer: records of more than 1.4 KB on average make chunks of fewer of them)
				if (!run.idx[0].locate(4 * (rec + take), &end[0]) || !run.idx[1].locate(4 * (rec + take), &end[1]))
					break;
				fits = end[0] - at[0] <= text_cap && end[1] - at[1] <= text_cap;
				if (!fits) {
					if (take == 1) {
						run.sh.fail(PAMD_EINVAL, "input", "a record of more text than a chunk holds");
						break;
					}
					take = (take + 1) / 2;
					// (locate() only walks forwards: start over from this chunk's first segment)
					for (int s = 0; s < 2; s++) {
						run.idx[s].cum_seg = 0;
						run.idx[s].cum_lines = 0;
					}
				}
			}
			if (!fits)
				break;
			Pump &p = *run.pumps[n_chunks % (uint64_t) n_devices];
			const uint64_t local = n_chunks / (uint64_t) n_devices;
			{
				std::lock_guard<std::mutex> g(p.in[0].m);
				p.lens.emplace_back(end[0] - at[0], end[1] - at[1]);
			}
			for (int s = 0; s < 2; s++) {
				DevInput &d = p.in[s];
				const uint64_t len = end[s] - at[s];
				{
					std::lock_guard<std::mutex> g(d.m);
					uint64_t o2 = 0;
					do {
						const uint64_t l = std::min<uint64_t>(block, len - o2);
						d.plan.push_back({ at[s] + o2, l, local, o2, o2 + l >= len });
						o2 += l;
					} while (o2 < len);
				}
				d.cv.notify_all();
			}
			n_chunks++;
			at[0] = end[0];
			at[1] = end[1];
			rec += take;
			// the devices need not be dealt work far ahead of what they have done: two chunks each in hand
			{
				std::unique_lock<std::mutex> g(p.in[0].m);
				p.in[0].cv.wait(g, [&] { return run.sh.failed.load() || p.in[0].released + 3 > local; });
			}
		}
	}
	{	// nothing more is dealt
		for (auto &p : run.pumps)
			for (DevInput &s : p->in) {
				{
					std::lock_guard<std::mutex> g(s.m);
					s.plan_final = true;
				}
				s.cv.notify_all();
			}
		for (FileIndex &f : run.idx) {
			{
				std::lock_guard<std::mutex> g(f.m);
				f.next_batch = ~0ull / FileIndex::BATCH;	// (counting stops)
			}
			f.cv.notify_all();
		}
	}
	for (auto &p : run.pumps)
		p->thread.join();
	// ---- wind down: nothing may be left in flight ----
	stop_inputs();
	join_inputs();
	for (FileIndex &f : run.idx) {
		for (std::thread &t : f.threads)
			t.join();
		f.threads.clear();
	}
	run.out.drain_and_stop();
	const double t_end = now_s();
	// what the chunks came to: everything up to and including the chunk the reader gave up in
	uint64_t records = 0, dropped = 0, in_bytes = 0, out_bytes = 0, stop_record = 0;
	int stop = PAMD_FQ_END;
	long long ctr[PAMD_NUM_COUNTERS];
	memset(ctr, 0, sizeof ctr);
	if (!run.sh.failed.load()) {
		Ledger &L = run.ledger;
		const uint64_t upto = L.stopped ? L.stop_chunk + 1 : n_chunks;
		for (uint64_t k = 0; k < upto; k++) {
			if (k >= L.res.size() || !L.res[k].known) {
				run.sh.fail(PAMD_EINVAL, "ledger", "a chunk was never assembled");
				break;
			}
			const ChunkResult &c = L.res[k];
			records += c.good;
			dropped += c.dropped;
			in_bytes += c.in_bytes;
			out_bytes += c.text_bytes;
			for (unsigned i = 0; i < PAMD_NUM_COUNTERS && i < c.counters.size(); i++)
				ctr[i] += c.counters[i];
			if (c.stop_code != PAMD_FQ_END) {
				stop = c.stop_code;
				stop_record = records;
			} else if (k + 1 == upto && c.left_fwd && c.left_rev) {
				// both files end inside a record -- the headers are read and read_seq() runs out of lines: PANDA_CODE_PREMATURE_EOF
				// (fastq.c:58-60,69-71,84-86); one file alone ending early ends the stream without a word (fastq.c:123,130)
				stop = PAMD_FQ_EOF;
				stop_record = records;
			} else if (k + 1 < upto && (c.left_fwd || c.left_rev)) {
				run.sh.fail(PAMD_EINVAL, "chunk", "a chunk of whole records was not parsed to its end");
			}
		}
	}
	for (unsigned k = 0; k < PAMD_NUM_COUNTERS; k++)
		stats->counters[k] = ctr[k];
	stats->pairs = records;
	stats->dropped = dropped;
	stats->chunks = n_chunks;
	stats->in_bytes = in_bytes;
	stats->out_bytes = out_bytes;
	stats->stop_code = stop;
	stats->stop_record = stop_record;
	stats->seconds_setup = t_ready - t_start;
	stats->seconds_pump = t_end - t_ready;
	for (auto &p : run.pumps)
		p->destroy();
	{	// (a regular file: what was written is its size -- nothing was written beyond it, the pieces of chunks dealt behind the one the reader gave up in never left)
		struct stat sb;
		if (!run.sh.failed.load() && run.out.seekable && fstat(run.out.fd, &sb) == 0 && S_ISREG(sb.st_mode) && ftruncate(run.out.fd, (off_t) out_bytes) != 0)
			run.sh.fail(PAMD_EIO, "ftruncate", strerror(errno));
	}
	close_all();
	if (run.sh.failed.load()) {
		pamd_set_error("pamd_pipe_run_multi: %s", run.sh.msg.c_str());
		return run.sh.code != PAMD_OK ? run.sh.code : PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
