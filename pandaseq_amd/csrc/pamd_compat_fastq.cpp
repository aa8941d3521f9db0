// pamd_compat_fastq.cpp -- the reference's FASTQ-reader constructors on the device parser.
//
//   panda_create_fastq_reader()          fastq.c:200-237        -> a PandaNextSeq whose pairs come from
//                                                                  pamd_fastq_parse_device() (pamd_io.hip)
//   panda_open_buffer()                  fileio.c:86-96         plain or gzip files through zlib (no bzip2 here)
//   panda_open_fastq()                   fileio.c:98-137
//   panda_assembler_new_fastq_reader()   assembler_support.c:102-117
//   panda_assembler_open_fastq() / panda_assembler_open()      fileio.c:139-153 (the header declares the
//                                                                  second name, the reference defines the first)
//
// The reader keeps the reference's per-pair contract -- next() hands out one pair, its buffers valid until
// the following call, and returns false where stream_next_seq() (fastq.c:108-198) returns false -- but
// parses a whole chunk of text on the GPU per refill: bytes are pulled through the two PandaBufferRead
// callbacks into host buffers, copied to HBM, turned into rows of panda_qual and parsed headers there, and
// copied back.  A caller that wants the rows to STAY on the device uses the batch ABI instead
// (include/pandaseq_amd_io.h); this file is the drop-in for code written against
// pandaseq.h.  The index-read file (`-i`, fastq.c:155-187) is not supported: with one the constructors fail.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "pamd_internal.h"
#include "pamd_bz2.h"
#include "../../include/pandaseq_amd_compat.h"
#include "../../include/pandaseq_amd_io.h"

namespace {

constexpr size_t CHUNK_PAIRS = 65536;		// pairs parsed per refill
constexpr size_t CHUNK_BYTES = 96u << 20;	// text kept per side: room for CHUNK_PAIRS records of ~1.4 KB
constexpr size_t STRIDE = 456;			// row stride in bases (PANDA_MAX_LEN rounded up to 8)

thread_local int g_last_stop = PAMD_FQ_END;

struct Source {
	PandaBufferRead read = nullptr;
	void *data = nullptr;
	PandaDestroy destroy = nullptr;
	std::vector<char> buf;	// [pos, len) holds text not yet parsed
	size_t pos = 0, len = 0;
	bool eof = false;
	// top the buffer up to CHUNK_BYTES; false on a read error (the reference returns NULL from its line reader then)
	bool fill() {
		if (pos > 0) {
			memmove(buf.data(), buf.data() + pos, len - pos);
			len -= pos;
			pos = 0;
		}
		while (!eof && len < CHUNK_BYTES) {
			size_t got = 0;
			if (!read(buf.data() + len, CHUNK_BYTES - len, &got, data)) {
				// linebuf.c:66-70 returns NULL at the first line it cannot complete: the whole lines already read are still handed
				// out, the broken one is not (a bzip2 source ends this way: fileio.c:48-58 reports the call after BZ_STREAM_END)
				eof = true;
				while (len > 0 && buf[len - 1] != '\n')
					len--;
				break;
			}
			if (got == 0)
				eof = true;
			len += got;
		}
		return true;
	}
	void close() {
		if (destroy && data)
			destroy(data);
		data = nullptr;
	}
};

struct Reader {
	int device = 0;
	Source f, r;
	pamd_fastq_parser *parser = nullptr;
	char *d_f = nullptr, *d_r = nullptr;
	pamd_qual *d_fwd = nullptr, *d_rev = nullptr;
	uint16_t *d_flen = nullptr, *d_rlen = nullptr;
	pamd_seqid *d_ids = nullptr;
	std::vector<pamd_qual> h_fwd, h_rev;
	std::vector<uint16_t> h_flen, h_rlen;
	std::vector<pamd_seqid> h_ids;
	std::vector<char> h_text;	// the forward text of the current chunk (the ids point into it)
	size_t n = 0, at = 0;	// parsed records of the current chunk, next one to hand out
	bool dead = false;	// the reader has stopped for good
};

void reader_free(Reader *rd) {
	if (rd == nullptr)
		return;
	(void) hipSetDevice(rd->device);
	rd->f.close();
	rd->r.close();
	if (rd->parser)
		pamd_fastq_parser_destroy(rd->parser);
	void *bufs[] = { rd->d_f, rd->d_r, rd->d_fwd, rd->d_rev, rd->d_flen, rd->d_rlen, rd->d_ids };
	for (void *p : bufs)
		if (p)
			(void) hipFree(p);
	delete rd;
}

// parse the next chunk; false = nothing more will come
bool refill(Reader *rd) {
	rd->n = rd->at = 0;
	if (rd->dead)
		return false;
	if (!rd->f.fill() || !rd->r.fill()) {
		rd->dead = true;
		return false;
	}
	const size_t fn = rd->f.len - rd->f.pos, rn = rd->r.len - rd->r.pos;
	if (fn == 0 || rn == 0) {	// one file is exhausted: panda_linebuf_next() == NULL on a header line (fastq.c:123,130)
		rd->dead = true;
		return false;
	}
	if (hipSetDevice(rd->device) != hipSuccess || hipMemcpy(rd->d_f, rd->f.buf.data() + rd->f.pos, fn, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(rd->d_r, rd->r.buf.data() + rd->r.pos, rn, hipMemcpyHostToDevice) != hipSuccess) {
		pamd_set_error("FASTQ reader: copy to the device failed");
		rd->dead = true;
		return false;
	}
	pamd_fastq_status st;
	if (pamd_fastq_parse_device(rd->parser, rd->d_f, fn, rd->d_r, rn, CHUNK_PAIRS, rd->d_fwd, STRIDE, rd->d_rev, STRIDE, rd->d_flen, rd->d_rlen, rd->d_ids, &st, nullptr) != PAMD_OK) {
		fprintf(stderr, "pandaseq_amd: %s\n", pamd_last_error());
		rd->dead = true;
		return false;
	}
	const size_t good = (size_t) st.n_good;
	if (good > 0) {
		if (hipMemcpy(rd->h_fwd.data(), rd->d_fwd, good * STRIDE * sizeof(pamd_qual), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(rd->h_rev.data(), rd->d_rev, good * STRIDE * sizeof(pamd_qual), hipMemcpyDeviceToHost) != hipSuccess
		    || hipMemcpy(rd->h_flen.data(), rd->d_flen, good * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(rd->h_rlen.data(), rd->d_rlen, good * 2, hipMemcpyDeviceToHost) != hipSuccess
		    || hipMemcpy(rd->h_ids.data(), rd->d_ids, good * sizeof(pamd_seqid), hipMemcpyDeviceToHost) != hipSuccess) {
			pamd_set_error("FASTQ reader: copy from the device failed");
			rd->dead = true;
			return false;
		}
		// the ids are spans of this chunk's forward text: keep the part they can point into
		const size_t keep = (size_t) st.fwd_consumed < fn ? (size_t) st.fwd_consumed : fn;
		rd->h_text.assign(rd->f.buf.data() + rd->f.pos, rd->f.buf.data() + rd->f.pos + keep);
	}
	rd->n = good;
	if (st.stop_code != PAMD_FQ_END) {	// the reference's reader logs this code and returns false here, for good
		g_last_stop = st.stop_code;
		rd->dead = true;
	} else if (st.n_records == 0) {
		// no whole record left on one side: only a partial record (PANDA_CODE_PREMATURE_EOF in the reference when at
		// the end of the files) or a line longer than the whole buffer
		if (fn > 0 || rn > 0)
			g_last_stop = (rd->f.eof && rd->r.eof) ? PAMD_FQ_EOF : PAMD_FQ_LONG_HEADER;
		rd->dead = true;
	}
	rd->f.pos += (size_t) st.fwd_consumed;
	rd->r.pos += (size_t) st.rev_consumed;
	return rd->n > 0;
}

// panda_seq_identifier from the parsed spans (what panda_seqid_parse_fail() leaves in it, seqid.c:142-285)
void fill_id(panda_seq_identifier *id, const pamd_seqid &s, const char *text) {
	memset(id, 0, sizeof *id);
	const char *h = text + s.hdr_off;
	if (s.sra >= 0)
		snprintf(id->instrument, sizeof id->instrument, "%cRR%d", h[0], s.sra);	// seqid.c:163
	else
		memcpy(id->instrument, h, s.instr_len);
	memcpy(id->run, h + s.run_off, s.run_len);
	memcpy(id->flowcell, h + s.fc_off, s.fc_len);
	memcpy(id->tag, h + s.tag_off, s.tag_len < PAMD_TAG_LEN ? s.tag_len : PAMD_TAG_LEN);
	id->lane = s.lane;
	id->tile = s.tile;
	id->x = s.x;
	id->y = s.y;
}

bool reader_next(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	Reader *rd = static_cast<Reader *>(user_data);
	*forward = *reverse = nullptr;
	*forward_length = *reverse_length = 0;
	for (;;) {
		if (rd->at == rd->n && !refill(rd))
			return false;
		const size_t i = rd->at++;
		if (rd->h_ids[i].flag == 1)
			continue;	// an empty forward read: the reference's reader loops on (fastq.c:192)
		fill_id(id, rd->h_ids[i], rd->h_text.data());
		*forward = reinterpret_cast<const panda_qual *>(&rd->h_fwd[i * STRIDE]);
		*reverse = reinterpret_cast<const panda_qual *>(&rd->h_rev[i * STRIDE]);
		*forward_length = rd->h_flen[i];
		*reverse_length = rd->h_rlen[i];
		return true;
	}
}

void reader_destroy(void *user_data) {
	reader_free(static_cast<Reader *>(user_data));
}

bool gz_read(char *buffer, size_t buffer_length, size_t *read, void *user_data) {	// fileio.c:31-46
	const int got = gzread(static_cast<gzFile>(user_data), buffer, (unsigned) (buffer_length > (1u << 30) ? (1u << 30) : buffer_length));
	if (got < 0) {
		*read = 0;
		return false;
	}
	*read = (size_t) got;
	return true;
}

void gz_close(void *user_data) {
	gzclose(static_cast<gzFile>(user_data));
}

bool bz_read(char *buffer, size_t buffer_length, size_t *read, void *user_data) {	// fileio.c:48-58
	int bzerror = 0;
	const int got = pamd_bz2()->bzRead(&bzerror, user_data, buffer, (int) (buffer_length > (1u << 30) ? (1u << 30) : buffer_length));
	*read = got > 0 ? (size_t) got : 0;
	return bzerror == PAMD_BZ_OK || bzerror == PAMD_BZ_STREAM_END;
}

void bz_close(void *user_data) {
	pamd_bz2()->bzclose(user_data);
}

int g_fastq_device = 0;

}	// namespace

extern "C" {

void pamd_compat_set_fastq_device(int device) {
	g_fastq_device = device;
}

int pamd_compat_last_fastq_stop(void) {
	return g_last_stop;
}

// The device reader under a name of its own: what panda_create_fastq_reader() is in the standalone build, and what a program
// that links the reference's fastq.c / fileio.c (-DPAMD_WITH_REFERENCE_OBJECTS: those files stay)
// calls when it wants its text parsed on the GPU instead.
PandaNextSeq pamd_compat_create_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, PandaBufferRead index, void *index_data, PandaDestroy index_destroy, void **user_data, PandaDestroy *destroy) {
	(void) logger;
	*user_data = nullptr;
	*destroy = nullptr;
	Reader *rd = new (std::nothrow) Reader;
	bool ok = rd != nullptr && forward != nullptr && reverse != nullptr && index == nullptr;
	if (rd) {
		rd->f.read = forward;
		rd->f.data = forward_data;
		rd->f.destroy = forward_destroy;
		rd->r.read = reverse;
		rd->r.data = reverse_data;
		rd->r.destroy = reverse_destroy;
		rd->device = g_fastq_device;
	}
	if (index != nullptr) {
		pamd_set_error("panda_create_fastq_reader: a separate index-read file is not supported on the device path");
		if (index_destroy && index_data)
			index_destroy(index_data);
	}
	if (ok) {
		rd->parser = pamd_fastq_parser_create(rd->device, qualmin, (int) policy);
		ok = rd->parser != nullptr;
	}
	if (ok) {
		try {
			rd->f.buf.resize(CHUNK_BYTES);
			rd->r.buf.resize(CHUNK_BYTES);
			rd->h_fwd.resize(CHUNK_PAIRS * STRIDE);
			rd->h_rev.resize(CHUNK_PAIRS * STRIDE);
			rd->h_flen.resize(CHUNK_PAIRS);
			rd->h_rlen.resize(CHUNK_PAIRS);
			rd->h_ids.resize(CHUNK_PAIRS);
		} catch (const std::bad_alloc &) {
			ok = false;
		}
	}
	if (ok)
		ok = hipMalloc(reinterpret_cast<void **>(&rd->d_f), CHUNK_BYTES + 16) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&rd->d_r), CHUNK_BYTES + 16) == hipSuccess
			&& hipMalloc(reinterpret_cast<void **>(&rd->d_fwd), CHUNK_PAIRS * STRIDE * sizeof(pamd_qual)) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&rd->d_rev), CHUNK_PAIRS * STRIDE * sizeof(pamd_qual)) == hipSuccess
			&& hipMalloc(reinterpret_cast<void **>(&rd->d_flen), CHUNK_PAIRS * 2) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&rd->d_rlen), CHUNK_PAIRS * 2) == hipSuccess
			&& hipMalloc(reinterpret_cast<void **>(&rd->d_ids), CHUNK_PAIRS * sizeof(pamd_seqid)) == hipSuccess;
	if (!ok) {
		reader_free(rd);	// closes the two sources as the reference's constructor owns them
		return nullptr;
	}
	g_last_stop = PAMD_FQ_END;
	*user_data = rd;
	*destroy = reader_destroy;
	return reader_next;
}

#ifndef PAMD_WITH_REFERENCE_OBJECTS	// fastq.c:200-237, fileio.c:86-153: the reference's own in a link with its objects
PandaNextSeq panda_create_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, PandaBufferRead index, void *index_data, PandaDestroy index_destroy, void **user_data, PandaDestroy *destroy) {
	return pamd_compat_create_fastq_reader(forward, forward_data, forward_destroy, reverse, reverse_data, reverse_destroy, logger, qualmin, policy, index, index_data, index_destroy, user_data, destroy);
}

PandaBufferRead panda_open_buffer(const char *file_name, PandaLogProxy logger, void **user_data, PandaDestroy *destroy) {
	(void) logger;
	*user_data = nullptr;
	*destroy = nullptr;
	char magic[2];
	const int fd = open(file_name, O_RDONLY);
	if (fd < 0 || ::read(fd, magic, 2) != 2 || lseek(fd, 0, SEEK_SET) != 0) {	// fileio.c:70-74
		pamd_set_error("%s: cannot open", file_name);
		if (fd >= 0)
			close(fd);
		return nullptr;
	}
	if (magic[0] == 'B' && magic[1] == 'Z') {	// fileio.c:75-85
		const PamdBz2 *bz = pamd_bz2();
		void *b = bz ? bz->bzdopen(fd, "r") : nullptr;
		if (b == nullptr) {
			pamd_set_error(bz ? "%s: cannot open" : "%s: bzip2 input needs libbz2.so.1, which cannot be loaded here", file_name);
			close(fd);
			return nullptr;
		}
		*user_data = b;
		*destroy = bz_close;
		return bz_read;
	}
	gzFile f = gzdopen(fd, "rb");	// reads gzip and plain files alike (fileio.c:86-96)
	if (f == nullptr) {
		pamd_set_error("%s: cannot open", file_name);
		close(fd);
		return nullptr;
	}
	gzbuffer(f, 1u << 20);
	*user_data = f;
	*destroy = gz_close;
	return gz_read;
}

PandaNextSeq panda_open_fastq(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, const char *index, void **user_data, PandaDestroy *destroy) {
	*user_data = nullptr;
	*destroy = nullptr;
	if (index != nullptr) {
		pamd_set_error("panda_open_fastq: a separate index-read file is not supported on the device path");
		return nullptr;
	}
	void *fd = nullptr, *rdd = nullptr;
	PandaDestroy fdes = nullptr, rdes = nullptr;
	PandaBufferRead fr = panda_open_buffer(forward, logger, &fd, &fdes);
	if (fr == nullptr)
		return nullptr;
	PandaBufferRead rr = panda_open_buffer(reverse, logger, &rdd, &rdes);
	if (rr == nullptr) {
		fdes(fd);
		return nullptr;
	}
	return panda_create_fastq_reader(fr, fd, fdes, rr, rdd, rdes, logger, qualmin, policy, nullptr, nullptr, nullptr, user_data, destroy);
}

#endif	// PAMD_WITH_REFERENCE_OBJECTS

// assembler_support.c:102-117 (a file this library replaces in every build): on whichever panda_create_fastq_reader() the link has
PandaAssembler panda_assembler_new_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy) {
	void *user_data;
	PandaDestroy destroy;
	PandaNextSeq next = panda_create_fastq_reader(forward, forward_data, forward_destroy, reverse, reverse_data, reverse_destroy, logger, qualmin, policy, nullptr, nullptr, nullptr, &user_data, &destroy);
	if (next == nullptr)
		return nullptr;
	return panda_assembler_new(next, user_data, destroy, logger);
}

#ifndef PAMD_WITH_REFERENCE_OBJECTS	// fileio.c:139-153
PandaAssembler panda_assembler_open_fastq(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy) {
	void *user_data;
	PandaDestroy destroy;
	PandaNextSeq next = panda_open_fastq(forward, reverse, logger, qualmin, policy, nullptr, &user_data, &destroy);
	if (next == nullptr)
		return nullptr;
	return panda_assembler_new(next, user_data, destroy, logger);
}

#endif

// pandaseq-assembler.h:93-101 declares this name; the reference defines only panda_assembler_open_fastq (fileio.c:139)
PandaAssembler panda_assembler_open(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy) {
	return panda_assembler_open_fastq(forward, reverse, logger, qualmin, policy);
}

}	// extern "C"
