// pamd_io.hip -- gfx950 kernels and C ABI of include/pandaseq_amd_io.h: FASTQ text -> batch rows,
// and accepted pairs -> FASTA / FASTQ text.  Byte work bounded by HBM bandwidth; no MFMA.
//
// Ingest (replaces fastq.c:45-198 + linebuf.c:57-89 + seqid.c:142-285 + nt.c:46-114):
//   fq_count_kernel   newlines per 8 KB tile of each text                      (one wave per tile)
//   fq_scan_kernel    exclusive scan of the tile counts                        (one workgroup per text)
//   fq_index_kernel   byte offset of every '\n', in order                      (one wave per tile)
//   fq_parse_kernel   64 records per wave: headers parsed one record per LANE (a serial character
//                     walk, 64 of them side by side), then reads converted one record per WAVE
//                     (4 bases per lane, 8-byte row stores)
//   fq_final_kernel   first failing record, dropped records, longest reads, bytes consumed
// Formatting (replaces output.c:85-126 + seqid.c:109-116 + nt.c:126-150,167-173):
//   fmt_len_kernel    text length of every pair + per-workgroup sums
//   fmt_scan_kernel   exclusive scan of the workgroup sums
//   fmt_write_kernel  headers composed one pair per lane in LDS, then each pair's text written by its wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include "pamd_internal.h"
#include "../../include/pandaseq_amd_io.h"

namespace {

typedef unsigned long long u64;

#define HIP_TRY(call)                                                            \
	do {                                                                     \
		hipError_t e_ = (call);                                          \
		if (e_ != hipSuccess) {                                          \
			pamd_set_error("%s: %s", #call, hipGetErrorString(e_));  \
			return PAMD_ENODEVICE;                                   \
		}                                                                \
	} while (0)

constexpr unsigned TILE_BYTES = 8192;	// text bytes one wave indexes: 8 steps of 64 lanes x 16 bytes
constexpr unsigned TILE_STEPS = TILE_BYTES / 1024;
constexpr unsigned ID_STR_MAX = 100;	// sizeof(panda_seq_identifier.instrument), pandaseq-common.h:236-238

__device__ __forceinline__ int uni(int v) {
	return __builtin_amdgcn_readfirstlane(v);
}

// ---- newline index ------------------------------------------------------------------------------
// 4-bit map of the bytes of w that equal '\n' (exact zero-byte test on w ^ 0x0A0A0A0A, then the four
// flag bits gathered with one multiply)
__device__ __forceinline__ unsigned nl_nibble(unsigned w) {
	const unsigned x = w ^ 0x0A0A0A0Au;
	const unsigned t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);	// 0x80 in every zero byte of x
	return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// 16-bit newline map of the 16 bytes at byte offset `off` of a text of `bytes` bytes (16-byte aligned base)
__device__ __forceinline__ unsigned nl_mask16(const uint8_t *text, u64 bytes, u64 off) {
	if (off >= bytes)
		return 0u;
	unsigned m;
	if (off + 16 <= bytes) {
		const uint4 v = *reinterpret_cast<const uint4 *>(text + off);
		m = nl_nibble(v.x) | (nl_nibble(v.y) << 4) | (nl_nibble(v.z) << 8) | (nl_nibble(v.w) << 12);
	} else {
		// the last, partial group: only words that hold a valid byte are touched
		m = 0;
		const unsigned *wp = reinterpret_cast<const unsigned *>(text + off);
		for (unsigned k = 0; k < 4 && off + 4 * k < bytes; k++)
			m |= nl_nibble(wp[k]) << (4 * k);
		m &= (1u << (unsigned) (bytes - off)) - 1u;
	}
	return m;
}

struct TextRef {
	const uint8_t *text;
	u64 bytes;
	unsigned *tile_counts;	// becomes the exclusive prefix after the scan
	unsigned *line_end;
	unsigned n_tiles;
};

__global__ __launch_bounds__(256) void fq_count_kernel(const TextRef a, const TextRef b) {
	const int lane = threadIdx.x & 63;
	const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
	for (unsigned t = wave; t < a.n_tiles + b.n_tiles; t += nwaves) {
		const TextRef &T = t < a.n_tiles ? a : b;
		const unsigned tile = t < a.n_tiles ? t : t - a.n_tiles;
		const u64 base = (u64) tile * TILE_BYTES + (u64) lane * 16u;
		unsigned cnt = 0;
#pragma unroll
		for (unsigned s = 0; s < TILE_STEPS; s++)
			cnt += __popc(nl_mask16(T.text, T.bytes, base + (u64) s * 1024u));
		for (int o = 32; o > 0; o >>= 1)
			cnt += __shfl_xor(cnt, o);
		if (lane == 0)
			T.tile_counts[tile] = cnt;
	}
}

// exclusive scan of n unsigned values in place by one workgroup; total -> *total
__device__ void block_scan_u32(unsigned *c, unsigned n, unsigned *total) {
	__shared__ unsigned part[1024];
	const unsigned tid = threadIdx.x;
	const unsigned per = (n + 1023u) / 1024u;
	const unsigned lo = tid * per < n ? tid * per : n;
	const unsigned hi = lo + per < n ? lo + per : n;
	unsigned s = 0;
	for (unsigned i = lo; i < hi; i++)
		s += c[i];
	part[tid] = s;
	__syncthreads();
	for (unsigned off = 1; off < 1024; off <<= 1) {
		const unsigned v = tid >= off ? part[tid - off] : 0u;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	unsigned run = part[tid] - s;
	for (unsigned i = lo; i < hi; i++) {
		const unsigned t = c[i];
		c[i] = run;
		run += t;
	}
	if (tid == 1023)
		*total = part[1023];
}

__global__ __launch_bounds__(1024) void fq_scan_kernel(unsigned *c0, unsigned n0, unsigned *c1, unsigned n1, unsigned *totals) {
	if (blockIdx.x == 0)
		block_scan_u32(c0, n0, &totals[0]);
	else
		block_scan_u32(c1, n1, &totals[1]);
}

__global__ __launch_bounds__(256) void fq_index_kernel(const TextRef a, const TextRef b) {
	const int lane = threadIdx.x & 63;
	const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
	for (unsigned t = wave; t < a.n_tiles + b.n_tiles; t += nwaves) {
		const TextRef &T = t < a.n_tiles ? a : b;
		const unsigned tile = t < a.n_tiles ? t : t - a.n_tiles;
		unsigned out = T.tile_counts[tile];
		for (unsigned s = 0; s < TILE_STEPS; s++) {
			const u64 off = (u64) tile * TILE_BYTES + (u64) s * 1024u + (u64) lane * 16u;
			unsigned m = nl_mask16(T.text, T.bytes, off);
			const unsigned cnt = __popc(m);
			unsigned incl = cnt;	// inclusive prefix over lanes
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned v = __shfl_up(incl, o);
				if (lane >= o)
					incl += v;
			}
			unsigned at = out + incl - cnt;
			while (m != 0) {
				const unsigned bit = __builtin_ctz(m);
				m &= m - 1;
				T.line_end[at++] = (unsigned) off + bit;
			}
			out += __shfl(incl, 63);
		}
	}
}

// ---- header parsing, one header per lane ----------------------------------------------------------
// A header as the C string the reference's parser sees: the bytes of the line after the '@', then a
// terminator.  Bytes come from aligned words of the text, one word cached in registers.
struct Str {
	const uint8_t *text;	// 4-byte aligned
	unsigned start, len;
	unsigned cached;	// byte offset of the cached word, ~0 = none
	unsigned word;
	__device__ __forceinline__ int at(unsigned i) {
		if (i >= len)
			return 0;
		const unsigned a = start + i;
		if ((a & ~3u) != cached) {
			cached = a & ~3u;
			word = *reinterpret_cast<const unsigned *>(text + cached);
		}
		return (int) ((word >> (8u * (a & 3u))) & 0xFFu);
	}
};

struct IdOut {
	unsigned instr_len, run_off, run_len, fc_off, fc_len, tag_off, tag_len;
	int lane, tile, x, y, sra;
	int fmt;
};

__device__ __forceinline__ bool ends_chunk(int ch) {	// PARSE_CHUNK_MAYBE, seqid.c:136
	return ch == 0 || ch == ':' || ch == '#' || ch == '/' || ch == ' ';
}

__device__ __forceinline__ bool step(Str &h, unsigned &i) {	// PARSE_PUSH, seqid.c:141
	if (h.at(i) == 0)
		return false;
	i++;
	return true;
}

__device__ __forceinline__ bool chunk_int(Str &h, unsigned &i, int &out) {	// PARSE_INT, seqid.c:138 (32-bit wrap)
	unsigned v = 0;
	int ch = h.at(i);
	if (ch == 0)
		return false;
	while (!ends_chunk(ch)) {
		if (ch < '0' || ch > '9')
			return false;
		v = v * 10u + (unsigned) (ch - '0');
		i++;
		ch = h.at(i);
	}
	out = (int) v;
	return true;
}

__device__ __forceinline__ bool sra_int(Str &h, unsigned &i, int &out) {	// PARSE_SRA_INT, seqid.c:139
	unsigned v = 0;
	for (;;) {
		const int ch = h.at(i);
		if (ch == 0 || ch == '.' || ch == ' ')
			break;
		if (ch < '0' || ch > '9')
			return false;
		v = v * 10u + (unsigned) (ch - '0');
		i++;
	}
	out = (int) v;
	return true;
}

// PARSE_STR, seqid.c:140; a chunk of 100 characters or more fails (the reference overruns its field there)
__device__ __forceinline__ bool chunk_str(Str &h, unsigned &i, unsigned &off, unsigned &len) {
	if (h.at(i) == 0)
		return false;
	off = i;
	while (!ends_chunk(h.at(i))) {
		if (i - off + 1 >= ID_STR_MAX)
			return false;
		i++;
	}
	len = i - off;
	return true;
}

__device__ __forceinline__ bool chunk_tag(Str &h, unsigned &i, unsigned &off, unsigned &len) {	// seqid.c:222-228,268-274
	off = i;
	while (!ends_chunk(h.at(i))) {
		if (i - off >= PAMD_TAG_LEN)
			return false;
		i++;
	}
	len = i - off;
	return true;
}

__device__ __forceinline__ bool policy_ok(unsigned tag_len, int policy) {
	if (policy == PAMD_TAG_OPTIONAL)
		return true;
	return policy == (tag_len == 0 ? PAMD_TAG_ABSENT : PAMD_TAG_PRESENT);
}

#define NEED(x)               \
	do {                  \
		if (!(x))     \
			return 0; \
	} while (0)

// panda_seqid_parse_fail (seqid.c:142-285): the mate direction, 0 = not parsed
__device__ int parse_header(Str &h, int policy, IdOut &o) {
	unsigned i = 0;
	int value = 0;
	o.instr_len = o.run_off = o.run_len = o.fc_off = o.fc_len = o.tag_off = o.tag_len = 0;
	o.lane = o.tile = o.x = o.y = 0;
	o.sra = -1;
	o.fmt = PAMD_IDFMT_UNKNOWN;
	const int c0 = h.at(0), c1 = h.at(1), c2 = h.at(2), c3 = h.at(3);
	if (c0 != 0 && c1 != 0 && c2 != 0 && c3 != 0 && (c0 == 'E' || c0 == 'S') && c1 == 'R' && c2 == 'R') {
		o.fmt = c0 == 'S' ? PAMD_IDFMT_SRA : PAMD_IDFMT_EBI_SRA;
		i = 3;
		NEED(sra_int(h, i, value));
		NEED(step(h, i));
		o.sra = value;
		o.instr_len = 3;
		NEED(sra_int(h, i, value));
		NEED(step(h, i));
		o.lane = value;
		NEED(step(h, i));
		return 1;
	}
	bool slash = false;
	unsigned colons = 0;
	bool before_hash = true;
	for (unsigned k = 0;; k++) {
		const int ch = h.at(k);
		if (ch == 0)
			break;
		slash |= ch == '/';
		before_hash &= ch != '#';
		colons += (before_hash && ch == ':') ? 1u : 0u;
	}
	unsigned off, len;
	if (slash) {
		if (colons == 6) {
			o.fmt = PAMD_IDFMT_CASAVA_CONVERTED;
			NEED(chunk_str(h, i, off, len));
			o.instr_len = len;
			NEED(step(h, i));
			NEED(chunk_str(h, i, off, len));
			o.run_off = off;
			o.run_len = len;
			NEED(step(h, i));
			NEED(chunk_str(h, i, off, len));
			o.fc_off = off;
			o.fc_len = len;
			NEED(step(h, i));
		} else {
			o.fmt = PAMD_IDFMT_CASAVA_1_4;
			NEED(chunk_str(h, i, off, len));
			o.instr_len = len;
			NEED(step(h, i));
		}
		NEED(chunk_int(h, i, o.lane));
		NEED(step(h, i));
		NEED(chunk_int(h, i, o.tile));
		NEED(step(h, i));
		NEED(chunk_int(h, i, o.x));
		NEED(step(h, i));
		NEED(chunk_int(h, i, o.y));
		NEED(step(h, i));
		if (h.at(i - 1) == '#') {
			NEED(chunk_tag(h, i, off, len));
			o.tag_off = off;
			o.tag_len = len;
			NEED(step(h, i));
		}
		NEED(policy_ok(o.tag_len, policy));
		NEED(chunk_int(h, i, value));
		return value;
	}
	int mate = 0;
	o.fmt = PAMD_IDFMT_CASAVA_1_7;
	NEED(chunk_str(h, i, off, len));
	o.instr_len = len;
	NEED(step(h, i));
	NEED(chunk_str(h, i, off, len));
	o.run_off = off;
	o.run_len = len;
	NEED(step(h, i));
	NEED(chunk_str(h, i, off, len));
	o.fc_off = off;
	o.fc_len = len;
	NEED(step(h, i));
	NEED(chunk_int(h, i, o.lane));
	NEED(step(h, i));
	NEED(chunk_int(h, i, o.tile));
	NEED(step(h, i));
	NEED(chunk_int(h, i, o.x));
	NEED(step(h, i));
	NEED(chunk_int(h, i, o.y));
	NEED(step(h, i));
	NEED(chunk_int(h, i, mate));
	NEED(step(h, i));
	NEED(h.at(i) != 0);	// "filtered": any chunk, but not the end of the string
	while (!ends_chunk(h.at(i)))
		i++;
	NEED(step(h, i));
	NEED(chunk_int(h, i, value));	// control bits
	NEED(step(h, i));
	NEED(chunk_tag(h, i, off, len));
	o.tag_off = off;
	o.tag_len = len;
	NEED(policy_ok(o.tag_len, policy));
	return mate;
}

#undef NEED

__device__ bool span_equal(Str &a, unsigned aoff, unsigned alen, Str &b, unsigned boff, unsigned blen) {
	if (alen != blen)
		return false;
	for (unsigned k = 0; k < alen; k++)
		if (a.at(aoff + k) != b.at(boff + k))
			return false;
	return true;
}

// panda_seqid_equal (seqid.c:96-101) on the parsed spans
__device__ bool ids_equal(Str &f, const IdOut &a, Str &r, const IdOut &b) {
	if (a.lane != b.lane || a.tile != b.tile || a.x != b.x || a.y != b.y)
		return false;
	if ((a.sra >= 0) != (b.sra >= 0))
		return false;	// "<E|S>RR<n>" can never equal a parsed instrument chunk: such a chunk would itself be read as SRA
	if (a.sra >= 0) {
		if (a.sra != b.sra || f.at(0) != r.at(0))
			return false;
	} else if (!span_equal(f, 0, a.instr_len, r, 0, b.instr_len)) {
		return false;
	}
	return span_equal(f, a.run_off, a.run_len, r, b.run_off, b.run_len) && span_equal(f, a.fc_off, a.fc_len, r, b.fc_off, b.fc_len) && span_equal(f, a.tag_off, a.tag_len, r, b.tag_off, b.tag_len);
}

// ---- ASCII -> nt ------------------------------------------------------------------------------------
// iupac_forward (nt.c:46-79) as 32 four-bit entries indexed by (c & 0x1F): the IUPAC meaning of each
// letter, U as T, X as N, everything else 0.  The reverse table (nt.c:81-114) is the same sets
// complemented base by base.
constexpr unsigned nt_set(const char *s) {
	unsigned v = 0;
	for (; *s; s++)
		v |= *s == 'A' ? 1u : (*s == 'C' ? 2u : (*s == 'G' ? 4u : 8u));
	return v;
}

constexpr u64 nt_table(int half) {
	const char *sets[32] = { "", "A", "CGT", "C", "AGT", "", "", "G", "ACT", "", "", "GT", "", "AC", "ACGT", "",
				 "", "", "AG", "CG", "T", "T", "ACG", "AT", "ACGT", "CT", "", "", "", "", "", "" };
	u64 v = 0;
	for (int k = 0; k < 16; k++)
		v |= (u64) nt_set(sets[half * 16 + k]) << (4 * k);
	return v;
}

constexpr u64 NT_LO = nt_table(0), NT_HI = nt_table(1);

constexpr u64 ascii_table(int half) {	// ntchar, nt.c:25
	const char *letters = "NACMGRSVTWYHKDBN";
	u64 v = 0;
	for (int k = 0; k < 8; k++)
		v |= (u64) (unsigned char) letters[half * 8 + k] << (8 * k);
	return v;
}

constexpr u64 ASCII_LO = ascii_table(0), ASCII_HI = ascii_table(1);

__device__ __forceinline__ unsigned nt_forward(unsigned c) {
	const unsigned idx = c & 31u;
	return (unsigned) ((idx < 16u ? NT_LO : NT_HI) >> (4u * (idx & 15u))) & 15u;
}

__device__ __forceinline__ unsigned nt_complement(unsigned v) {
	return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3);
}

// four bytes of a text starting at byte offset a (any alignment), from two aligned words
__device__ __forceinline__ unsigned load4(const uint8_t *text, u64 bytes, unsigned a) {
	const unsigned a0 = a & ~3u;
	const unsigned lo = *reinterpret_cast<const unsigned *>(text + a0);
	unsigned hi = 0;
	if ((a & 3u) != 0 && (u64) a0 + 4u < bytes)
		hi = *reinterpret_cast<const unsigned *>(text + a0 + 4u);
	return __builtin_amdgcn_alignbyte(hi, lo, a & 3u);
}

struct ParseParams {
	TextRef f, r;
	u64 n_records;
	uint8_t *frows, *rrows;	// pamd_qual rows
	unsigned fstride, rstride;	// bases
	uint16_t *flen, *rlen;
	pamd_seqid *ids;
	uint8_t *flags;		// one byte per record: 0 pair, 1 dropped, else 16 + PAMD_FQ_* (+128: reverse read)
	u64 *first_bad;		// atomicMin: first record the reader fails on
	u64 *first_u64;		// atomicMin: first record that sets seen_under_64
	int qualmin;
	int policy;
};

// One read: its three lines -> one row.  Returns 0 or the PAMD_FQ_* code read_seq() fails with
// (fastq.c:45-107).  Wave-uniform arguments and result.
__device__ int convert_read(const TextRef &T, bool complement, int qualmin, unsigned s_seq, unsigned l_seq, bool seq_long, unsigned s_plus, unsigned l_plus, bool plus_long, unsigned s_q, unsigned l_q, bool q_long, uint8_t *row, unsigned stride, int lane, unsigned &len_out, bool &under64) {
	len_out = 0;
	if (seq_long)
		return PAMD_FQ_EOF;
	// sequence line: the first PAMD_MAX_LEN characters, up to the first NUL (fastq.c:62)
	unsigned nts[2] = { 0u, 0u };	// four nt codes per step, one byte each
	unsigned eff = 0xFFFFFFFFu, bad = 0xFFFFFFFFu;
#pragma unroll
	for (int it = 0; it < 2; it++) {
		const unsigned idx = 256u * it + 4u * (unsigned) lane;
		unsigned w = 0;
		if (idx < l_seq)
			w = load4(T.text, T.bytes, s_seq + idx);
		unsigned packed = 0, zero_at = 4, bad_at = 4;
#pragma unroll
		for (int k = 3; k >= 0; k--) {
			const unsigned c = idx + k < l_seq ? (w >> (8 * k)) & 0xFFu : 0u;
			unsigned nt = nt_forward(c);
			if (complement)
				nt = nt_complement(nt);
			packed |= nt << (8 * k);
			zero_at = c == 0 ? (unsigned) k : zero_at;
			bad_at = (c != 0 && nt == 0) ? (unsigned) k : bad_at;
		}
		nts[it] = packed;
		const u64 zb = __ballot(zero_at < 4);
		if (zb != 0 && eff == 0xFFFFFFFFu) {
			const int l0 = __builtin_ctzll(zb);
			eff = 256u * it + 4u * l0 + (unsigned) __builtin_amdgcn_readlane((int) zero_at, l0);
		}
		const u64 bb = __ballot(bad_at < 4);
		if (bb != 0 && bad == 0xFFFFFFFFu) {
			const int l0 = __builtin_ctzll(bb);
			bad = 256u * it + 4u * l0 + (unsigned) __builtin_amdgcn_readlane((int) bad_at, l0);
		}
	}
	const unsigned pos = eff < PAMD_MAX_LEN ? eff : PAMD_MAX_LEN;
	if (bad < pos)
		return PAMD_FQ_BADNT;
	// '+' line (fastq.c:68-83)
	if (plus_long)
		return PAMD_FQ_EOF;
	{
		unsigned c = 0;
		if (l_plus > 0)
			c = load4(T.text, T.bytes, s_plus) & 0xFFu;
		c = (unsigned) uni((int) c);
		if (c != '+')
			return nt_forward(c) != 0 ? PAMD_FQ_READLEN : PAMD_FQ_BADSEQ;
	}
	// quality line: every character up to the first NUL counts (fastq.c:89-94)
	if (q_long)
		return PAMD_FQ_EOF;
	unsigned quals0 = 0u, quals1 = 0u;
	unsigned qeff = 0xFFFFFFFFu;
	bool u64seen = false;
	for (unsigned base = 0; base < l_q || base < 512u; base += 256u) {
		const unsigned idx = base + 4u * (unsigned) lane;
		unsigned w = 0;
		if (idx < l_q)
			w = load4(T.text, T.bytes, s_q + idx);
		unsigned packed = 0, zero_at = 4;
		bool low = false;
#pragma unroll
		for (int k = 3; k >= 0; k--) {
			const unsigned c = idx + k < l_q ? (w >> (8 * k)) & 0xFFu : 0u;
			const int v = (int) (int8_t) c;
			// TOINDEX, fastq.c:44 (above qualmin + 46 the result is 46 - qualmin: the reference's own quirk)
			const int q = v < qualmin ? 0 : ((v > qualmin + 46 ? 46 : v) - qualmin);
			packed |= ((unsigned) q & 0xFFu) << (8 * k);
			zero_at = c == 0 ? (unsigned) k : zero_at;
		}
		// characters before this lane's first NUL that are below 64
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const unsigned c = (w >> (8 * k)) & 0xFFu;
			low |= (unsigned) k < zero_at && idx + k < l_q && (int) (int8_t) c < 64;
		}
		if (base == 0u)
			quals0 = packed;
		else if (base == 256u)
			quals1 = packed;
		const u64 zb = __ballot(zero_at < 4);
		const u64 lowb = __ballot(low);
		if (zb != 0) {
			const int l0 = __builtin_ctzll(zb);
			qeff = base + 4u * l0 + (unsigned) __builtin_amdgcn_readlane((int) zero_at, l0);
			u64seen |= (lowb & ((2ull << l0) - 1ull)) != 0;
			break;
		}
		u64seen |= lowb != 0;
	}
	under64 |= u64seen;
	if (qeff != pos)
		return PAMD_FQ_NOQUAL;
	if (pos > stride)
		return PAMD_FQ_ROW_OVERFLOW;
	len_out = pos;
	// the row: (nt, qual) pairs, zeros past the read
#pragma unroll
	for (int it = 0; it < 2; it++) {
		const unsigned idx = 256u * it + 4u * (unsigned) lane;
		if (idx < stride) {
			const unsigned n4 = nts[it], q4 = it == 0 ? quals0 : quals1;
			unsigned o[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
				o[k] = idx + k < pos ? (((n4 >> (8 * k)) & 0xFFu) | (((q4 >> (8 * k)) & 0xFFu) << 8)) : 0u;
			*reinterpret_cast<uint2 *>(row + 2u * idx) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
		}
	}
	return 0;
}

__global__ __launch_bounds__(256) void fq_parse_kernel(const ParseParams P) {
	const int lane = threadIdx.x & 63;
	const u64 wave = ((u64) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const u64 rec0 = wave * 64ull;
	if (rec0 >= P.n_records)
		return;
	const u64 rec = rec0 + (u64) lane;
	const bool have = rec < P.n_records;
	// ---- this lane's record: line spans ----
	unsigned fs[4], fl[4], rs[4], rl[4];
	bool flong[4], rlong[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		fs[k] = fl[k] = rs[k] = rl[k] = 0;
		flong[k] = rlong[k] = false;
	}
	if (have) {
#pragma unroll
		for (int t = 0; t < 2; t++) {
			const TextRef &T = t == 0 ? P.f : P.r;
			unsigned prev = rec == 0 ? 0u : T.line_end[4 * rec - 1] + 1u;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const unsigned end = T.line_end[4 * rec + k];
				unsigned len = end - prev;
				const bool toolong = len >= PAMD_LINE_MAX;	// the '\n' must lie inside the line reader's buffer, linebuf.c:68
				if (len > 0 && T.text[end - 1] == '\r')	// DOS line ends, linebuf.c:81-84
					len--;
				if (t == 0) {
					fs[k] = prev;
					fl[k] = len;
					flong[k] = toolong;
				} else {
					rs[k] = prev;
					rl[k] = len;
					rlong[k] = toolong;
				}
				prev = end + 1u;
			}
		}
	}
	// ---- headers, one record per lane (fastq.c:123-144) ----
	int err = 0;
	IdOut fid;
	fid.instr_len = fid.run_off = fid.run_len = fid.fc_off = fid.fc_len = fid.tag_off = fid.tag_len = 0;
	fid.lane = fid.tile = fid.x = fid.y = 0;
	fid.sra = -1;
	fid.fmt = 0;
	int fdir = 0;
	if (have) {
		Str hf = { P.f.text, fs[0] + 1u, fl[0] > 0 ? fl[0] - 1u : 0u, 0xFFFFFFFFu, 0u };
		Str hr = { P.r.text, rs[0] + 1u, rl[0] > 0 ? rl[0] - 1u : 0u, 0xFFFFFFFFu, 0u };
		IdOut rid;
		if (flong[0]) {
			err = PAMD_FQ_LONG_HEADER;
		} else if (fl[0] == 0 || (fdir = parse_header(hf, P.policy, fid)) == 0) {
			err = PAMD_FQ_BADID_FWD;	// an empty header line fails too (the reference reads past its terminator)
		} else if (rlong[0]) {
			err = PAMD_FQ_LONG_HEADER;
		} else {
			const int rdir = rl[0] == 0 ? 0 : parse_header(hr, P.policy, rid);
			if (rdir == 0)
				err = PAMD_FQ_BADID_REV;
			else if (!ids_equal(hf, fid, hr, rid) || (fid.fmt != PAMD_IDFMT_SRA && fid.fmt != PAMD_IDFMT_EBI_SRA && rdir == fdir))
				err = PAMD_FQ_NOTPAIRED;
		}
	}
	bool lane_u64 = have && err == 0 && fid.fmt == PAMD_IDFMT_CASAVA_1_7;	// fastq.c:141-144
	int in_rev = 0;
	unsigned my_flen = 0, my_rlen = 0;
	// ---- reads, one record per wave (fastq.c:145-154) ----
	const int nrec = (int) (P.n_records - rec0 < 64ull ? P.n_records - rec0 : 64ull);
	for (int j = 0; j < nrec; j++) {
		if (__builtin_amdgcn_readlane(err, j) != 0)
			continue;
		unsigned a[7], b[7];
#pragma unroll
		for (int k = 1; k < 4; k++) {
			a[k] = (unsigned) __builtin_amdgcn_readlane((int) fs[k], j);
			a[k + 3] = (unsigned) __builtin_amdgcn_readlane((int) fl[k], j);
			b[k] = (unsigned) __builtin_amdgcn_readlane((int) rs[k], j);
			b[k + 3] = (unsigned) __builtin_amdgcn_readlane((int) rl[k], j);
		}
		const bool fL1 = __builtin_amdgcn_readlane((int) flong[1], j), fL2 = __builtin_amdgcn_readlane((int) flong[2], j), fL3 = __builtin_amdgcn_readlane((int) flong[3], j);
		const bool rL1 = __builtin_amdgcn_readlane((int) rlong[1], j), rL2 = __builtin_amdgcn_readlane((int) rlong[2], j), rL3 = __builtin_amdgcn_readlane((int) rlong[3], j);
		unsigned flen_j = 0, rlen_j = 0;
		bool low64 = false;
		const u64 recj = rec0 + (u64) j;
		int code = convert_read(P.f, false, P.qualmin, a[1], a[4], fL1, a[2], a[5], fL2, a[3], a[6], fL3, P.frows + recj * (u64) P.fstride * 2ull, P.fstride, lane, flen_j, low64);
		int rev_fail = 0;
		if (code == 0) {
			code = convert_read(P.r, true, P.qualmin, b[1], b[4], rL1, b[2], b[5], rL2, b[3], b[6], rL3, P.rrows + recj * (u64) P.rstride * 2ull, P.rstride, lane, rlen_j, low64);
			rev_fail = code != 0;
		}
		if (lane == j) {
			err = code;
			in_rev = rev_fail;
			my_flen = code == 0 ? flen_j : 0u;
			my_rlen = code == 0 ? rlen_j : 0u;
			lane_u64 |= low64;
		}
	}
	// ---- per-record outputs ----
	if (have) {
		const unsigned flag = err != 0 ? (16u + (unsigned) err) | (in_rev ? 128u : 0u) : (my_flen == 0 ? 1u : 0u);
		P.flen[rec] = (uint16_t) my_flen;
		P.rlen[rec] = (uint16_t) my_rlen;
		P.flags[rec] = (uint8_t) flag;
		const u64 hdr = (u64) fs[0] + 1ull;
		uint4 *dst = reinterpret_cast<uint4 *>(&P.ids[rec]);
		dst[0] = make_uint4((unsigned) hdr, (unsigned) (hdr >> 32), (unsigned) fid.lane, (unsigned) fid.tile);
		dst[1] = make_uint4((unsigned) fid.x, (unsigned) fid.y, (unsigned) fid.sra, fid.instr_len | (fid.run_off << 16));
		dst[2] = make_uint4(fid.run_len | (fid.fc_off << 16), fid.fc_len | (fid.tag_off << 16), fid.tag_len | ((unsigned) fid.fmt << 16) | (((unsigned) fdir & 0xFFu) << 24), flag);
	}
	const u64 badb = __ballot(have && err != 0);
	if (badb != 0 && lane == 0)
		atomicMin(P.first_bad, rec0 + (u64) __builtin_ctzll(badb));
	const u64 ub = __ballot(lane_u64);
	if (ub != 0 && lane == 0)
		atomicMin(P.first_u64, rec0 + (u64) __builtin_ctzll(ub));
}

struct FinalOut {	// device words the host reads back
	u64 first_bad, first_u64, n_dropped, fwd_consumed, rev_consumed;
	unsigned max_flen, max_rlen, stop_flag, pad;
};

__global__ __launch_bounds__(256) void fq_final_kernel(const ParseParams P, FinalOut *out) {
	const u64 n_good = out->first_bad < P.n_records ? out->first_bad : P.n_records;
	const u64 tid = (u64) blockIdx.x * blockDim.x + threadIdx.x, nt = (u64) gridDim.x * blockDim.x;
	unsigned dropped = 0, mf = 0, mr = 0;
	for (u64 i = tid; i < n_good; i += nt) {
		dropped += P.flags[i] == 1 ? 1u : 0u;
		const unsigned a = P.flen[i], b = P.rlen[i];
		mf = a > mf ? a : mf;
		mr = b > mr ? b : mr;
	}
	for (int o = 32; o > 0; o >>= 1) {
		dropped += __shfl_xor(dropped, o);
		const unsigned a = __shfl_xor(mf, o), b = __shfl_xor(mr, o);
		mf = a > mf ? a : mf;
		mr = b > mr ? b : mr;
	}
	if ((threadIdx.x & 63) == 0) {
		if (dropped)
			atomicAdd(&out->n_dropped, (u64) dropped);
		atomicMax(&out->max_flen, mf);
		atomicMax(&out->max_rlen, mr);
	}
	if (tid == 0) {
		out->stop_flag = n_good < P.n_records ? P.flags[n_good] : 0u;
		out->fwd_consumed = P.n_records > 0 ? (u64) P.f.line_end[4 * P.n_records - 1] + 1ull : 0ull;
		out->rev_consumed = P.n_records > 0 ? (u64) P.r.line_end[4 * P.n_records - 1] + 1ull : 0ull;
	}
}

// ---- result formatting --------------------------------------------------------------------------------
struct FormatParams {
	u64 n_pairs;
	const pamd_pair_result *results;
	const uint8_t *seq_nt;
	const double *seq_p;
	const pamd_seqid *ids;
	const uint8_t *hdr_text;
	uint8_t *out;
	u64 out_capacity;
	unsigned *lens;		// text bytes per pair
	u64 *wg_sums;		// per workgroup of 1024 pairs; exclusive prefix after the scan
	u64 *total;
	u64 *fixups;		// (offset, pair) records
	unsigned *n_fixups;
	unsigned max_fixups;
	const double *score;	// qual_score[47] (device)
	int fastq;
};

__device__ __forceinline__ unsigned dec_len(int v) {	// characters printf("%d") writes
	unsigned a = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	unsigned n = v < 0 ? 2u : 1u;
	while (a >= 10u) {
		a /= 10u;
		n++;
	}
	return n;
}

struct IdView {
	u64 hdr;
	int lane, tile, x, y, sra;
	unsigned instr_len, run_off, run_len, fc_off, fc_len, tag_off, tag_len, flag;
};

__device__ __forceinline__ IdView load_id(const pamd_seqid *ids, u64 i) {
	const uint4 *src = reinterpret_cast<const uint4 *>(&ids[i]);
	const uint4 a = src[0], b = src[1], c = src[2];
	IdView v;
	v.hdr = (u64) a.x | ((u64) a.y << 32);
	v.lane = (int) a.z;
	v.tile = (int) a.w;
	v.x = (int) b.x;
	v.y = (int) b.y;
	v.sra = (int) b.z;
	v.instr_len = b.w & 0xFFFFu;
	v.run_off = b.w >> 16;
	v.run_len = c.x & 0xFFFFu;
	v.fc_off = c.x >> 16;
	v.fc_len = c.y & 0xFFFFu;
	v.tag_off = c.y >> 16;
	v.tag_len = c.z & 0xFFFFu;
	v.flag = c.w & 0xFFu;
	return v;
}

// characters of "instrument:run:flowcell:lane:tile:x:y:tag" (seqid.c:115)
__device__ __forceinline__ unsigned id_text_len(const IdView &v) {
	const unsigned instr = v.sra >= 0 ? 3u + dec_len(v.sra) : v.instr_len;
	return instr + 1u + v.run_len + 1u + v.fc_len + 1u + dec_len(v.lane) + 1u + dec_len(v.tile) + 1u + dec_len(v.x) + 1u + dec_len(v.y) + 1u + v.tag_len;
}

__device__ __forceinline__ bool pair_prints(const FormatParams &P, u64 i, unsigned &seq_len, unsigned &status) {
	const uint4 *r = reinterpret_cast<const uint4 *>(&P.results[i]);
	const uint4 b = r[1], c = r[2];
	seq_len = b.z & 0xFFFFu;
	status = (c.y >> 16) & 0xFFu;
	const bool has_row = (b.x & b.y) != 0xFFFFFFFFu;	// ~0 = the output arena was too small for this pair's row
	return status == PAMD_ST_OK && seq_len > 0 && has_row;	// output.c:89,107
}

__global__ __launch_bounds__(1024) void fmt_len_kernel(const FormatParams P) {
	__shared__ u64 wsum[16];
	const u64 i = (u64) blockIdx.x * 1024ull + threadIdx.x;
	unsigned len = 0;
	if (i < P.n_pairs) {
		unsigned seq_len, status;
		if (pair_prints(P, i, seq_len, status)) {
			const IdView v = load_id(P.ids, i);
			if (v.flag == 0) {
				len = 1u + id_text_len(v) + 9u + 1u + seq_len + 1u;	// lead, id, ";d.dddddd", '\n', bases, '\n'
				if (P.fastq)
					len += 2u + seq_len + 1u;	// "+\n", qualities, '\n'
			}
		}
		P.lens[i] = len;
	}
	u64 s = len;
	for (int o = 32; o > 0; o >>= 1)
		s += __shfl_xor(s, o);
	if ((threadIdx.x & 63) == 0)
		wsum[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 t = 0;
		for (int k = 0; k < 16; k++)
			t += wsum[k];
		P.wg_sums[blockIdx.x] = t;
	}
}

__global__ __launch_bounds__(1024) void fmt_scan_kernel(u64 *c, unsigned n, u64 *total) {
	__shared__ u64 part[1024];
	const unsigned tid = threadIdx.x;
	const unsigned per = (n + 1023u) / 1024u;
	const unsigned lo = tid * per < n ? tid * per : n;
	const unsigned hi = lo + per < n ? lo + per : n;
	u64 s = 0;
	for (unsigned i = lo; i < hi; i++)
		s += c[i];
	part[tid] = s;
	__syncthreads();
	for (unsigned off = 1; off < 1024; off <<= 1) {
		const u64 v = tid >= off ? part[tid - off] : 0ull;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	u64 run = part[tid] - s;
	for (unsigned i = lo; i < hi; i++) {
		const u64 t = c[i];
		c[i] = run;
		run += t;
	}
	if (tid == 1023)
		*total = part[1023];
}

__device__ __forceinline__ unsigned put_dec(uint8_t *dst, int v) {	// printf("%d")
	unsigned a = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	const unsigned n = dec_len(v);
	unsigned k = n;
	do {
		dst[--k] = (uint8_t) ('0' + a % 10u);
		a /= 10u;
	} while (a != 0u);
	if (v < 0)
		dst[0] = '-';
	return n;
}

// panda_result_phred (nt.c:126-150)
__device__ __forceinline__ int result_phred_io(const double *score, double p) {
	int lower = 0, upper = 46;
	if (p <= score[0])
		return 1;
	while (lower < upper) {
		const int mid = lower + (upper - lower) / 2;
		const double s = score[mid];
		if (s == p)
			return mid;
		if (mid == lower)
			return lower;
		else if (s > p)
			upper = mid;
		else if (s < p)
			lower = mid + 1;
	}
	return lower;
}

__global__ __launch_bounds__(256) void fmt_write_kernel(const FormatParams P) {
	// one workgroup = 4 waves = 256 pairs (a quarter of a fmt_len_kernel group); each wave owns 64 pairs
	__shared__ double score[48];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x < 47)
		score[threadIdx.x] = P.score[threadIdx.x];
	__syncthreads();
	const u64 i0 = (u64) blockIdx.x * 256ull + (u64) wv * 64ull;
	if (i0 >= P.n_pairs)
		return;
	const u64 i = i0 + (u64) lane;
	// ---- offsets: group prefix + prefix inside the group of 1024 ----
	unsigned len = i < P.n_pairs ? P.lens[i] : 0u;
	u64 before = 0;	// text bytes of the pairs of this 1024-group that precede this wave's 64
	{
		const u64 g0 = i0 & ~1023ull;
		for (u64 k = g0 + (u64) lane; k < i0; k += 64ull)
			before += P.lens[k];
		for (int o = 32; o > 0; o >>= 1)
			before += __shfl_xor(before, o);
	}
	unsigned incl = len;
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned v = __shfl_up(incl, o);
		if (lane >= o)
			incl += v;
	}
	const u64 my_off = P.wg_sums[i0 >> 10] + before + (u64) (incl - len);
	// ---- header lines, one pair per lane, byte by byte straight into the output (L2 gathers the
	// partial lines; the bases that follow complete them) ----
	unsigned hlen = 0;
	unsigned seq_len = 0;
	u64 seq_off = 0;
	if (len != 0 && my_off + len > P.out_capacity)
		len = 0;	// does not fit: reported by the host from the total
	if (len != 0) {
		const IdView v = load_id(P.ids, i);
		const uint4 *r = reinterpret_cast<const uint4 *>(&P.results[i]);
		const uint4 ra = r[0], rb = r[1];
		seq_off = (u64) rb.x | ((u64) rb.y << 32);
		seq_len = rb.z & 0xFFFFu;
		uint8_t *h = P.out + my_off;
		unsigned n = 0;
		h[n++] = P.fastq ? '@' : '>';
		const uint8_t *src = P.hdr_text + v.hdr;
		for (unsigned k = 0; k < v.instr_len; k++)
			h[n++] = src[k];
		if (v.sra >= 0)
			n += put_dec(h + n, v.sra);
		h[n++] = ':';
		for (unsigned k = 0; k < v.run_len; k++)
			h[n++] = src[v.run_off + k];
		h[n++] = ':';
		for (unsigned k = 0; k < v.fc_len; k++)
			h[n++] = src[v.fc_off + k];
		h[n++] = ':';
		n += put_dec(h + n, v.lane);
		h[n++] = ':';
		n += put_dec(h + n, v.tile);
		h[n++] = ':';
		n += put_dec(h + n, v.x);
		h[n++] = ':';
		n += put_dec(h + n, v.y);
		h[n++] = ':';
		for (unsigned k = 0; k < v.tag_len; k++)
			h[n++] = src[v.tag_off + k];
		// ";%f" of exp(quality) (output.c:93,111): six decimals of a value in [0, 1]
		union {
			double d;
			unsigned u[2];
		} q;
		q.u[0] = ra.x;
		q.u[1] = ra.y;
		const double e = exp(q.d);
		const double s = e * 1e6;
		const double fl = floor(s);
		const double frac = s - fl;
		unsigned m = (unsigned) fl + (frac > 0.5 ? 1u : 0u);
		if (fabs(frac - 0.5) < 4e-9) {
			// too close to a rounding boundary to trust this device's exp(): the host re-prints it
			const unsigned slot = atomicAdd(P.n_fixups, 1u);
			if (slot < P.max_fixups) {
				P.fixups[2 * slot] = my_off + n + 1;
				P.fixups[2 * slot + 1] = i;
			}
		}
		h[n++] = ';';
		h[n++] = (uint8_t) ('0' + m / 1000000u);
		h[n++] = '.';
		m %= 1000000u;
		unsigned div = 100000u;
		for (int k = 0; k < 6; k++) {
			h[n++] = (uint8_t) ('0' + m / div);
			m %= div;
			div /= 10u;
		}
		h[n++] = '\n';
		hlen = n;
	}
	// ---- bases (and qualities), one pair per wave ----
	const int npairs = (int) (P.n_pairs - i0 < 64ull ? P.n_pairs - i0 : 64ull);
	for (int j = 0; j < npairs; j++) {
		const unsigned lj = (unsigned) __builtin_amdgcn_readlane((int) len, j);
		if (lj == 0)
			continue;
		const unsigned hj = (unsigned) __builtin_amdgcn_readlane((int) hlen, j);
		const unsigned sl = (unsigned) __builtin_amdgcn_readlane((int) seq_len, j);
		const u64 off = (u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) my_off, j) | ((u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (my_off >> 32), j) << 32);
		const u64 so = (u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) seq_off, j) | ((u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (seq_off >> 32), j) << 32);
		uint8_t *dst = P.out + off + hj;
		const uint8_t *nt = P.seq_nt + so;
		for (unsigned k = lane; k < sl; k += 64) {
			const unsigned v = nt[k];
			// panda_nt_to_ascii, nt.c:167-173 ("NACMGRSVTWYHKDBN"; anything outside 0..15 prints N)
			dst[k] = v > 15u ? (uint8_t) 'N' : (uint8_t) ((v < 8u ? ASCII_LO >> (8u * v) : ASCII_HI >> (8u * (v - 8u))) & 0xFFu);
		}
		if (lane == 0)
			dst[sl] = '\n';
		if (P.fastq) {
			dst += sl + 1;
			if (lane == 0) {
				dst[0] = '+';
				dst[1] = '\n';
				dst[2 + sl] = '\n';
			}
			const double *p = P.seq_p + so;
			for (unsigned k = lane; k < sl; k += 64)
				dst[2 + k] = (uint8_t) (33 + result_phred_io(score, p[k]));
		}
	}
}

inline unsigned grid_for(u64 waves_wanted, unsigned waves_per_block, unsigned cap_blocks) {
	u64 blocks = (waves_wanted + waves_per_block - 1) / waves_per_block;
	if (blocks < 1)
		blocks = 1;
	return (unsigned) (blocks > cap_blocks ? cap_blocks : blocks);
}

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes) {
		if (bytes <= cap)
			return PAMD_OK;
		if (p)
			(void) hipFree(p);
		p = nullptr;
		cap = 0;
		const size_t want = bytes + bytes / 4 + 4096;
		if (hipMalloc(&p, want) != hipSuccess) {
			pamd_set_error("hipMalloc(%zu) failed", want);
			return PAMD_ENOMEM;
		}
		cap = want;
		return PAMD_OK;
	}
	void release() {
		if (p)
			(void) hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

}	// namespace

struct pamd_fastq_parser {
	int device = 0;
	int qualmin = 33;
	int policy = PAMD_TAG_OPTIONAL;
	DevBuf counts_f, counts_r, lines_f, lines_r, flags, small;
	hipEvent_t ev[3] = { nullptr, nullptr, nullptr };
	float index_ms = -1.f, parse_ms = -1.f;
};

struct pamd_formatter {
	int device = 0;
	DevBuf lens, sums, small, fix;
	double *d_score = nullptr;
	hipEvent_t ev[2] = { nullptr, nullptr };
	float ms = -1.f;
};

static int io_pick_device(int device, const char *what) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		pamd_set_error("no HIP device available (%s); %s has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e), what);
		return PAMD_ENODEVICE;
	}
	if (device < 0 || device >= n) {
		pamd_set_error("device %d out of range (0..%d)", device, n - 1);
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(device));
	return PAMD_OK;
}

extern "C" {

pamd_fastq_parser *pamd_fastq_parser_create(int device, int qualmin, int policy) {
	if (io_pick_device(device, "the FASTQ parser") != PAMD_OK)
		return nullptr;
	if (qualmin < 0 || qualmin > 127 || policy < PAMD_TAG_PRESENT || policy > PAMD_TAG_OPTIONAL) {
		pamd_set_error("pamd_fastq_parser_create: qualmin %d / policy %d out of range", qualmin, policy);
		return nullptr;
	}
	pamd_fastq_parser *p = new (std::nothrow) pamd_fastq_parser;
	if (p == nullptr)
		return nullptr;
	p->device = device;
	p->qualmin = qualmin;
	p->policy = policy;
	for (int k = 0; k < 3; k++)
		if (hipEventCreate(&p->ev[k]) != hipSuccess) {
			pamd_set_error("hipEventCreate failed");
			pamd_fastq_parser_destroy(p);
			return nullptr;
		}
	return p;
}

void pamd_fastq_parser_destroy(pamd_fastq_parser *p) {
	if (p == nullptr)
		return;
	(void) hipSetDevice(p->device);
	p->counts_f.release();
	p->counts_r.release();
	p->lines_f.release();
	p->lines_r.release();
	p->flags.release();
	p->small.release();
	for (int k = 0; k < 3; k++)
		if (p->ev[k])
			(void) hipEventDestroy(p->ev[k]);
	delete p;
}

int pamd_fastq_parse_device(pamd_fastq_parser *p, const char *fwd_text, size_t fwd_bytes, const char *rev_text, size_t rev_bytes, size_t max_pairs, pamd_qual *fwd_rows, size_t fwd_stride, pamd_qual *rev_rows, size_t rev_stride, uint16_t *flen, uint16_t *rlen, pamd_seqid *ids, pamd_fastq_status *status, void *stream) {
	if (p == nullptr || status == nullptr) {
		pamd_set_error("pamd_fastq_parse_device: null parser or status");
		return PAMD_EINVAL;
	}
	memset(status, 0, sizeof *status);
	if ((fwd_bytes && fwd_text == nullptr) || (rev_bytes && rev_text == nullptr) || fwd_rows == nullptr || rev_rows == nullptr || flen == nullptr || rlen == nullptr || ids == nullptr) {
		pamd_set_error("pamd_fastq_parse_device: null buffer");
		return PAMD_EINVAL;
	}
	if (((uintptr_t) fwd_text | (uintptr_t) rev_text | (uintptr_t) fwd_rows | (uintptr_t) rev_rows | (uintptr_t) ids) & 15u) {
		pamd_set_error("pamd_fastq_parse_device: texts, rows and ids must be 16-byte aligned");
		return PAMD_EINVAL;
	}
	if (fwd_bytes >= (1ull << 32) || rev_bytes >= (1ull << 32)) {
		pamd_set_error("pamd_fastq_parse_device: a text chunk must be smaller than 4 GiB");
		return PAMD_EINVAL;
	}
	if (fwd_stride == 0 || rev_stride == 0 || (fwd_stride & 7u) || (rev_stride & 7u) || fwd_stride > 456 || rev_stride > 456) {
		pamd_set_error("pamd_fastq_parse_device: row strides must be multiples of 8 bases in 8..456");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(p->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	p->index_ms = p->parse_ms = -1.f;
	if (fwd_bytes == 0 || rev_bytes == 0 || max_pairs == 0)
		return PAMD_OK;

	TextRef F, R;
	F.text = reinterpret_cast<const uint8_t *>(fwd_text);
	F.bytes = fwd_bytes;
	F.n_tiles = (unsigned) ((fwd_bytes + TILE_BYTES - 1) / TILE_BYTES);
	R.text = reinterpret_cast<const uint8_t *>(rev_text);
	R.bytes = rev_bytes;
	R.n_tiles = (unsigned) ((rev_bytes + TILE_BYTES - 1) / TILE_BYTES);
	int rc;
	if ((rc = p->counts_f.ensure((size_t) F.n_tiles * 4)) != PAMD_OK || (rc = p->counts_r.ensure((size_t) R.n_tiles * 4)) != PAMD_OK || (rc = p->small.ensure(256)) != PAMD_OK)
		return rc;
	F.tile_counts = static_cast<unsigned *>(p->counts_f.p);
	R.tile_counts = static_cast<unsigned *>(p->counts_r.p);
	F.line_end = R.line_end = nullptr;
	unsigned *d_totals = static_cast<unsigned *>(p->small.p);	// [0..1] line totals; FinalOut at +64 bytes
	FinalOut *d_final = reinterpret_cast<FinalOut *>(static_cast<uint8_t *>(p->small.p) + 64);

	HIP_TRY(hipEventRecord(p->ev[0], st));
	const unsigned tiles = F.n_tiles + R.n_tiles;
	hipLaunchKernelGGL(fq_count_kernel, dim3(grid_for(tiles, 4, 65535)), dim3(256), 0, st, F, R);
	hipLaunchKernelGGL(fq_scan_kernel, dim3(2), dim3(1024), 0, st, F.tile_counts, F.n_tiles, R.tile_counts, R.n_tiles, d_totals);
	unsigned totals[2] = { 0, 0 };
	HIP_TRY(hipMemcpyAsync(totals, d_totals, sizeof totals, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if ((rc = p->lines_f.ensure((size_t) totals[0] * 4 + 16)) != PAMD_OK || (rc = p->lines_r.ensure((size_t) totals[1] * 4 + 16)) != PAMD_OK)
		return rc;
	F.line_end = static_cast<unsigned *>(p->lines_f.p);
	R.line_end = static_cast<unsigned *>(p->lines_r.p);
	hipLaunchKernelGGL(fq_index_kernel, dim3(grid_for(tiles, 4, 65535)), dim3(256), 0, st, F, R);
	HIP_TRY(hipEventRecord(p->ev[1], st));

	u64 n_records = (totals[0] < totals[1] ? totals[0] : totals[1]) / 4u;
	if (n_records > max_pairs)
		n_records = max_pairs;
	FinalOut init;
	memset(&init, 0, sizeof init);
	init.first_bad = ~0ull;
	init.first_u64 = ~0ull;
	HIP_TRY(hipMemcpyAsync(d_final, &init, sizeof init, hipMemcpyHostToDevice, st));
	FinalOut fin = init;
	if (n_records > 0) {
		if ((rc = p->flags.ensure((size_t) n_records)) != PAMD_OK)
			return rc;
		ParseParams P;
		P.f = F;
		P.r = R;
		P.n_records = n_records;
		P.frows = reinterpret_cast<uint8_t *>(fwd_rows);
		P.rrows = reinterpret_cast<uint8_t *>(rev_rows);
		P.fstride = (unsigned) fwd_stride;
		P.rstride = (unsigned) rev_stride;
		P.flen = flen;
		P.rlen = rlen;
		P.ids = ids;
		P.flags = static_cast<uint8_t *>(p->flags.p);
		P.first_bad = &d_final->first_bad;
		P.first_u64 = &d_final->first_u64;
		P.qualmin = p->qualmin;
		P.policy = p->policy;
		const u64 waves = (n_records + 63) / 64;
		hipLaunchKernelGGL(fq_parse_kernel, dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st, P);
		hipLaunchKernelGGL(fq_final_kernel, dim3(grid_for((n_records + 63) / 64, 4, 1024)), dim3(256), 0, st, P, d_final);
		HIP_TRY(hipEventRecord(p->ev[2], st));
		HIP_TRY(hipMemcpyAsync(&fin, d_final, sizeof fin, hipMemcpyDeviceToHost, st));
	} else {
		HIP_TRY(hipEventRecord(p->ev[2], st));
	}
	HIP_TRY(hipStreamSynchronize(st));
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("FASTQ kernels: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	(void) hipEventElapsedTime(&p->index_ms, p->ev[0], p->ev[1]);
	(void) hipEventElapsedTime(&p->parse_ms, p->ev[1], p->ev[2]);
	status->n_records = n_records;
	status->n_good = fin.first_bad < n_records ? fin.first_bad : n_records;
	status->n_dropped = fin.n_dropped;
	status->fwd_consumed = fin.fwd_consumed;
	status->rev_consumed = fin.rev_consumed;
	status->stop_code = status->n_good < n_records ? (int) (fin.stop_flag & 0x7Fu) - 16 : PAMD_FQ_END;
	status->stop_in_reverse = status->n_good < n_records ? (int) (fin.stop_flag >> 7) : 0;
	status->seen_under_64 = (fin.first_u64 <= status->n_good && fin.first_u64 < n_records) ? 1 : 0;	// the failing record counts with what it saw before failing
	status->max_flen = fin.max_flen;
	status->max_rlen = fin.max_rlen;
	return PAMD_OK;
}

int pamd_fastq_last_kernel_ms(pamd_fastq_parser *p, float *index_ms, float *parse_ms) {
	if (p == nullptr)
		return PAMD_EINVAL;
	if (index_ms)
		*index_ms = p->index_ms;
	if (parse_ms)
		*parse_ms = p->parse_ms;
	return PAMD_OK;
}

pamd_formatter *pamd_formatter_create(int device) {
	if (io_pick_device(device, "the result formatter") != PAMD_OK)
		return nullptr;
	pamd_formatter *f = new (std::nothrow) pamd_formatter;
	if (f == nullptr)
		return nullptr;
	f->device = device;
	PamdTables *t = new (std::nothrow) PamdTables;
	if (t == nullptr) {
		delete f;
		return nullptr;
	}
	pamd_build_tables(t);
	bool ok = hipMalloc(reinterpret_cast<void **>(&f->d_score), sizeof(double) * 48) == hipSuccess && hipMemcpy(f->d_score, t->score, sizeof(double) * PAMD_NQ, hipMemcpyHostToDevice) == hipSuccess;
	delete t;
	for (int k = 0; ok && k < 2; k++)
		ok = hipEventCreate(&f->ev[k]) == hipSuccess;
	if (!ok) {
		pamd_set_error("pamd_formatter_create: HIP allocation failed");
		pamd_formatter_destroy(f);
		return nullptr;
	}
	return f;
}

void pamd_formatter_destroy(pamd_formatter *f) {
	if (f == nullptr)
		return;
	(void) hipSetDevice(f->device);
	f->lens.release();
	f->sums.release();
	f->small.release();
	f->fix.release();
	if (f->d_score)
		(void) hipFree(f->d_score);
	for (int k = 0; k < 2; k++)
		if (f->ev[k])
			(void) hipEventDestroy(f->ev[k]);
	delete f;
}

int pamd_format_device(pamd_formatter *f, int format, size_t n_pairs, const pamd_pair_result *results, const uint8_t *seq_nt, const double *seq_p, const pamd_seqid *ids, const char *hdr_text, char *out, size_t out_capacity, size_t *out_bytes, uint64_t *fixups, size_t max_fixups, size_t *n_fixups, void *stream) {
	if (f == nullptr || out_bytes == nullptr) {
		pamd_set_error("pamd_format_device: null formatter or out_bytes");
		return PAMD_EINVAL;
	}
	*out_bytes = 0;
	if (n_fixups)
		*n_fixups = 0;
	if (format != PAMD_OUT_FASTA && format != PAMD_OUT_FASTQ) {
		pamd_set_error("pamd_format_device: unknown format %d", format);
		return PAMD_EINVAL;
	}
	if (n_pairs == 0)
		return PAMD_OK;
	if (results == nullptr || seq_nt == nullptr || ids == nullptr || hdr_text == nullptr || out == nullptr || (format == PAMD_OUT_FASTQ && seq_p == nullptr) || (max_fixups && (fixups == nullptr || n_fixups == nullptr))) {
		pamd_set_error("pamd_format_device: null buffer");
		return PAMD_EINVAL;
	}
	if (((uintptr_t) results | (uintptr_t) ids) & 15u) {
		pamd_set_error("pamd_format_device: results and ids must be 16-byte aligned");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(f->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	const unsigned groups = (unsigned) ((n_pairs + 1023) / 1024);
	int rc;
	const size_t fix_cap = max_fixups > 65536 ? 65536 : max_fixups;
	if ((rc = f->lens.ensure(n_pairs * 4)) != PAMD_OK || (rc = f->sums.ensure((size_t) groups * 8)) != PAMD_OK || (rc = f->small.ensure(64)) != PAMD_OK || (rc = f->fix.ensure((fix_cap + 1) * 16)) != PAMD_OK)
		return rc;
	FormatParams P;
	P.n_pairs = n_pairs;
	P.results = results;
	P.seq_nt = seq_nt;
	P.seq_p = seq_p;
	P.ids = ids;
	P.hdr_text = reinterpret_cast<const uint8_t *>(hdr_text);
	P.out = reinterpret_cast<uint8_t *>(out);
	P.out_capacity = out_capacity;
	P.lens = static_cast<unsigned *>(f->lens.p);
	P.wg_sums = static_cast<u64 *>(f->sums.p);
	P.total = static_cast<u64 *>(f->small.p);
	P.n_fixups = reinterpret_cast<unsigned *>(static_cast<uint8_t *>(f->small.p) + 8);
	P.fixups = static_cast<u64 *>(f->fix.p);
	P.max_fixups = (unsigned) fix_cap;
	P.score = f->d_score;
	P.fastq = format == PAMD_OUT_FASTQ;
	HIP_TRY(hipMemsetAsync(f->small.p, 0, 64, st));
	HIP_TRY(hipEventRecord(f->ev[0], st));
	hipLaunchKernelGGL(fmt_len_kernel, dim3(groups), dim3(1024), 0, st, P);
	hipLaunchKernelGGL(fmt_scan_kernel, dim3(1), dim3(1024), 0, st, P.wg_sums, groups, P.total);
	hipLaunchKernelGGL(fmt_write_kernel, dim3((unsigned) ((n_pairs + 255) / 256)), dim3(256), 0, st, P);
	HIP_TRY(hipEventRecord(f->ev[1], st));
	struct {
		u64 total;
		unsigned nfix, pad;
	} back = { 0, 0, 0 };
	HIP_TRY(hipMemcpyAsync(&back, f->small.p, sizeof back, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("format kernels: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	(void) hipEventElapsedTime(&f->ms, f->ev[0], f->ev[1]);
	*out_bytes = (size_t) back.total;
	if (back.total > out_capacity) {
		pamd_set_error("pamd_format_device: the text needs %llu bytes, the buffer holds %zu", (unsigned long long) back.total, out_capacity);
		return PAMD_EINVAL;
	}
	if (back.nfix > 0) {
		if (back.nfix > fix_cap) {
			pamd_set_error("pamd_format_device: %u pairs need a host re-print of their quality field, room for %zu", back.nfix, fix_cap);
			return PAMD_EINVAL;
		}
		HIP_TRY(hipMemcpy(fixups, f->fix.p, (size_t) back.nfix * 16, hipMemcpyDeviceToHost));
		*n_fixups = back.nfix;
	}
	return PAMD_OK;
}

int pamd_format_last_kernel_ms(pamd_formatter *f, float *ms) {
	if (f == nullptr || ms == nullptr)
		return PAMD_EINVAL;
	*ms = f->ms;
	return PAMD_OK;
}

}	// extern "C"
