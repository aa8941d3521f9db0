// pamd_io.hip -- gfx950 kernels and C ABI of include/pandaseq_amd_io.h: FASTQ text -> batch rows,
// and accepted pairs -> FASTA / FASTQ text.  Byte work bounded by HBM bandwidth; no MFMA.
//
// Ingest (replaces fastq.c:45-198 + linebuf.c:57-89 + seqid.c:142-285 + nt.c:46-114):
//   fq_index_kernel   byte offset of every '\n', in order: one pass over the text, 8 KB tiles, one wave per
//                     tile, line counts carried from tile to tile by a chained scan with look-back
//   fq_ids_kernel     headers: two LANES per record (forward / reverse header), a serial character walk
//                     driven by a per-format field program; the pair compares through lane exchanges
//   fq_reads_kernel   reads: 32 records per wave, each converted by the whole WAVE (4 characters per lane
//                     as one word, loaded one record ahead; 8-byte row stores)
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
	unsigned last_word;	// byte offset of the aligned word that holds the last byte
	u64 *tile_state;	// one word per group of 16 tiles for the chained scan (zeroed before the launch)
	unsigned *line_end;	// byte offset of the first line_cap newlines
	unsigned line_cap;
	unsigned n_tiles;
};

// One pass over both texts: every wave takes the next tile (a ticket, so that every tile it may have
// to wait for is already owned by a running wave), counts the tile's newlines, learns how many lines
// precede the tile from its predecessors' published states (chained scan with look-back: a state is
// "aggregate" = this tile's count, or "prefix" = count of this and all earlier tiles), and writes the
// byte offsets of its newlines at their final positions.  The text is read once.
constexpr u64 STATE_AGGREGATE = 1ull << 62, STATE_PREFIX = 2ull << 62, STATE_VALUE = (1ull << 62) - 1ull;

constexpr unsigned INDEX_WAVES = 16;	// waves (8 KB tiles) per workgroup: one chained-scan step per 128 KB of text

__global__ __launch_bounds__(INDEX_WAVES * 64) void fq_index_kernel(const TextRef a, const TextRef b, unsigned *ticket, unsigned *totals) {
	const int lane = threadIdx.x & 63;
	const unsigned wv = threadIdx.x >> 6;
	__shared__ unsigned group_ticket, group_before;
	__shared__ unsigned wave_count[INDEX_WAVES];
	if (threadIdx.x == 0)
		group_ticket = atomicAdd(ticket, 1u);	// groups enter the chain in the order they start running
	__syncthreads();
	const unsigned ga = (a.n_tiles + INDEX_WAVES - 1) / INDEX_WAVES, gb = (b.n_tiles + INDEX_WAVES - 1) / INDEX_WAVES;
	const unsigned g = group_ticket;
	if (g >= ga + gb)
		return;
	const bool second = g >= ga;
	const TextRef &T = second ? b : a;
	const unsigned group = second ? g - ga : g;
	const unsigned n_groups = second ? gb : ga;
	const unsigned tile = group * INDEX_WAVES + wv;
	// newline maps of this wave's 8 KB: 8 steps of 64 lanes x 16 bytes, two 16-bit maps per register
	unsigned maps[TILE_STEPS / 2];
	unsigned cnt = 0;
#pragma unroll
	for (unsigned s = 0; s < TILE_STEPS; s += 2) {
		const u64 off = (u64) tile * TILE_BYTES + (u64) s * 1024u + (u64) lane * 16u;
		const unsigned m0 = nl_mask16(T.text, T.bytes, off), m1 = nl_mask16(T.text, T.bytes, off + 1024u);
		maps[s / 2] = m0 | (m1 << 16);
		cnt += __popc(maps[s / 2]);
	}
	unsigned agg = cnt;
	for (int o = 32; o > 0; o >>= 1)
		agg += __shfl_xor(agg, o);
	if (lane == 0)
		wave_count[wv] = agg;
	__syncthreads();
	// ---- wave 0: lines before this group, from the groups before it ----
	if (wv == 0) {
		unsigned gsum = lane < (int) INDEX_WAVES ? wave_count[lane] : 0u;
		for (int o = 32; o > 0; o >>= 1)
			gsum += __shfl_xor(gsum, o);
		unsigned before = 0;
		if (group != 0) {
			if (lane == 0)
				__hip_atomic_store(&T.tile_state[group], STATE_AGGREGATE | (u64) gsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			int pos = (int) group - 1;	// nearest predecessor not yet accounted for
			unsigned spins = 0;
			for (;;) {
				const int idx = pos - lane;
				u64 st = STATE_PREFIX;	// before the first group: a prefix of zero
				if (idx >= 0)
					st = __hip_atomic_load(&T.tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const u64 ready = __ballot((st >> 62) != 0ull);
				const u64 isprefix = __ballot((st >> 62) == 2ull);
				// usable: the run of ready states from lane 0 up to and including the first prefix
				const int first_gap = ~ready != 0 ? __builtin_ctzll(~ready) : 64;
				const int first_prefix = isprefix != 0 ? __builtin_ctzll(isprefix) : 64;
				if (first_prefix < first_gap) {
					unsigned v = lane <= first_prefix ? (unsigned) (st & STATE_VALUE) : 0u;
					for (int o = 32; o > 0; o >>= 1)
						v += __shfl_xor(v, o);
					before += v;
					break;
				}
				if (first_gap == 64) {	// 64 aggregates, no prefix yet: take them all and look further back
					unsigned v = (unsigned) (st & STATE_VALUE);
					for (int o = 32; o > 0; o >>= 1)
						v += __shfl_xor(v, o);
					before += v;
					pos -= 64;
				} else {
					__builtin_amdgcn_s_sleep(1);	// a predecessor has not published yet
					if (++spins > (1u << 24)) {	// never seen; reported instead of hanging the GPU
						if (lane == 0)
							atomicAdd(&totals[2], 1u);
						break;
					}
				}
			}
		}
		if (lane == 0) {
			__hip_atomic_store(&T.tile_state[group], STATE_PREFIX | (u64) (before + gsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			group_before = before;
			if (group == n_groups - 1)
				totals[second ? 1 : 0] = before + gsum;
		}
	}
	__syncthreads();
	unsigned out = group_before;
	for (unsigned k = 0; k < wv; k++)
		out += wave_count[k];
	// ---- positions ----
	if (out >= T.line_cap)
		return;
#pragma unroll
	for (unsigned s = 0; s < TILE_STEPS; s++) {
		unsigned m = (maps[s / 2] >> (16 * (s & 1))) & 0xFFFFu;
		const unsigned c = __popc(m);
		unsigned incl = c;	// inclusive prefix over lanes
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned v = __shfl_up(incl, o);
			if (lane >= o)
				incl += v;
		}
		unsigned at = out + incl - c;
		const unsigned off = tile * TILE_BYTES + s * 1024u + (unsigned) lane * 16u;
		while (m != 0) {
			const unsigned bit = __builtin_ctz(m);
			m &= m - 1;
			if (at < T.line_cap)
				T.line_end[at] = off + bit;
			at++;
		}
		out += __shfl(incl, 63);
	}
}

// ---- header parsing, one header per lane ----------------------------------------------------------
// A header as the C string the reference's parser sees: the bytes of the line after the '@', then a
// terminator.  Bytes come from aligned words of the text, one word cached in registers.
constexpr unsigned STAGE_WORDS = 24;	// words of each header staged in LDS (96 bytes from the 16-byte group the header starts in)

struct Str {
	const uint8_t *text;	// 4-byte aligned
	unsigned start, len;
	unsigned cached;	// byte offset of the cached word, ~0 = none
	unsigned word;
	const unsigned *lds;	// this lane's column of the staged words (word w at lds[64 * w])
	unsigned base16;	// text offset of staged word 0
	__device__ __forceinline__ int at(unsigned i) {
		if (i >= len)
			return 0;
		const unsigned a = start + i;
		if ((a & ~3u) != cached) {
			cached = a & ~3u;
			const unsigned rel = (cached - base16) >> 2;
			word = rel < STAGE_WORDS ? lds[64u * rel] : *reinterpret_cast<const unsigned *>(text + cached);
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

// panda_seqid_parse_fail (seqid.c:142-285) as a small interpreter: each header format is a list of field
// steps, one nibble each, and ONE scanning loop serves every step -- the whole parser is a few hundred
// instructions instead of one inlined loop per field (which overflowed the instruction cache).
//   STR_*    PARSE_STR  + PARSE_PUSH   (a chunk that must not start at the terminator; 100 characters or
//                                       more fail: the reference overruns its 100-byte field there)
//   INT_*    PARSE_INT  + PARSE_PUSH   (digits only; 32-bit wrap)
//   SKIP     PARSE_CHUNK + PARSE_PUSH  (CASAVA 1.7 "filtered")
//   TAG      the tag chunk: may be empty, at most PAMD_TAG_LEN characters, no push
//   HASHTAG  CASAVA <= 1.6: a tag chunk + push, only when the character before the cursor is '#'
//   POLICY   the PandaTagging test
//   MATE     the final PARSE_INT (no push); RET returns the mate number
enum { OP_STR_INSTR = 1, OP_STR_RUN, OP_STR_FC, OP_INT_LANE, OP_INT_TILE, OP_INT_X, OP_INT_Y, OP_INT_MATE, OP_SKIP, OP_INT_DROP, OP_TAG, OP_POLICY, OP_HASHTAG, OP_MATE, OP_RET };

constexpr u64 id_program(const int *ops, int n) {
	u64 v = 0;
	for (int k = 0; k < n; k++)
		v |= (u64) ops[k] << (4 * k);
	return v;
}

constexpr int OPS_17[] = { OP_STR_INSTR, OP_STR_RUN, OP_STR_FC, OP_INT_LANE, OP_INT_TILE, OP_INT_X, OP_INT_Y, OP_INT_MATE, OP_SKIP, OP_INT_DROP, OP_TAG, OP_POLICY, OP_RET };	// seqid.c:236-284
constexpr int OPS_14[] = { OP_STR_INSTR, OP_INT_LANE, OP_INT_TILE, OP_INT_X, OP_INT_Y, OP_HASHTAG, OP_POLICY, OP_MATE, OP_RET };	// seqid.c:201-235
constexpr int OPS_CONV[] = { OP_STR_INSTR, OP_STR_RUN, OP_STR_FC, OP_INT_LANE, OP_INT_TILE, OP_INT_X, OP_INT_Y, OP_HASHTAG, OP_POLICY, OP_MATE, OP_RET };	// seqid.c:181-200,219-235
constexpr u64 PROG_17 = id_program(OPS_17, 13), PROG_14 = id_program(OPS_14, 9), PROG_CONV = id_program(OPS_CONV, 11);

// returns the mate direction, 0 = not parsed
__device__ int parse_header(Str &h, int policy, IdOut &o) {
	unsigned i = 0;
	o.instr_len = o.run_off = o.run_len = o.fc_off = o.fc_len = o.tag_off = o.tag_len = 0;
	o.lane = o.tile = o.x = o.y = 0;
	o.sra = -1;
	o.fmt = PAMD_IDFMT_UNKNOWN;
	const int c0 = h.at(0), c1 = h.at(1), c2 = h.at(2), c3 = h.at(3);
	if (c0 != 0 && c1 != 0 && c2 != 0 && c3 != 0 && (c0 == 'E' || c0 == 'S') && c1 == 'R' && c2 == 'R') {
		// Short Read Archive (seqid.c:157-172): <E|S>RR<number>.<number> and nothing else is kept
		o.fmt = c0 == 'S' ? PAMD_IDFMT_SRA : PAMD_IDFMT_EBI_SRA;
		o.instr_len = 3;
		i = 3;
		for (int field = 0; field < 2; field++) {	// PARSE_SRA_INT + PARSE_PUSH, twice
			unsigned v = 0;
			for (;;) {
				const int ch = h.at(i);
				if (ch == 0 || ch == '.' || ch == ' ')
					break;
				if (ch < '0' || ch > '9')
					return 0;
				v = v * 10u + (unsigned) (ch - '0');
				i++;
			}
			if (h.at(i) == 0)
				return 0;
			i++;
			if (field == 0)
				o.sra = (int) v;
			else
				o.lane = (int) v;
		}
		return h.at(i) == 0 ? 0 : 1;	// the last PARSE_PUSH
	}
	bool slash = false, before_hash = true;
	unsigned colons = 0;
	for (unsigned k = 0;; k++) {
		const int ch = h.at(k);
		if (ch == 0)
			break;
		slash |= ch == '/';
		before_hash &= ch != '#';
		colons += (before_hash && ch == ':') ? 1u : 0u;
	}
	o.fmt = !slash ? PAMD_IDFMT_CASAVA_1_7 : (colons == 6 ? PAMD_IDFMT_CASAVA_CONVERTED : PAMD_IDFMT_CASAVA_1_4);
	u64 prog = !slash ? PROG_17 : (colons == 6 ? PROG_CONV : PROG_14);
	int mate = 0;
	for (;;) {
		const int op = (int) (prog & 15ull);
		prog >>= 4;
		if (op == OP_RET)
			return mate;
		if (op == OP_POLICY) {
			if (policy != PAMD_TAG_OPTIONAL && policy != (o.tag_len == 0 ? PAMD_TAG_ABSENT : PAMD_TAG_PRESENT))
				return 0;
			continue;
		}
		if (op == OP_HASHTAG && h.at(i - 1) != '#')
			continue;
		const bool is_int = (op >= OP_INT_LANE && op <= OP_INT_MATE) || op == OP_INT_DROP || op == OP_MATE;
		const bool is_tag = op == OP_TAG || op == OP_HASHTAG;
		const bool push = op <= OP_INT_DROP || op == OP_HASHTAG;
		const unsigned limit = op <= OP_STR_FC ? ID_STR_MAX - 1u : (is_tag ? (unsigned) PAMD_TAG_LEN : 0xFFFFFFFFu);
		int ch = h.at(i);
		if (!is_tag && ch == 0)
			return 0;
		const unsigned off = i;
		unsigned v = 0;
		while (!ends_chunk(ch)) {
			if (is_int && (ch < '0' || ch > '9'))
				return 0;
			if (i - off >= limit)
				return 0;
			v = v * 10u + (unsigned) (ch - '0');
			i++;
			ch = h.at(i);
		}
		const unsigned len = i - off;
		if (op == OP_STR_INSTR)
			o.instr_len = len;
		if (op == OP_STR_RUN) {
			o.run_off = off;
			o.run_len = len;
		}
		if (op == OP_STR_FC) {
			o.fc_off = off;
			o.fc_len = len;
		}
		if (is_tag) {
			o.tag_off = off;
			o.tag_len = len;
		}
		o.lane = op == OP_INT_LANE ? (int) v : o.lane;
		o.tile = op == OP_INT_TILE ? (int) v : o.tile;
		o.x = op == OP_INT_X ? (int) v : o.x;
		o.y = op == OP_INT_Y ? (int) v : o.y;
		mate = (op == OP_INT_MATE || op == OP_MATE) ? (int) v : mate;
		if (push) {
			if (ch == 0)
				return 0;
			i++;
		}
	}
}

// are the spans [aoff, aoff+len) of this lane's string and [boff, ...) of its partner's (lane ^ 1) equal?
// Both lanes of a pair call this with mirrored arguments.
__device__ __forceinline__ bool span_equal_pair(Str &h, unsigned off, unsigned len) {
	const unsigned plen = (unsigned) __shfl_xor((int) len, 1);
	bool same = len == plen;
	const unsigned n = same ? len : 0u;	// the same trip count on both lanes of the pair
	for (unsigned k = 0; k < n; k++) {
		const int c = h.at(off + k);
		same &= c == __shfl_xor(c, 1);
	}
	return same;
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

// four bytes of a text starting at byte offset a < bytes (any alignment), from two aligned words; no
// branch: the second word's address is clamped to the text's last word (bytes past the text are always
// past the line too, and the callers mask those)
__device__ __forceinline__ unsigned load4(const TextRef &T, unsigned a) {
	const unsigned a0 = a & ~3u;
	const unsigned a1 = a0 + 4u < T.last_word ? a0 + 4u : T.last_word;
	const unsigned lo = *reinterpret_cast<const unsigned *>(T.text + a0);
	const unsigned hi = *reinterpret_cast<const unsigned *>(T.text + a1);
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

// ---- four characters at a time ----------------------------------------------------------------------
// 0x80 in every byte of x that is zero (exact: no borrow between bytes)
__device__ __forceinline__ unsigned zero_bytes(unsigned x) {
	return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// 0xFF in each of the first n (0..4) bytes
__device__ __forceinline__ unsigned first_bytes(int n) {
	n = n < 0 ? 0 : (n > 4 ? 4 : n);
	return n == 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
}

// The 32-entry tables as bytes, eight entries per 64-bit word: a v_perm_b32 looks four characters up in one
// eight-entry slice, two bit-selects pick the slice each character belongs to.
constexpr u64 nt_slice(int slice, bool complement) {
	u64 v = 0;
	for (int k = 0; k < 8; k++) {
		const int idx = slice * 8 + k;
		unsigned e = (unsigned) ((idx < 16 ? NT_LO : NT_HI) >> (4 * (idx & 15))) & 15u;
		if (complement)
			e = ((e & 1u) << 3) | ((e & 2u) << 1) | ((e & 4u) >> 1) | ((e & 8u) >> 3);
		v |= (u64) e << (8 * k);
	}
	return v;
}

template <bool COMPLEMENT>
__device__ __forceinline__ unsigned nt_of_chars(unsigned w) {
	constexpr u64 S0 = nt_slice(0, COMPLEMENT), S1 = nt_slice(1, COMPLEMENT), S2 = nt_slice(2, COMPLEMENT), S3 = nt_slice(3, COMPLEMENT);
	const unsigned sel = w & 0x07070707u;
	const unsigned p0 = __builtin_amdgcn_perm((unsigned) (S0 >> 32), (unsigned) S0, sel);
	const unsigned p1 = __builtin_amdgcn_perm((unsigned) (S1 >> 32), (unsigned) S1, sel);
	const unsigned p2 = __builtin_amdgcn_perm((unsigned) (S2 >> 32), (unsigned) S2, sel);
	const unsigned p3 = __builtin_amdgcn_perm((unsigned) (S3 >> 32), (unsigned) S3, sel);
	const unsigned b3 = (w >> 3) & 0x01010101u, b4 = (w >> 4) & 0x01010101u;
	const unsigned m3 = (b3 << 8) - b3, m4 = (b4 << 8) - b4;	// 0xFF where the bit is set
	const unsigned lo = (p1 & m3) | (p0 & ~m3), hi = (p3 & m3) | (p2 & ~m3);
	return (hi & m4) | (lo & ~m4);
}

// TOINDEX (fastq.c:44) on four characters: below the offset -> 0, above offset + 46 -> 46 - offset
// (the reference's own clamp quirk), else c - offset; characters are signed.  `valid` marks the bytes
// that belong to the line.  The wave takes the one-subtraction path when every lane's characters are
// inside [offset, offset + 46].
__device__ __forceinline__ unsigned quals_of_chars(unsigned w, int qualmin, unsigned valid) {
	const unsigned ones = 0x01010101u;
	const unsigned x = (w & valid) | ((unsigned) qualmin * ones & ~valid);
	const unsigned ge = ((x | 0x80808080u) - (unsigned) qualmin * ones) & 0x80808080u;		// bit 7 kept <=> byte >= offset
	const unsigned le = (((unsigned) (qualmin + 46) * ones | 0x80808080u) - x) & 0x80808080u;	// bit 7 kept <=> byte <= offset + 46
	const bool plain = qualmin <= 81 && ((x & 0x80808080u) == 0u) && ge == 0x80808080u && le == 0x80808080u;
	if (__ballot(!plain) == 0)
		return (x - (unsigned) qualmin * ones) & valid;
	unsigned out = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int v = (int) (int8_t) ((w >> (8 * k)) & 0xFFu);
		const int q = v < qualmin ? 0 : ((v > qualmin + 46 ? 46 : v) - qualmin);
		out |= ((unsigned) q & 0xFFu) << (8 * k);
	}
	return out & valid;
}

// The characters one lane owns of one read's three lines: 4 of the sequence and 4 of the quality line
// per step of 256 (two steps cover PAMD_MAX_LEN), and the first character of the '+' line.  Loaded in one
// go, one record ahead of their use, so the memory round trips overlap the previous record's arithmetic.
struct ReadWords {
	unsigned seq[2], qual[2], plus;
};

struct ReadSpan {	// wave-uniform: start and length of the three lines, "too long for the line reader" flags
	unsigned s_seq, l_seq, s_plus, l_plus, s_q, l_q;
	bool seq_long, plus_long, q_long;
};

__device__ __forceinline__ ReadWords load_words(const TextRef &T, const ReadSpan &R, int lane) {
	ReadWords W;
	// no branch around a load: a lane past the end of its line re-reads the line's first word (masked later)
	const unsigned i0 = 4u * (unsigned) lane;
	W.seq[0] = load4(T, R.s_seq + (i0 < R.l_seq ? i0 : 0u));
	W.qual[0] = load4(T, R.s_q + (i0 < R.l_q ? i0 : 0u));
	W.plus = load4(T, R.s_plus) & 0xFFu;
	W.seq[1] = 0;
	W.qual[1] = 0;
	if (R.l_seq > 256u || R.l_q > 256u) {	// wave-uniform: reads of up to 256 bases take one step
		const unsigned i1 = 256u + i0;
		W.seq[1] = load4(T, R.s_seq + (i1 < R.l_seq ? i1 : 0u));
		W.qual[1] = load4(T, R.s_q + (i1 < R.l_q ? i1 : 0u));
	}
	return W;
}

// One read: its three lines -> one row.  Returns 0 or the PAMD_FQ_* code read_seq() fails with
// (fastq.c:45-107).  Wave-uniform result.  Lane l owns characters 4l .. 4l+3 of a step of 256.
template <bool COMPLEMENT>
__device__ __forceinline__ int finish_read(const TextRef &T, const ReadSpan &R, const ReadWords &W, int qualmin, uint8_t *row, unsigned stride, int lane, unsigned &len_out, bool &under64) {
	len_out = 0;
	if (R.seq_long)
		return PAMD_FQ_EOF;
	// sequence line: the first PAMD_MAX_LEN characters, up to the first NUL (fastq.c:62).  Characters past
	// the line read as NUL, so the first NUL found is the string length.
	unsigned nts[2] = { 0u, 0u };
	unsigned eff = 512u, bad = 0xFFFFFFFFu;
#pragma unroll
	for (int it = 0; it < 2; it++) {
		if (it == 1 && eff < 512u)
			break;
		const unsigned idx = 256u * it + 4u * (unsigned) lane;
		const unsigned vs = first_bytes((int) R.l_seq - (int) idx);
		const unsigned w = W.seq[it] & vs;
		const unsigned nt4 = nt_of_chars<COMPLEMENT>(w);
		const unsigned zw = zero_bytes(w);
		const unsigned bw = zero_bytes(nt4) & ~zw;	// a character that is no nucleotide
		nts[it] = nt4;
		const u64 zb = __ballot(zw != 0u);
		if (zb != 0) {
			const int l0 = __builtin_ctzll(zb);
			eff = 256u * it + 4u * l0 + ((unsigned) __builtin_ctz((unsigned) __builtin_amdgcn_readlane((int) zw, l0)) >> 3);
		}
		const u64 bb = __ballot(bw != 0u);
		if (bb != 0 && bad == 0xFFFFFFFFu) {
			const int l0 = __builtin_ctzll(bb);
			bad = 256u * it + 4u * l0 + ((unsigned) __builtin_ctz((unsigned) __builtin_amdgcn_readlane((int) bw, l0)) >> 3);
		}
	}
	const unsigned pos = eff < PAMD_MAX_LEN ? eff : PAMD_MAX_LEN;
	if (bad < pos)
		return PAMD_FQ_BADNT;
	// '+' line (fastq.c:68-83)
	if (R.plus_long)
		return PAMD_FQ_EOF;
	{
		const unsigned c = R.l_plus > 0 ? (unsigned) uni((int) W.plus) : 0u;
		if (c != '+')
			return (nt_of_chars<false>(c) & 0xFFu) != 0 ? PAMD_FQ_READLEN : PAMD_FQ_BADSEQ;
	}
	// quality line: every character up to the first NUL counts (fastq.c:89-94)
	if (R.q_long)
		return PAMD_FQ_EOF;
	unsigned quals[2] = { 0u, 0u };
	unsigned qeff = 0xFFFFFFFFu;
	bool u64seen = false;
#pragma unroll
	for (int it = 0; it < 2; it++) {
		if (it == 1 && qeff != 0xFFFFFFFFu)
			break;
		const unsigned idx = 256u * it + 4u * (unsigned) lane;
		const unsigned valid = first_bytes((int) R.l_q - (int) idx);
		const unsigned w = W.qual[it] & valid;
		const unsigned zw = zero_bytes(w);
		// valid characters before this lane's first NUL that are below 64 (signed: 128.. count as well)
		const unsigned before = zw != 0u ? (1u << (__builtin_ctz(zw) & ~7)) - 1u : 0xFFFFFFFFu;
		const unsigned ge64 = ((w & 0x7F7F7F7Fu) + 0x40404040u) & ~w & 0x80808080u;	// bit 7 <=> 64 <= byte < 128
		const bool low = (~ge64 & 0x80808080u & valid & before) != 0u;
		quals[it] = quals_of_chars(w, qualmin, valid);
		const u64 zb = __ballot(zw != 0u);
		const u64 lowb = __ballot(low);
		if (zb != 0) {
			const int l0 = __builtin_ctzll(zb);
			qeff = 256u * it + 4u * l0 + ((unsigned) __builtin_ctz((unsigned) __builtin_amdgcn_readlane((int) zw, l0)) >> 3);
			u64seen |= (lowb & ((2ull << l0) - 1ull)) != 0;
		} else {
			u64seen |= lowb != 0;
		}
	}
	if (qeff == 0xFFFFFFFFu) {
		// 512 characters and no end yet: already too long for any read (NOQUAL below); only the under-64
		// hint still looks at the rest
		qeff = R.l_q;
		for (unsigned base = 512u; base < R.l_q; base += 64u) {
			const unsigned i = base + (unsigned) lane;
			const int c = i < R.l_q ? (int) T.text[R.s_q + i] : 1;
			const u64 zb = __ballot(c == 0);
			const u64 lowb = __ballot(i < R.l_q && (int) (int8_t) c < 64);
			if (zb != 0) {
				const int l0 = __builtin_ctzll(zb);
				qeff = base + (unsigned) l0;
				u64seen |= (lowb & ((1ull << l0) - 1ull)) != 0;
				break;
			}
			u64seen |= lowb != 0;
		}
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
			const unsigned keep = first_bytes((int) pos - (int) idx);
			const unsigned n4 = nts[it] & keep, q4 = quals[it] & keep;
			// interleave: nt0 q0 nt1 q1 | nt2 q2 nt3 q3
			// (nontemporal, as every bulk output of this file: written once, read by a later kernel or copied out; +1..3 % on ingest and formatting)
			typedef unsigned v2u __attribute__((ext_vector_type(2)));
			v2u o;
			o.x = __builtin_amdgcn_perm(q4, n4, 0x05010400u);
			o.y = __builtin_amdgcn_perm(q4, n4, 0x07030602u);
			__builtin_nontemporal_store(o, reinterpret_cast<v2u *>(row + 2u * idx));
		}
	}
	return 0;
}

// a monotone device minimum that stops issuing atomics once it has converged
__device__ __forceinline__ void note_first(u64 *word, u64 rec) {
	if (rec < __atomic_load_n(word, __ATOMIC_RELAXED))
		atomicMin(word, rec);
}

// the four lines of record `rec` of one text: start and length (without the line end) of each, and a bit
// per line that is too long for the reference's line reader
struct Lines {
	unsigned s0, l0, s1, l1, s2, l2, s3, l3, toolong;
};

__device__ __forceinline__ void one_line(const TextRef &T, unsigned &prev, unsigned end, unsigned &start, unsigned &len, unsigned &toolong, unsigned bit) {
	len = end - prev;
	toolong |= len >= PAMD_LINE_MAX ? bit : 0u;	// the '\n' must lie inside the line reader's buffer, linebuf.c:68
	if (len > 0 && T.text[end - 1] == '\r')	// DOS line ends, linebuf.c:81-84
		len--;
	start = prev;
	prev = end + 1u;
}

__device__ __forceinline__ Lines record_lines(const TextRef &T, u64 rec) {
	Lines L;
	unsigned prev = rec == 0 ? 0u : T.line_end[4 * rec - 1] + 1u;
	const unsigned e0 = T.line_end[4 * rec], e1 = T.line_end[4 * rec + 1], e2 = T.line_end[4 * rec + 2], e3 = T.line_end[4 * rec + 3];
	L.toolong = 0;
	one_line(T, prev, e0, L.s0, L.l0, L.toolong, 1u);
	one_line(T, prev, e1, L.s1, L.l1, L.toolong, 2u);
	one_line(T, prev, e2, L.s2, L.l2, L.toolong, 4u);
	one_line(T, prev, e3, L.s3, L.l3, L.toolong, 8u);
	return L;
}

__device__ __forceinline__ ReadSpan span_of(const Lines &L, int j) {
	ReadSpan R;
	R.s_seq = (unsigned) __builtin_amdgcn_readlane((int) L.s1, j);
	R.l_seq = (unsigned) __builtin_amdgcn_readlane((int) L.l1, j);
	R.s_plus = (unsigned) __builtin_amdgcn_readlane((int) L.s2, j);
	R.l_plus = (unsigned) __builtin_amdgcn_readlane((int) L.l2, j);
	R.s_q = (unsigned) __builtin_amdgcn_readlane((int) L.s3, j);
	R.l_q = (unsigned) __builtin_amdgcn_readlane((int) L.l3, j);
	const unsigned t = (unsigned) __builtin_amdgcn_readlane((int) L.toolong, j);
	R.seq_long = (t & 2u) != 0;
	R.plus_long = (t & 4u) != 0;
	R.q_long = (t & 8u) != 0;
	return R;
}

// ---- headers (fastq.c:123-144): two lanes per record, the even one on the forward header, the odd one on
// the reverse header; the pair compares its results through lane exchanges (panda_seqid_equal, seqid.c:96-101)
__global__ __launch_bounds__(256) void fq_ids_kernel(const ParseParams P) {
	const u64 tid = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	const u64 rec = tid >> 1;
	const bool rev = (tid & 1ull) != 0;
	const bool have = rec < P.n_records;	// both lanes of a pair agree
	const TextRef &T = rev ? P.r : P.f;
	unsigned start = 0, len = 0;
	bool toolong = false;
	if (have) {
		start = rec == 0 ? 0u : T.line_end[4 * rec - 1] + 1u;
		const unsigned end = T.line_end[4 * rec];
		len = end - start;
		toolong = len >= PAMD_LINE_MAX;	// the '\n' must lie inside the line reader's buffer, linebuf.c:68
		if (len > 0 && T.text[end - 1] == '\r')	// DOS line ends, linebuf.c:81-84
			len--;
	}
	// The first 96 bytes of the header go to LDS with six independent 16-byte loads, so the character
	// walk below waits on LDS, not on one dependent global load per word.  Word w of lane l sits at
	// [w][l]: lanes walking in step hit 64 different banks.
	__shared__ unsigned stage[4][STAGE_WORDS * 64];
	unsigned *col = &stage[threadIdx.x >> 6][threadIdx.x & 63];
	const unsigned base16 = (start + 1u) & ~15u;
	{
		uint4 v[STAGE_WORDS / 4];
#pragma unroll
		for (unsigned k = 0; k < STAGE_WORDS / 4; k++) {
			const unsigned a = base16 + 16u * k;
			// a 16-byte group is read whole when any of it lies inside the text (allocations are at least that granular)
			v[k] = (have && (u64) a < T.bytes) ? *reinterpret_cast<const uint4 *>(T.text + a) : make_uint4(0u, 0u, 0u, 0u);
		}
#pragma unroll
		for (unsigned k = 0; k < STAGE_WORDS / 4; k++) {
			col[64u * (4u * k + 0u)] = v[k].x;
			col[64u * (4u * k + 1u)] = v[k].y;
			col[64u * (4u * k + 2u)] = v[k].z;
			col[64u * (4u * k + 3u)] = v[k].w;
		}
	}
	Str h = { T.text, start + 1u, len > 0 ? len - 1u : 0u, 0xFFFFFFFFu, 0u, col, base16 };
	IdOut id;
	id.instr_len = id.run_off = id.run_len = id.fc_off = id.fc_len = id.tag_off = id.tag_len = 0;
	id.lane = id.tile = id.x = id.y = 0;
	id.sra = -1;
	id.fmt = 0;
	int dir = 0;
	if (have && !toolong && len > 0)	// an empty header line fails too (the reference reads past its terminator)
		dir = parse_header(h, P.policy, id);
	// ---- the pair compares ----
	const bool p_toolong = __shfl_xor((int) toolong, 1) != 0;
	const int p_dir = __shfl_xor(dir, 1);
	bool equal = id.lane == __shfl_xor(id.lane, 1) && id.tile == __shfl_xor(id.tile, 1) && id.x == __shfl_xor(id.x, 1) && id.y == __shfl_xor(id.y, 1);
	const int p_sra = __shfl_xor(id.sra, 1);
	// "<E|S>RR<n>" can never equal a parsed instrument chunk (such a chunk would itself be read as SRA)
	equal &= (id.sra >= 0) == (p_sra >= 0);
	const int first = h.at(0);
	const bool sra_same = id.sra == p_sra && first == __shfl_xor(first, 1);
	const bool instr_same = span_equal_pair(h, 0, id.sra >= 0 ? 0u : id.instr_len);
	equal &= id.sra >= 0 ? sra_same : instr_same;
	equal &= span_equal_pair(h, id.run_off, id.run_len);
	equal &= span_equal_pair(h, id.fc_off, id.fc_len);
	equal &= span_equal_pair(h, id.tag_off, id.tag_len);
	if (!have || rev)
		return;
	int err = 0;
	if (toolong)
		err = PAMD_FQ_LONG_HEADER;
	else if (dir == 0)
		err = PAMD_FQ_BADID_FWD;
	else if (p_toolong)
		err = PAMD_FQ_LONG_HEADER;
	else if (p_dir == 0)
		err = PAMD_FQ_BADID_REV;
	else if (!equal || (id.fmt != PAMD_IDFMT_SRA && id.fmt != PAMD_IDFMT_EBI_SRA && p_dir == dir))
		err = PAMD_FQ_NOTPAIRED;
	const unsigned flag = err != 0 ? 16u + (unsigned) err : 0u;
	P.flags[rec] = (uint8_t) flag;
	const u64 hdr = (u64) start + 1ull;
	uint4 *dst = reinterpret_cast<uint4 *>(&P.ids[rec]);
	dst[0] = make_uint4((unsigned) hdr, (unsigned) (hdr >> 32), (unsigned) id.lane, (unsigned) id.tile);
	dst[1] = make_uint4((unsigned) id.x, (unsigned) id.y, (unsigned) id.sra, id.instr_len | (id.run_off << 16));
	dst[2] = make_uint4(id.run_len | (id.fc_off << 16), id.fc_len | (id.tag_off << 16), id.tag_len | ((unsigned) id.fmt << 16) | (((unsigned) dir & 0xFFu) << 24), flag);
	if (err != 0)
		note_first(P.first_bad, rec);
	else if (id.fmt == PAMD_IDFMT_CASAVA_1_7)	// CASAVA 1.7+ is always PHRED+33: fastq.c:141-144
		note_first(P.first_u64, rec);
}

// ---- reads (fastq.c:145-154): 32 records per wave, converted one record at a time by the whole wave;
// lanes 0..31 hold the forward line spans of the 32 records, lanes 32..63 the reverse ones, and record
// j+1's characters are loaded while record j is converted
constexpr int READS_PER_WAVE = 32;

__global__ __launch_bounds__(256) void fq_reads_kernel(const ParseParams P) {
	const int lane = threadIdx.x & 63;
	const u64 wave = ((u64) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const u64 rec0 = wave * READS_PER_WAVE;
	if (rec0 >= P.n_records)
		return;
	const int nrec = (int) (P.n_records - rec0 < (u64) READS_PER_WAVE ? P.n_records - rec0 : (u64) READS_PER_WAVE);
	const int slot = lane & (READS_PER_WAVE - 1);
	const u64 rec = rec0 + (u64) slot;
	Lines L = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	int err = 0;
	if (slot < nrec) {
		L = record_lines(lane < READS_PER_WAVE ? P.f : P.r, rec);
		err = P.flags[rec];	// the headers may already have stopped the reader here
	}
	bool lane_u64 = false;
	int in_rev = 0;
	unsigned my_flen = 0, my_rlen = 0;
	ReadSpan nf = span_of(L, 0), nr = span_of(L, READS_PER_WAVE);
	ReadWords nwf = load_words(P.f, nf, lane), nwr = load_words(P.r, nr, lane);
	for (int j = 0; j < nrec; j++) {
		const ReadSpan cf = nf, cr = nr;
		const ReadWords wf = nwf, wr = nwr;
		const int jn = j + 1 < nrec ? j + 1 : j;
		nf = span_of(L, jn);
		nr = span_of(L, READS_PER_WAVE + jn);
		nwf = load_words(P.f, nf, lane);
		nwr = load_words(P.r, nr, lane);
		if (__builtin_amdgcn_readlane(err, j) != 0)
			continue;
		unsigned flen_j = 0, rlen_j = 0;
		bool low64 = false;
		const u64 recj = rec0 + (u64) j;
		int code = finish_read<false>(P.f, cf, wf, P.qualmin, P.frows + recj * (u64) P.fstride * 2ull, P.fstride, lane, flen_j, low64);
		int rev_fail = 0;
		if (code == 0) {
			code = finish_read<true>(P.r, cr, wr, P.qualmin, P.rrows + recj * (u64) P.rstride * 2ull, P.rstride, lane, rlen_j, low64);
			rev_fail = code != 0;
		}
		if (lane == j) {
			err = code;
			in_rev = rev_fail;
			my_flen = code == 0 ? flen_j : 0u;
			my_rlen = code == 0 ? rlen_j : 0u;
			lane_u64 = low64;
		}
	}
	// ---- per-record outputs (lanes 0..31) ----
	const bool mine = lane < nrec;
	if (mine) {
		const bool from_header = my_flen == 0 && my_rlen == 0 && err != 0 && P.flags[rec] != 0;
		P.flen[rec] = (uint16_t) my_flen;
		P.rlen[rec] = (uint16_t) my_rlen;
		if (!from_header) {
			const unsigned flag = err != 0 ? (16u + (unsigned) err) | (in_rev ? 128u : 0u) : (my_flen == 0 ? 1u : 0u);
			if (flag != 0) {
				P.flags[rec] = (uint8_t) flag;
				reinterpret_cast<uint8_t *>(&P.ids[rec])[44] = (uint8_t) flag;
			}
		} else {
			err = 0;	// already reported by the header kernel
		}
	}
	const u64 badb = __ballot(mine && err != 0);
	if (badb != 0 && lane == 0)
		note_first(P.first_bad, rec0 + (u64) __builtin_ctzll(badb));
	const u64 ub = __ballot(mine && lane_u64);
	if (ub != 0 && lane == 0)
		note_first(P.first_u64, rec0 + (u64) __builtin_ctzll(ub));
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
		const unsigned flag = P.flags[i];
		dropped += flag == 1 ? 1u : 0u;
		const unsigned a = flag == 0 ? P.flen[i] : 0u, b = flag == 0 ? P.rlen[i] : 0u;	// (a dropped record's other read does not count)
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

// The same function without the walk.  The walk's answer depends only on where p lies among the 47 table values: on one of
// them (exact[k]), or strictly between two (mid[k]: between entries k and k+1; mid[46]: above the last) -- both small tables
// are made by the walk itself when the kernel starts.  Where p lies follows from arithmetic, qual_score[k] being
// log(1 - 10^(-k/10)): k = -10 log10(1 - e^p) in single precision is off by far less than one, so p lies in one of two
// neighbouring intervals and three table reads (one round trip instead of six dependent ones) say which.  If they do not --
// it has not happened -- the walk decides.
__device__ __forceinline__ int result_phred_fast(const double *score, const uint8_t *exact, const uint8_t *mid, double p) {
	if (p <= score[0])
		return 1;
	const float x = (float) p;
	const float y = x > -0.01f ? -x * (1.0f + x * (0.5f + x * (1.0f / 6.0f))) : 1.0f - __expf(x);
	const float kf = -3.01029995664f * __log2f(y);
	int r0 = kf > 46.0f ? 46 : (int) kf;	// (p = 0: y = 0, kf = +inf)
	r0 = r0 < 1 ? 1 : r0;
	const double s_lo = score[r0 - 1], s_mid = score[r0], s_hi = score[r0 + 1];	// score[47] = +inf
	if (!(p > s_lo && p <= s_hi))
		return result_phred_io(score, p);
	const bool low = p <= s_mid;
	const int r = low ? r0 : r0 + 1;	// score[r - 1] < p <= score[r]
	return p == (low ? s_mid : s_hi) ? exact[r] : mid[r - 1];
}

// ---- the writer ----
// Lanes of one wavefront exchange data through LDS: the LDS unit executes one wave's operations in issue order, so ordering
// the compiler is all that is needed.
// (not a fence: a release fence would also wait for every global store issued so far -- the pairs written before this one)
__device__ __forceinline__ void io_wave_sync() {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

// sixteen bytes to any address: text has no alignment, and a store of a packed type is one global_store_dwordx4 (the ROCm
// runtime runs compute queues with unaligned access enabled, which is what the compiler assumes for this target)
struct __attribute__((packed, aligned(1))) Text16 {
	unsigned x, y, z, w;
};
__device__ __forceinline__ void store16(uint8_t *dst, unsigned x, unsigned y, unsigned z, unsigned w) {
	typedef unsigned v4u_unaligned __attribute__((ext_vector_type(4), aligned(1)));
	v4u_unaligned t;
	t.x = x;
	t.y = y;
	t.z = z;
	t.w = w;
	__builtin_nontemporal_store(t, reinterpret_cast<v4u_unaligned *>(dst));
}

// panda_nt_to_ascii, nt.c:167-173 ("NACMGRSVTWYHKDBN"; anything outside 0..15 prints N): one nt / four nts of a word
__device__ __forceinline__ unsigned ascii1(unsigned v) {
	return v > 15u ? (unsigned) 'N' : (unsigned) ((v < 8u ? ASCII_LO >> (8u * v) : ASCII_HI >> (8u * (v - 8u))) & 0xFFu);
}
__device__ __forceinline__ unsigned ascii4(unsigned w) {
	const unsigned sel = w & 0x07070707u;
	const unsigned lo = __builtin_amdgcn_perm((unsigned) (ASCII_LO >> 32), (unsigned) ASCII_LO, sel);
	const unsigned hi = __builtin_amdgcn_perm((unsigned) (ASCII_HI >> 32), (unsigned) ASCII_HI, sel);
	const unsigned b3 = (w >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;	// 0xFF where nt >= 8
	const unsigned r = (hi & m3) | (lo & ~m3);
	const unsigned big = w & 0xF0F0F0F0u;
	const unsigned nz = (((big & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | big) & 0x80808080u, mb = (nz >> 7) * 255u;	// 0xFF where nt > 15
	return (r & ~mb) | (0x4E4E4E4Eu & mb);
}

// a pair's header line, byte by byte into `h` (LDS or the output itself); returns its length
template <typename Sink>
__device__ __forceinline__ unsigned compose_header(Sink *h, const FormatParams &P, const IdView &v, u64 i, u64 my_off, unsigned qlo, unsigned qhi) {
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
	q.u[0] = qlo;
	q.u[1] = qhi;
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
	return n;
}

// Header lines are composed one pair per lane -- into LDS, where a lane's single bytes cost nothing (straight into the output
// they were sixty scattered one-byte writes per pair, each its own memory transaction) -- and leave with the pair's bases, one
// pair per wave: the header as contiguous bytes, bases and qualities sixteen per lane in one store each.
constexpr unsigned HDR_SLOT = 92;	// bytes between two lanes' slots (23 words, odd: the lanes' writes spread over the banks)
constexpr unsigned HDR_MAX = 88;	// a longer header line goes straight to the output
constexpr unsigned QUAL_BLOCK = 512;	// qualities staged per round (26 KB of LDS per workgroup in all: six workgroups per CU)

__global__ __launch_bounds__(256) void fmt_write_kernel(const FormatParams P) {
	// one workgroup = 4 waves = 256 pairs (a quarter of a fmt_len_kernel group); each wave owns 64 pairs
	__shared__ double score[48];
	__shared__ uint8_t ph_exact[48], ph_mid[48];	// result_phred_fast's two tables
	__shared__ __attribute__((aligned(16))) uint8_t hdrs[4][64 * HDR_SLOT];
	__shared__ __attribute__((aligned(16))) uint8_t quals[4][2][QUAL_BLOCK];	// two pairs' qualities: one being written out, one being made
	__shared__ __attribute__((aligned(16))) uint8_t tails[4][16];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x < 48)
		score[threadIdx.x] = threadIdx.x < 47 ? P.score[threadIdx.x] : INFINITY;
	__syncthreads();
	if (P.fastq && threadIdx.x < 47) {	// what the walk answers on every table value and between any two
		const int t = threadIdx.x;
		ph_exact[t] = (uint8_t) result_phred_io(score, score[t]);
		ph_mid[t] = (uint8_t) result_phred_io(score, t < 46 ? 0.5 * (score[t] + score[t + 1]) : 0.5 * score[46]);
	}
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
	// ---- header lines, one pair per lane ----
	unsigned hlen = 0;
	unsigned seq_len = 0;
	u64 seq_off = 0;
	bool direct = false;
	if (len != 0 && my_off + len > P.out_capacity)
		len = 0;	// does not fit: reported by the host from the total
	if (len != 0) {
		const IdView v = load_id(P.ids, i);
		const uint4 *r = reinterpret_cast<const uint4 *>(&P.results[i]);
		const uint4 ra = r[0], rb = r[1];
		seq_off = (u64) rb.x | ((u64) rb.y << 32);
		seq_len = rb.z & 0xFFFFu;
		direct = 1u + id_text_len(v) + 10u > HDR_MAX;
		if (direct)
			hlen = compose_header(P.out + my_off, P, v, i, my_off, ra.x, ra.y);
		else
			hlen = compose_header(&hdrs[wv][lane * HDR_SLOT], P, v, i, my_off, ra.x, ra.y);
	}
	io_wave_sync();
	// ---- header, bases (and qualities), one pair per wave ----
	// Software-pipelined over the wave's pairs: the NEXT pair's row loads are issued before this pair's stores, and used
	// after them.  vmcnt counts loads and stores in one in-order queue: a pair that loaded its log-p row after the previous
	// pair's stores would wait for those stores to be acknowledged before it could look at a single value (measured: the
	// FASTQ form ran 35 % slower with wide stores than with the one-byte stores it had, until the loads moved ahead).
	constexpr int NP = (int) (QUAL_BLOCK / 64u);
	struct Row {
		unsigned hj, sl;
		bool dj, chunky;
		u64 off, so;
	};
	auto row_of = [&](int j) {
		Row r;
		r.hj = (unsigned) __builtin_amdgcn_readlane((int) hlen, j);
		r.sl = (unsigned) __builtin_amdgcn_readlane((int) seq_len, j);
		r.dj = __builtin_amdgcn_readlane((int) direct, j) != 0;
		r.off = (u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) my_off, j) | ((u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (my_off >> 32), j) << 32);
		r.so = (u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) seq_off, j) | ((u64) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (seq_off >> 32), j) << 32);
		// sixteen bases per lane from one aligned load: rows of the assembler's arenas are 16-byte aligned and padded
		r.chunky = ((reinterpret_cast<uintptr_t>(P.seq_nt + r.so)) & 15u) == 0 && r.sl <= 1024u;
		return r;
	};
	auto issue_loads = [&](const Row &r, uint4 &ntv, double (&pv)[NP]) {
		ntv = make_uint4(0u, 0u, 0u, 0u);
		if (r.chunky && 16u * (unsigned) lane < r.sl)
			ntv = reinterpret_cast<const uint4 *>(P.seq_nt + r.so)[lane];
		if (P.fastq) {
			const double *p = P.seq_p + r.so;
#pragma unroll
			for (int q = 0; q < NP; q++) {
				const unsigned k = 64u * (unsigned) q + (unsigned) lane;
				pv[q] = k < r.sl ? p[k] : 0.0;
			}
		}
	};
	auto stage_quals = [&](const Row &r, const double (&pv)[NP], uint8_t *qb) {
		if (P.fastq) {
#pragma unroll
			for (int q = 0; q < NP; q++) {
				const unsigned k = 64u * (unsigned) q + (unsigned) lane;
				if (k < r.sl)
					qb[k] = (uint8_t) (33 + result_phred_fast(score, ph_exact, ph_mid, pv[q]));
			}
		}
	};
	u64 todo = __ballot(len != 0);
	if (todo == 0)
		return;
	int j = __builtin_ctzll(todo);
	todo &= todo - 1;
	Row cur = row_of(j);
	uint4 ntc;
	int buf = 0;
	{
		double pv[NP];
		issue_loads(cur, ntc, pv);
		stage_quals(cur, pv, quals[wv][0]);
	}
	io_wave_sync();
	for (;;) {
		const bool more = todo != 0;
		Row nxt = cur;
		int jn = j;
		uint4 ntn = ntc;
		double pn[NP];
		if (more) {
			jn = __builtin_ctzll(todo);
			todo &= todo - 1;
			nxt = row_of(jn);
			issue_loads(nxt, ntn, pn);
		}
		// ---- this pair's text ----
		const unsigned sl = cur.sl;
		uint8_t *dst = P.out + cur.off;
		if (!cur.dj) {
			const uint8_t *hs = &hdrs[wv][j * HDR_SLOT];
			for (unsigned k = lane; k < cur.hj; k += 64)
				dst[k] = hs[k];
		}
		dst += cur.hj;
		if (cur.chunky) {
			const unsigned full = sl >> 4, t0 = full << 4;
			const unsigned a0 = ascii4(ntc.x), a1 = ascii4(ntc.y), a2 = ascii4(ntc.z), a3 = ascii4(ntc.w);
			if ((unsigned) lane < full)
				store16(dst + 16u * (unsigned) lane, a0, a1, a2, a3);
			if (t0 < sl) {	// the last, partial chunk leaves byte by byte, through LDS
				if ((unsigned) lane == full)
					*reinterpret_cast<uint4 *>(tails[wv]) = make_uint4(a0, a1, a2, a3);
				io_wave_sync();
				if (t0 + (unsigned) lane < sl)
					dst[t0 + lane] = tails[wv][lane];
			}
		} else {
			const uint8_t *nt = P.seq_nt + cur.so;
			for (unsigned k = lane; k < sl; k += 64)
				dst[k] = (uint8_t) ascii1(nt[k]);
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
			uint8_t *dq = dst + 2;
			const uint8_t *qb = quals[wv][buf];
			const unsigned nb = sl < QUAL_BLOCK ? sl : QUAL_BLOCK, full = nb >> 4, t0 = full << 4;
			if ((unsigned) lane < full) {
				const uint4 v = reinterpret_cast<const uint4 *>(qb)[lane];
				store16(dq + 16u * (unsigned) lane, v.x, v.y, v.z, v.w);
			}
			if (t0 + (unsigned) lane < nb)
				dq[t0 + lane] = qb[t0 + lane];
			const double *p = P.seq_p + cur.so;	// (a row longer than the staged block: the rest straight from memory)
			for (unsigned k = QUAL_BLOCK + lane; k < sl; k += 64)
				dq[k] = (uint8_t) (33 + result_phred_fast(score, ph_exact, ph_mid, p[k]));
		}
		if (!more)
			break;
		stage_quals(nxt, pn, quals[wv][buf ^ 1]);
		io_wave_sync();
		cur = nxt;
		j = jn;
		ntc = ntn;
		buf ^= 1;
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
	DevBuf counts_f, lines_f, lines_r, flags, small;	// counts_f: chained-scan tile states + ticket
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
	F.last_word = (unsigned) ((fwd_bytes - 1) & ~(size_t) 3);
	R.text = reinterpret_cast<const uint8_t *>(rev_text);
	R.bytes = rev_bytes;
	R.n_tiles = (unsigned) ((rev_bytes + TILE_BYTES - 1) / TILE_BYTES);
	R.last_word = (unsigned) ((rev_bytes - 1) & ~(size_t) 3);
	int rc;
	// the chained scan needs one state word per tile; the line index holds the lines of max_pairs records
	const size_t cap64 = 4 * (size_t) max_pairs + 4;
	const unsigned line_cap = cap64 > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned) cap64;
	const unsigned groups_f = (F.n_tiles + INDEX_WAVES - 1) / INDEX_WAVES, groups_r = (R.n_tiles + INDEX_WAVES - 1) / INDEX_WAVES;
	const size_t state_bytes = ((size_t) groups_f + groups_r) * 8;
	if ((rc = p->counts_f.ensure(state_bytes + 64)) != PAMD_OK || (rc = p->small.ensure(256)) != PAMD_OK
	    || (rc = p->lines_f.ensure((size_t) line_cap * 4 + 16)) != PAMD_OK || (rc = p->lines_r.ensure((size_t) line_cap * 4 + 16)) != PAMD_OK)
		return rc;
	F.tile_state = static_cast<u64 *>(p->counts_f.p);
	R.tile_state = F.tile_state + groups_f;
	unsigned *d_ticket = reinterpret_cast<unsigned *>(F.tile_state + groups_f + groups_r);
	F.line_end = static_cast<unsigned *>(p->lines_f.p);
	R.line_end = static_cast<unsigned *>(p->lines_r.p);
	F.line_cap = R.line_cap = line_cap;
	unsigned *d_totals = static_cast<unsigned *>(p->small.p);	// [0..1] line totals; FinalOut at +64 bytes
	FinalOut *d_final = reinterpret_cast<FinalOut *>(static_cast<uint8_t *>(p->small.p) + 64);

	HIP_TRY(hipEventRecord(p->ev[0], st));
	HIP_TRY(hipMemsetAsync(p->counts_f.p, 0, state_bytes + 64, st));
	HIP_TRY(hipMemsetAsync(d_totals, 0, 16, st));
	hipLaunchKernelGGL(fq_index_kernel, dim3(groups_f + groups_r), dim3(INDEX_WAVES * 64), 0, st, F, R, d_ticket, d_totals);
	HIP_TRY(hipEventRecord(p->ev[1], st));
	unsigned totals[3] = { 0, 0, 0 };
	HIP_TRY(hipMemcpyAsync(totals, d_totals, sizeof totals, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (totals[2] != 0) {
		pamd_set_error("FASTQ line index: the chained scan gave up waiting on %u tiles", totals[2]);
		return PAMD_ENODEVICE;
	}

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
		hipLaunchKernelGGL(fq_ids_kernel, dim3((unsigned) ((2 * n_records + 255) / 256)), dim3(256), 0, st, P);
		const u64 waves = (n_records + READS_PER_WAVE - 1) / READS_PER_WAVE;
		hipLaunchKernelGGL(fq_reads_kernel, dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st, P);
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
