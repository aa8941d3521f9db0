// pamd_kernels.hip -- gfx950 (CDNA4) kernels of the overlap-assembly hot path.
//
// One 64-lane wavefront assembles one read pair end to end (everything assemble_seq(),
// assembler.c:252-348, does for it), inside a persistent workgroup that stages the probability
// tables in LDS once.  No MFMA: this is byte compares, table lookups and ordered f64 sums.
//
//   phase            reference                                   lanes work on
//   load+pack        fastq.c output contract (panda_qual)        8 bases each (one 16-B load per read)
//   primer offsets   offset.c:47-112 via assembler.c:263-284     one primer start position each
//   seed             misc.h:41-45 + assembler.c:84-118           one k-mer position each, LDS hash table
//   scan + argmax    assembler.c:120-143 + algo_*.c overlap_probability   one overlap base each
//   merge            assembler.c:146-247 + algo_*.c match_probability     one output base each
//   quality sums     assembler.c:171,226,236,244                 3 lanes, one ordered chain each
//   filters+counters assembler.c:334-346, plugin_min_*.c         wave-uniform
//
// Bit-exactness rules followed throughout: f64 everywhere the reference uses double; sums are
// accumulated in the reference's index order (never tree-reduced); no FMA contraction (this file
// must be compiled with -ffp-contract=off); table values are the 6-digit decimals of table.c.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pamd_internal.h"
#include "pamd_device.h"

namespace {

// k-mer table slots per wave: a power of two >= 512 (> 442 = most forward k-mers a 450-base read
// can emit), as many as fit in the log-p staging area it shares (PamdKernelParams::hash_slots)
constexpr unsigned HASH_EMPTY = 0xFFFFFFFFu;
constexpr unsigned NT_N = 15;

using namespace pamd_dev;

// Optional per-phase cycle accounting (build with -DPAMD_PHASE_TIMING; see tools/phase_timing.py).
// Costs a few % itself, so it is never compiled into the shipped library.
#ifdef PAMD_PHASE_TIMING
#define PH_DECL u64 ph_t0 = __builtin_amdgcn_s_memtime(), ph_t1
#define PH_MARK_END(k) PH_MARK(k)
#define PH_MARK(k)                                   \
	do {                                         \
		ph_t1 = __builtin_amdgcn_s_memtime(); \
		ph_acc[k] += ph_t1 - ph_t0;          \
		ph_t0 = ph_t1;                       \
	} while (0)
#elif defined(PAMD_STOP_AFTER)
// Ablation builds (tools/phase_ablation.py): every pair is abandoned (as NOALGN) after phase
// PAMD_STOP_AFTER, so the kernel time of build k minus build k-1 is what phase k costs.
#define PH_DECL
#define PH_MARK(k)                           \
	if ((k) == PAMD_STOP_AFTER) {        \
		status = PAMD_ST_NOALGN;     \
		break;                       \
	}
#define PH_MARK_END(k)
#else
#define PH_DECL
#define PH_MARK(k)
#define PH_MARK_END(k)
#endif

__device__ __forceinline__ unsigned kmer_hash(unsigned kmer, unsigned shift) {
	return (kmer * 0x9E3779B1u) >> shift;	// top log2(slots) bits
}

__device__ __forceinline__ int nt_of(unsigned v) {
	return (int) (v & 0xFFu);
}

__device__ __forceinline__ int qual_of(unsigned v) {
	return (int) (int8_t) (v >> 8);
}

// Per-wave views into LDS.
struct WaveLds {
	uint16_t *F;	// forward read, panda_qual as u16 (nt | qual << 8)
	uint16_t *R;	// reverse read
	uint32_t *fc;	// forward 2-bit codes, 16 per word, first base in the TOP bits
	uint32_t *rc;	// reverse 2-bit codes, 16 per word, first base in the LOW bits
	uint32_t *fn;	// forward "special base" map, 2 bits per base like the codes: 3 = N, 1 = any other nt that is not
			// exactly one of A C G T, 0 = plain base
	uint32_t *rn;	// reverse map, the two bits of every entry swapped like the reverse codes (3 = N, 2 = other)
	uint32_t *bits;	// candidate overlaps (BITS_INIT, assembler.c:41,84)
	uint8_t *nts;	// merged nt staging
	double *ps;	// merged log-p staging; the k-mer table lives here while seeding
	uint16_t *pidx;	// `-a` only: which table entry each merged log-p came from (index into the table image)
	uint32_t *hash;
};

// ---- load one read, stage it, derive its packed k-mer inputs --------------------------------
// Returns the B-cliff length: read length minus the trailing run of raw quality 2 (assembler.c:176-177).
// Branch-free so the compiler can count it in s_waitcnt vmcnt(N): every lane loads BPL bases (2*BPL
// bytes) of the row, lanes past the row end re-read its last bytes (rows are whole multiples of 16 B).
// BPL = 4 when a row fits 64 lanes x 4 bases (reads <= 256): all 64 lanes then share the packing work.
template <int BPL>
__device__ __forceinline__ uint4 load_read(const uint8_t *row, unsigned row_bytes, int lane) {
	unsigned off = (unsigned) lane * (2u * BPL);
	off = off < row_bytes ? off : row_bytes - 2u * BPL;
	if (BPL == 8)
		return *reinterpret_cast<const uint4 *>(row + off);
	const uint2 v = *reinterpret_cast<const uint2 *>(row + off);
	return make_uint4(v.x, v.y, 0u, 0u);
}

// word w (0..3) of the 16-entry byte table nt -> (2-bit k-mer code | special-base map entry << 2) of one read
// direction (see stage_read); entries 4w .. 4w+3, lowest nt in the low byte
template <bool FORWARD>
__device__ __forceinline__ constexpr unsigned lut_word(int w) {
	unsigned v = 0;
	for (int k = 0; k < 4; k++) {
		const int nt = 4 * w + k;
		const unsigned code = ((FORWARD ? 0x00030210u : 0x00030120u) >> (2 * nt)) & 3u;
		const unsigned special = ((FORWARD ? 0xD5545441u : 0xEAA8A882u) >> (2 * nt)) & 3u;
		v |= (code | (special << 2)) << (8 * k);
	}
	return v;
}

template <bool FORWARD, int BPL>
__device__ __forceinline__ int stage_read(const uint4 v, int len, int lane, uint16_t *dst, uint32_t *codes, uint32_t *nmap) {
	if (lane * BPL < len) {
		if (BPL == 8)
			reinterpret_cast<uint4 *>(dst)[lane] = v;
		else
			reinterpret_cast<uint2 *>(dst)[lane] = make_uint2(v.x, v.y);
	}
	// Branch-free on purpose (straight-line bit arithmetic; ?: chains here compile to exec-mask
	// branches per base).  nt is 0..15 in the device domain.
	//   2-bit code (misc.h:41: T=3 G=2 C=1, anything else 0) = two bits of a constant indexed by nt
	//              (the reverse read stores C=2 G=1, see below)
	//   N          = (nt + 1) >> 4
	//   qual == 2  = bit 8 of ((qual ^ 2) - 1) taken as an unsigned byte
	unsigned codes_k = 0, is2 = 0, nbits = 0;
	// special-base map entry by nt: 3 for N (15), 0 for A C G T (1 2 4 8), 1 for everything else; the
	// reverse read stores 2 instead of 1 (entries are bit-swapped like its codes)
	constexpr unsigned SPECIAL_F = 0xD5545441u, SPECIAL_R = 0xEAA8A882u;	// 16 two-bit entries, nt 0 in the low bits
	constexpr unsigned CODE_OF = FORWARD ? 0x00030210u : 0x00030120u;
	if (BPL == 4) {
		// four bases at once: the nt and quality bytes are gathered into one word each, the two 2-bit
		// maps come out of ONE 16-entry byte table (code | special << 2) read with v_perm_b32, and
		// "quality == 2" is a zero-byte test
		const unsigned n4 = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u) & 0x0F0F0F0Fu;
		const unsigned q4 = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
		const unsigned sel = n4 & 0x07070707u;
		const unsigned b3 = (n4 >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;	// 0xFF where nt >= 8
		const unsigned lo = __builtin_amdgcn_perm(lut_word<FORWARD>(1), lut_word<FORWARD>(0), sel);
		const unsigned hi = __builtin_amdgcn_perm(lut_word<FORWARD>(3), lut_word<FORWARD>(2), sel);
		const unsigned r4 = (hi & m3) | (lo & ~m3);
		const unsigned c4 = r4 & 0x03030303u, s4 = (r4 >> 2) & 0x03030303u;
		const unsigned cy = c4 | (c4 >> 6), sy = s4 | (s4 >> 6);
		codes_k = (cy | (cy >> 12)) & 0xFFu;	// base k at bits 2k..2k+1
		nbits = (sy | (sy >> 12)) & 0xFFu;
		const unsigned x = q4 ^ 0x02020202u;
		const unsigned t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);	// 0x80 where quality == 2
		is2 = (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
	} else {
		const unsigned w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int k = 0; k < BPL; k++) {
			const unsigned e = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
			const unsigned nt2 = (e & 15u) * 2u;
			const unsigned c = (CODE_OF >> nt2) & 3u;
			codes_k |= c << (2 * k);
			nbits |= (((FORWARD ? SPECIAL_F : SPECIAL_R) >> nt2) & 3u) << (2 * k);
			is2 |= (((((e >> 8) ^ 2u) & 0xFFu) - 1u) >> 8 & 1u) << k;
		}
	}
	// bases at or beyond the read length contribute nothing
	int cnt = len - lane * BPL;
	cnt = cnt < 0 ? 0 : (cnt > BPL ? BPL : cnt);
	const unsigned inmask = (1u << cnt) - 1u;	// bit k set <=> base k of this lane exists
	const unsigned inmask2 = (1u << (2 * cnt)) - 1u;
	nbits &= inmask2;
	const unsigned not2 = ~is2 & inmask;
	codes_k &= inmask2;
	// 16 codes per u32, base b at bits 2b..2b+1 of the stream.  A forward k-mer is the 16 bits
	// ending at its position (oldest base in the low bits).  A reverse k-mer must pair its base t
	// with the forward k-mer's base 7-t, so the reverse stream stores each code with its two bits
	// swapped (C=2, G=1) and the probe bit-reverses the 16-bit window: the pair order flips and the
	// bits inside each pair swap back.  Only equality of k-mers matters (the reference's table index
	// is just a key), so any consistent encoding gives the same candidate set.  The simple_bayesian
	// scan compares whole 16-base words of the two streams the same way.
	if (BPL == 8) {
		reinterpret_cast<uint16_t *>(codes)[lane] = (uint16_t) codes_k;
		reinterpret_cast<uint16_t *>(nmap)[lane] = (uint16_t) nbits;
	} else {
		reinterpret_cast<uint8_t *>(codes)[lane] = (uint8_t) codes_k;
		reinterpret_cast<uint8_t *>(nmap)[lane] = (uint8_t) nbits;
	}
	const int mine = not2 ? lane * BPL + (31 - __builtin_clz(not2)) + 1 : 0;
	const u64 has = ballot(not2 != 0);
	if (has == 0)
		return 0;
	const int top = 63 - __builtin_clzll(has);
	return __builtin_amdgcn_readlane(mine, top);
}

// ---- primer offset in the MERGED sequence (`-a`: panda_compute_offset_result, offset.c:114-133) --------
// The same scan as primer_offset() over the merged bases staged in LDS: a base that agrees with the primer
// adds its log-probability p, one that does not adds log(1 - exp(p)) (panda_log1mexp, offset.c:35-38).
// Every p is an entry of the probability tables, so that second value is read from a table the HOST
// computed with its libm for each entry (P.notp, same layout as the table image) -- no device log/exp
// enters the ordered sums.  reverse: the sequence is walked from its last base (offset.c:80).
__device__ int primer_offset_result(const PamdKernelParams &P, const WaveLds &L, int len, const uint8_t *primer, int plen, bool reverse, int lane) {
	if (plen > len)
		return 0;
	const bool logdomain = P.primer_penalty == 0.0;
	double best = logdomain ? (double) plen * P.threshold : exp((double) plen * P.threshold);
	int besti = 0;
	for (int base = 0; base + plen <= len - 1; base += 64) {
		const int s = base + lane;
		const int index = s + plen;
		if (index <= len - 1) {
			double acc = 0.0;
			// the order offset.c:79-87 fills bucket s in (x ascending); the reads of eight steps -- the LDS
			// ones and the table read of log(1 - exp(p)) -- are issued together, then added in order
			int x = 0;
			for (; x + 8 <= plen; x += 8) {
				unsigned nt[8], pn[8];
				double pp[8], np[8];
#pragma unroll
				for (int u = 0; u < 8; u++) {
					const int at = reverse ? len - 1 - (s + x + u) : s + x + u;
					nt[u] = L.nts[at];
					pp[u] = L.ps[at];
					np[u] = P.notp[L.pidx[at]];
					pn[u] = primer[x + u];
				}
#pragma unroll
				for (int u = 0; u < 8; u++)
					if (pn[u] != NT_N)
						acc += ((nt[u] & pn[u]) != 0) ? pp[u] : np[u];
			}
			for (; x < plen; x++) {
				const unsigned pn = primer[x];
				if (pn != NT_N) {
					const int at = reverse ? len - 1 - (s + x) : s + x;
					const unsigned nt = L.nts[at];
					acc += ((nt & pn) != 0) ? L.ps[at] : P.notp[L.pidx[at]];
				}
			}
			double v = acc / (double) (index + 1);
			if (!logdomain)
				v = exp(v) - (double) index * P.primer_penalty;
			if (v > best) {	// strict: the earliest index wins ties (offset.c:74)
				best = v;
				besti = index + 1;
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1) {
		const double ob = __shfl_xor(best, off);
		const int oi = __shfl_xor(besti, off);
		const bool take = (oi != 0) && (besti == 0 || ob > best || (ob == best && oi < besti));
		if (take) {
			best = ob;
			besti = oi;
		}
	}
	return uni(besti);
}

// ---- counting scorers: overlap_probability for up to 4 candidates at once (algo_simple_bayes.c:33-66) ----
// The three counts come from whole words of the 2-bit code streams and special-base maps.  A candidate
// owns ROWS 16-lane DPP rows (one when both reads fit 256 bases, else two); lane c of it owns the 16
// overlap positions i in [16c, 16c + 16): forward bases fi = flen - ov + i ascending, reverse bases
// ri = rlen - 1 - i descending, so the reverse words are taken at the mirrored position and bit-reversed
// (which also swaps the bits of every entry back, see stage_read).  For bases that are exactly one of
// A C G T "(f.nt & r.nt) != 0" is equality of the codes; N on either side is an unknown.  Any other nt
// inside the compared range (IUPAC ambiguity codes, 0) is reported through `odd` and the caller scores
// that step base by base instead.
// counts of the 16 overlap positions i0 .. i0+15 of candidate `ov`: unknowns in bits 0..9, matches in bits 10..
__device__ __forceinline__ unsigned code_word_counts(const WaveLds &L, int flen, int rlen, int ov, int i0, bool active, bool &odd) {
	// positions that exist on both reads: max(0, ov - flen) <= i < min(ov, rlen)
	int lo_i = ov - flen;
	lo_i = lo_i < 0 ? 0 : lo_i;
	const int hi_i = ov < rlen ? ov : rlen;
	int lo = lo_i - i0, hi = hi_i - i0;
	lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
	hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
	const unsigned below_hi = (unsigned) ((1ull << (2 * hi)) - 1ull);
	const unsigned below_lo = (unsigned) ((1ull << (2 * lo)) - 1ull);
	const unsigned valid = active ? (below_hi & ~below_lo & 0x55555555u) : 0u;	// bit 2k <=> position i0 + k counts
	// all eight words are read unconditionally (clamped, the mask discards what does not count) so the
	// LDS round trips overlap
	int fs = flen - ov + i0, rs = rlen - 16 - i0;	// first base of the forward / reverse word
	fs = fs < -16 ? -16 : fs;
	rs = rs < -16 ? -16 : rs;
	const int fb = 2 * fs, rb = 2 * rs;
	const uint32_t *fcw = L.fc + (fb >> 5), *fnw = L.fn + (fb >> 5);
	const uint32_t *rcw = L.rc + (rb >> 5), *rnw = L.rn + (rb >> 5);
	const unsigned fc0 = fcw[0], fc1 = fcw[1], fn0 = fnw[0], fn1 = fnw[1];
	const unsigned rc0 = rcw[0], rc1 = rcw[1], rn0 = rnw[0], rn1 = rnw[1];
	const unsigned f = __builtin_amdgcn_alignbit(fc1, fc0, (unsigned) fb & 31u);
	const unsigned fsp = __builtin_amdgcn_alignbit(fn1, fn0, (unsigned) fb & 31u);
	const unsigned r = __builtin_bitreverse32(__builtin_amdgcn_alignbit(rc1, rc0, (unsigned) rb & 31u));
	const unsigned rsp = __builtin_bitreverse32(__builtin_amdgcn_alignbit(rn1, rn0, (unsigned) rb & 31u));
	const unsigned x = f ^ r;
	const unsigned same = ~(x | (x >> 1));		// bit 2k: the two codes are equal
	const unsigned sp = fsp | rsp;			// bit 2k: a base that is not plain A C G T; bit 2k+1: an N
	const unsigned isn = sp >> 1;
	odd = (valid & sp & ~isn) != 0;
	return (unsigned) __popc(valid & isn) | ((unsigned) __popc(valid & same & ~isn) << 10);	// each count <= 450 < 1024
}

template <int ALGO, int ROWS>
__device__ __forceinline__ double score_sb_codes(const PamdKernelParams &P, const WaveLds &L, int flen, int rlen, int ov, bool active, int lane, bool &odd) {
	const int c = lane & (16 * ROWS - 1);
	unsigned packed = code_word_counts(L, flen, rlen, ov, 16 * c, active, odd);
	// sum over the 16 lanes of the row (rotate-and-add inside the DPP row)
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x128, 0xF, 0xF, false);	// row_ror:8
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x124, 0xF, 0xF, false);	// row_ror:4
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x122, 0xF, 0xF, false);	// row_ror:2
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x121, 0xF, 0xF, false);	// row_ror:1
	if (ROWS == 2)
		packed += (unsigned) __builtin_amdgcn_ds_swizzle((int) packed, 0x401F);	// the other row of the pair (lane ^ 16)
	const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
	int lo_i = ov - flen;
	lo_i = lo_i < 0 ? 0 : lo_i;
	const int hi_i = ov < rlen ? ov : rlen;
	const int walked = hi_i > lo_i ? hi_i - lo_i : 0;
	return count_score<ALGO>(P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, flen, rlen, ov, unknowns, matches, walked);
}

// ---- counting scorers with many candidates: one candidate per LANE -------------------------------------
// Pairs that scan dozens to hundreds of overlaps (tandem repeats, unseeded pairs: 1.5 % of config 1, 249
// candidates each) are scored 64 candidates at a time: every lane walks ITS candidate's overlap 16 positions
// per step with the same whole-word arithmetic, so no cross-lane reduction is needed and a step costs what
// one 4-candidate step costs above.
struct ManyScan {
	double best;
	int bo, examined, done;
};

// Kept out of line on purpose: inlined into the pair loop it pushed the whole kernel over its register budget
// (spills on the common path); as a call it only costs the 1.5 % of pairs that take it.
template <int ALGO>
__device__ __attribute__((noinline, cold)) ManyScan scan_many_counting(const uint32_t *fc, const uint32_t *rc, const uint32_t *fn, const uint32_t *rn, uint16_t *cand, unsigned mybits, int minov, int flen, int rlen, double qual_nn, double pmatch, double pmismatch, const double *count_table, double best, int lane) {
	WaveLds L;
	L.fc = const_cast<uint32_t *>(fc);
	L.rc = const_cast<uint32_t *>(rc);
	L.fn = const_cast<uint32_t *>(fn);
	L.rn = const_cast<uint32_t *>(rn);
	ManyScan out;
	out.bo = -1;
	out.done = 1;
	// the candidates in ascending order, u16 each
	int total = 0;
	u64 words = ballot(mybits != 0);
	while (words != 0) {
		const int w = __builtin_ctzll(words);
		words &= words - 1;
		const unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
		if (lane < 32 && ((m >> lane) & 1u))
			cand[total + __popc(m & ((1u << lane) - 1u))] = (uint16_t) (w * 32 + lane + minov);
		total += __popc(m);
	}
	wave_sync();
	for (int c0 = 0; c0 < total; c0 += 64) {
		const bool have = c0 + lane < total;
		const int last = total - c0 > 64 ? 63 : total - c0 - 1;	// the lane that holds the batch's largest overlap
		const int ov = (int) cand[have ? c0 + lane : c0 + last];
		int lo_i = ov - flen;
		lo_i = lo_i < 0 ? 0 : lo_i;
		const int hi_i = ov < rlen ? ov : rlen;
		const int first_i = __builtin_amdgcn_readfirstlane(lo_i) & ~15;	// candidates ascend: lane 0 starts lowest
		const int end_i = __builtin_amdgcn_readlane(hi_i, last);
		unsigned packed = 0;
		bool any_odd = false;
		for (int i0 = first_i; i0 < end_i; i0 += 16) {
			bool o;
			packed += code_word_counts(L, flen, rlen, ov, i0, have, o);
			any_odd |= o;
		}
		if (ballot(any_odd && have) != 0) {
			out.done = 0;	// an IUPAC ambiguity code in a compared range: the caller scans this pair step by step
			break;
		}
		const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
		const int walked = hi_i > lo_i ? hi_i - lo_i : 0;
		double pr = count_score<ALGO>(qual_nn, pmatch, pmismatch, count_table, flen, rlen, ov, unknowns, matches, walked);
		// first maximum of this batch: larger score wins, equal scores -> smaller overlap
		int bov = have ? ov : 0x7FFFFFFF;
		if (!have)
			pr = -__builtin_inf();
		for (int o = 32; o > 0; o >>= 1) {
			const double opr = __shfl_xor(pr, o);
			const int oov = __shfl_xor(bov, o);
			if (opr > pr || (opr == pr && oov < bov)) {
				pr = opr;
				bov = oov;
			}
		}
		pr = uni_f64(pr);
		bov = uni(bov);
		if (pr > best) {	// batches are in ascending overlap order, strict > keeps the earlier one
			best = pr;
			out.bo = bov;
		}
	}
	wave_sync();
	out.best = best;
	out.examined = total;
	return out;
}

// ---- pear / rdp_mle: the per-base terms of one candidate, in L.ps[0..ov) ---------------------------
// (skipped positions get +0.0, which leaves an f64 sum that started at +0.0 unchanged)
template <int ALGO>
__device__ __forceinline__ void score_terms(const PamdKernelParams &P, const double *ltab, const WaveLds &L, int flen, int rlen, int ov, int lane) {
	double nrb = -P.pear_random_base, nn = P.qual_nn;
	asm volatile("" : "+v"(nrb), "+v"(nn));	// register copies: keep the table reads pure LDS loads
	for (int i0 = 0; i0 < ov; i0 += 64) {
		const int i = i0 + lane;
		if (i < ov) {
			const int fi = flen + i - ov;
			const int ri = rlen - i - 1;
			double term = 0.0;
			if (fi >= 0 && ri >= 0) {
				const unsigned ef = L.F[fi];
				const unsigned er = L.R[ri];
				const unsigned a = ef & 0xFFu, b = er & 0xFFu;
				const int qa = clampq(qual_of(ef));
				if (ALGO == PAMD_ALGO_PEAR) {
					if (a == NT_N || b == NT_N) {
						term = nrb;	// "probability -= random_base"
					} else {
						// second index: the FORWARD read's quality at rindex (algo_pear.c:52,54);
						// past the forward read the reference is undefined, we define quality 0
						const int qb = ri < flen ? clampq(qual_of(L.F[ri])) : 0;
						term = ltab[((a & b) != 0 ? PAMD_LT_A : PAMD_LT_B) + qa * PAMD_NQ + qb];
					}
				} else {
					const int qb = clampq(qual_of(er));
					term = ltab[((a & b) != 0 ? PAMD_LT_A : PAMD_LT_B) + qa * PAMD_NQ + qb] - nn;
				}
			}
			L.ps[i] = term;
		}
	}
	wave_sync();
}

// the reference's own summation order, by one lane
__device__ __forceinline__ double sum_terms_exact(const WaveLds &L, int ov, int lane) {
	double acc = 0.0;
	if (lane == 0) {
		int i = 0;
		for (; i + 8 <= ov; i += 8) {
			const double v0 = L.ps[i], v1 = L.ps[i + 1], v2 = L.ps[i + 2], v3 = L.ps[i + 3];
			const double v4 = L.ps[i + 4], v5 = L.ps[i + 5], v6 = L.ps[i + 6], v7 = L.ps[i + 7];
			acc += v0;
			acc += v1;
			acc += v2;
			acc += v3;
			acc += v4;
			acc += v5;
			acc += v6;
			acc += v7;
		}
		for (; i < ov; i++)
			acc += L.ps[i];
	}
	wave_sync();
	return uni_f64(acc);
}

// any-order sum of the same terms: only used to DECIDE between candidates whose scores differ by far
// more than its rounding error (|error| < 1e-9 for 450 terms of magnitude < 1e3)
__device__ __forceinline__ double sum_terms_fast(const WaveLds &L, int ov, int lane) {
	double acc = 0.0;
	for (int i = lane; i < ov; i += 64)
		acc += L.ps[i];
	acc = wave_sum_any_order(acc);
	wave_sync();
	return acc;
}

// ---- overlap_probability, one candidate ------------------------------------------------------
template <int ALGO>
__device__ __forceinline__ double score_overlap(const PamdKernelParams &P, const double *ltab, const WaveLds &L, int flen, int rlen, int ov, int lane) {
	if (counting_algo<ALGO>()) {
		// algo_simple_bayes.c:33-66 and the other counting scorers: three integer counts, then a closed form
		int unknowns = 0, matches = 0, mismatches = 0;
		for (int i0 = 0; i0 < ov; i0 += 128) {	// two overlap positions per lane: four LDS reads in flight
			const int ia = i0 + lane, ib = i0 + 64 + lane;
			const int fa = flen + ia - ov, ra = rlen - ia - 1;
			const int fb = flen + ib - ov, rb = rlen - ib - 1;
			const bool ina = ia < ov && fa >= 0 && ra >= 0;
			const bool inb = ib < ov && fb >= 0 && rb >= 0;
			const unsigned a0 = L.F[ina ? fa : 0] & 0xFFu, b0 = L.R[ina ? ra : 0] & 0xFFu;
			const unsigned a1 = L.F[inb ? fb : 0] & 0xFFu, b1 = L.R[inb ? rb : 0] & 0xFFu;
			const bool n0 = a0 == NT_N || b0 == NT_N, n1 = a1 == NT_N || b1 == NT_N;
			const bool m0 = (a0 & b0) != 0, m1 = (a1 & b1) != 0;
			unknowns += __popcll(ballot(ina && n0)) + __popcll(ballot(inb && n1));
			matches += __popcll(ballot(ina && !n0 && m0)) + __popcll(ballot(inb && !n1 && m1));
			mismatches += __popcll(ballot(ina && !n0 && !m0)) + __popcll(ballot(inb && !n1 && !m1));
		}
		return count_score<ALGO>(P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, flen, rlen, ov, unknowns, matches, unknowns + matches + mismatches);
	} else {
		// algo_pear.c:32-59 / algo_rdp_mle.c:43-74: a sequential f64 sum over i = 0..ov-1.
		// Lanes produce the terms, lane 0 adds them in order.
		score_terms<ALGO>(P, ltab, L, flen, rlen, ov, lane);
		return sum_terms_exact(L, ov, lane);
	}
}

// ---- pear / rdp_mle with many candidates: a short and a long candidate per lane -------------------------------
// A pair without a seeded overlap has every overlap as a candidate (assembler.c:111-117): some thirty thousand table
// lookups, each candidate an ordered f64 chain (algo_pear.c:42-56, algo_rdp_mle.c:55-72).  Every lane walks its own
// candidates' overlaps in index order and adds the terms as the reference does -- first candidate `ova`, then candidate `ovb`
// (0 = none) -- and the candidates are dealt so that the lanes finish together: of 128 candidates in ascending order lane l
// gets numbers l and 127 - l, a short overlap with a long one.  The walk reads a digest of the reads made once per pair
// (clamped qualities, the forward quality already times 47): `sf[i]` = nt | 47 q << 8 | q << 20 of forward base i, `sr[i]` =
// nt | q << 8 of reverse base i.  INSIDE: no candidate is longer than either read (always, unless a maximum overlap is
// configured), so no position falls outside a read.
template <int ALGO, bool INSIDE>
__device__ __forceinline__ void score_two_overlaps_serial(const double *ltab, const uint32_t *sf, const uint16_t *sr, int flen, int rlen, int ova, int ovb, double neg_random_base, double qual_nn, double &pra, double &prb) {
	double acc = 0.0, first = 0.0;
	int fi = flen - ova, ri = rlen - 1;
	int rem = ova;
	bool second = false;
	const int steps = ova + ovb;
	for (int t = 0; t < steps; t++) {
		const bool ok = INSIDE || (fi >= 0 && ri >= 0);
		const unsigned wf = sf[INSIDE || ok ? fi : 0];
		const unsigned wr = sr[INSIDE || ok ? ri : 0];
		const unsigned a = wf & 0xFFu, b = wr & 0xFFu;
		const unsigned base = (a & b) != 0 ? PAMD_LT_A : PAMD_LT_B;
		double term;
		if (ALGO == PAMD_ALGO_PEAR) {
			const unsigned qb = ((INSIDE || ok) && ri < flen) ? sf[ri] >> 20 : 0u;	// sic: the forward quality at rindex (0 past the read)
			const double v = ltab[base + ((wf >> 8) & 0xFFFu) + qb];
			term = (a == NT_N || b == NT_N) ? neg_random_base : v;
		} else {
			term = ltab[base + ((wf >> 8) & 0xFFFu) + (wr >> 8)] - qual_nn;
		}
		acc += (INSIDE || ok) ? term : 0.0;	// (a position outside a read adds 0.0: no sum here is -0.0)
		fi++;
		ri--;
		rem--;
		if (rem == 0 && !second) {	// on to this lane's second candidate
			first = acc;
			acc = 0.0;
			second = true;
			rem = ovb;
			fi = flen - ovb;
			ri = rlen - 1;
		}
	}
	pra = second ? first : acc;
	prb = second ? acc : 0.0;
}

struct WaveArena {
	u64 cur;
	u64 end;
};

// ---- deferred finish ---------------------------------------------------------------------------
// The three quality sums of a pair (assembler.c:171,226,236) are ordered f64 chains: one lane can
// advance one chain, so doing them per pair keeps 3 of 64 lanes busy for ~200 dependent adds.
// Instead every wave parks each merged pair in a small LDS record, and once group_size<ALGO>() pairs are parked
// (or its work runs out) finishes them together: lane 3*slot + chain walks one chain of one pair,
// reading the log-probabilities back from the pair's output row (just written, L2/MALL resident),
// so 48 lanes advance 48 chains with the same instructions.  Then lane `slot` finishes pair `slot`:
// quality, LOWQ, check modules, the 48-byte record and the counters.
// 16 pairs per group for the counting scorers (three chains per pair: 48 lanes busy), 8 for pear / rdp_mle (four chains,
// and their larger tables leave less LDS for parked records: a sixteen-pair group costs them a wave per CU)
template <int ALGO>
__device__ __forceinline__ constexpr int group_size() {
	return (ALGO == PAMD_ALGO_PEAR || ALGO == PAMD_ALGO_RDP_MLE) ? 8 : 16;
}
constexpr int SCORE_STRIDE = 456;	// doubles per parked score-term row
#ifndef FINISH_DEPTH
#define FINISH_DEPTH 6	/* 16-byte loads in flight per chain */
#endif

struct PendRec {	// 80 bytes
	u64 pair;
	u64 off;	// row offset in seq_nt / seq_p, ~0 = no row (failed before the merge, or arena full)
	double best;	// estimated_overlap_probability
	int st;		// status so far: BADR / NOFP / NORP / NOALGN, or OK = merged, sums pending
	int seqlen, A, novl, len;
	int bo, examined, mism, degen, fo, ro, slow;
	int phmin;	// smallest panda_result_phred over the merged bases (only when a min_phred module is loaded)
	int score_n;	// pear / rdp_mle: > 0 = `best` is still to be summed from this many parked terms
};

static_assert(sizeof(PendRec) == 80, "PendRec is sized by pamd_api.cpp");

template <int ALGO>
__device__ void finish_group(const PamdKernelParams &P, const double *ltab, unsigned *lctr, const PendRec *pend, int npend, unsigned wave_gid, int lane) {
	constexpr int NCH = counting_algo<ALGO>() ? 3 : 4;	// chains per pair
	// all of this wave's row stores must have reached L2 before they are read back
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	wave_sync();
	// ---- ordered sums, one chain per lane ----
	const int slot = lane / NCH, chain = lane - NCH * slot;
	double acc = 0.0;
	{
		// chain 0: forward-only [0,A)   chain 1: reverse-only [A+novl, seqlen)   chain 2: overlap [A, A+novl)
		// Plain cached loads are correct here: the rows were written by this very wave, the wait above
		// retired those stores, and a CU's vector L1 is coherent for its own stores.  Two values per
		// load: one memory request serves two adds, and the eight loads of a 128-byte line hit in L1.
		int j = 0, end = 0;
		const double *rowp = P.seq_p;
		if (slot < npend) {
			const PendRec &r = pend[slot];
			if (chain == 3) {
				// pear / rdp_mle: the winning candidate's terms, parked by assemble_pair()
				end = r.score_n;
				rowp = P.score_scratch + ((size_t) wave_gid * group_size<ALGO>() + (size_t) slot) * SCORE_STRIDE;
			} else if (r.st == PAMD_ST_OK && r.off != ~0ull) {
				j = chain == 0 ? 0 : (chain == 1 ? r.A + r.novl : r.A);
				end = chain == 0 ? r.A : (chain == 1 ? r.seqlen : r.A + r.novl);
				rowp = P.seq_p + r.off;	// 128-byte aligned
			}
		}
		if (j < end && (j & 1)) {	// reach an even index
			acc += rowp[j];
			j++;
		}
		// same left-to-right order as the reference
		for (; j + 2 * FINISH_DEPTH <= end; j += 2 * FINISH_DEPTH) {
			double2 v[FINISH_DEPTH];
#pragma unroll
			for (int t = 0; t < FINISH_DEPTH; t++)
				v[t] = *reinterpret_cast<const double2 *>(rowp + j + 2 * t);
#pragma unroll
			for (int t = 0; t < FINISH_DEPTH; t++) {
				acc += v[t].x;
				acc += v[t].y;
			}
		}
		if (j < end) {
			// the last, partial batch at the same depth (a one-at-a-time tail would pay one memory
			// round trip per add); slots past the end re-read the last pair and are added as +0.0,
			// which leaves a sum that is negative or +0.0 unchanged
			const int lastpair = (end - 1) & ~1;
			double2 v[FINISH_DEPTH];
#pragma unroll
			for (int t = 0; t < FINISH_DEPTH; t++) {
				const int jj = j + 2 * t < end ? j + 2 * t : lastpair;
				v[t] = *reinterpret_cast<const double2 *>(rowp + jj);
			}
#pragma unroll
			for (int t = 0; t < FINISH_DEPTH; t++) {
				acc += (j + 2 * t < end) ? v[t].x : 0.0;
				acc += (j + 2 * t + 1 < end) ? v[t].y : 0.0;
			}
		}
	}
	// ---- one lane per pair ----
	const int src = lane < group_size<ALGO>() ? NCH * lane : 0;
	const double fquality = __shfl(acc, src);
	const double rquality = __shfl(acc, src + 1);
	const double oquality = __shfl(acc, src + 2);
	const double parked_score = NCH == 4 ? __shfl(acc, src + 3) : 0.0;
	int status = -1, reject = -1, overlap = 0;
	if (lane < npend) {
		PendRec r = pend[lane];
		if (NCH == 4 && r.score_n > 0)
			r.best = parked_score;	// the ordered sum of the parked terms = estimated_overlap_probability
		status = r.st;
		double quality = 0.0;
		u64 off = ~0ull;
		if (status == PAMD_ST_OK && r.off == ~0ull) {
			atomicAdd(P.error_flag, 1ull);	// the output arena was too small for this row (see finish64)
			status = PAMD_ST_NOSPACE;
		}
		if (status == PAMD_ST_OK) {
			quality = (fquality + rquality + oquality) / (double) r.len;	// assembler.c:244
			if (quality < P.threshold) {	// assembler.c:334-338
				status = PAMD_ST_LOWQ;
			} else {
				for (int mi = 0; mi < P.n_modules && reject < 0; mi++) {	// module.c:124-136
					const PamdModule md = P.modules[mi];
					if (md.kind == 1) {	// plugin_min_phred.c:8-22
						if (r.phmin < md.phred)
							reject = mi;
					} else if (md.kind == 2) {	// plugin_min_overlapbits.c:17-23
						if (!(md.nats <= r.best))
							reject = mi;
					}
				}
				if (reject >= 0)
					status = PAMD_ST_REJECT;
				else
					off = r.off;
			}
		}
		overlap = r.bo;
		// pamd_pair_result, 48 bytes, as three 16-byte stores
		union {
			double d;
			unsigned u[2];
		} q, e;
		q.d = quality;
		e.d = r.st == PAMD_ST_OK ? r.best : 0.0;	// (assembler.c:247: set once the pair has merged, as in finish64)
		uint4 *dst = reinterpret_cast<uint4 *>(&P.results[r.pair]);
		dst[0] = make_uint4(q.u[0], q.u[1], e.u[0], e.u[1]);
		dst[1] = make_uint4((unsigned) off, (unsigned) (off >> 32), (unsigned) r.seqlen | ((unsigned) r.bo << 16), (unsigned) r.examined | ((unsigned) r.mism << 16));
		dst[2] = make_uint4((unsigned) r.degen | ((unsigned) r.fo << 16), (unsigned) r.ro | ((unsigned) status << 16) | (((unsigned) reject & 0xFFu) << 24), (unsigned) r.slow, 0u);
		if (r.slow)
			atomicAdd(&lctr[PAMD_CTR_SLOW], 1u);
		if (status == PAMD_ST_OK)
			atomicAdd(&lctr[PAMD_CTR_OVERLAP0 + overlap], 1u);
		else if (status == PAMD_ST_REJECT)
			atomicAdd(&lctr[PAMD_CTR_REJECTED0 + reject], 1u);
	}
	// workgroup-local counters (assembler.h:59-67,75): one LDS atomic per status per group
	const int ctr_of[6] = { PAMD_CTR_OK, PAMD_CTR_BADR, PAMD_CTR_NOFP, PAMD_CTR_NORP, PAMD_CTR_NOALGN, PAMD_CTR_LOWQ };
#pragma unroll
	for (int st = 0; st < 6; st++) {
		const int c = __popcll(ballot(status == st));
		if (c != 0 && lane == 0)
			atomicAdd(&lctr[ctr_of[st]], (unsigned) c);
	}
	if (lane == 0)
		atomicAdd(&lctr[PAMD_CTR_COUNT], (unsigned) npend);
	wave_sync();
}

__device__ __forceinline__ void park64(Pend64 &pd, int slot_idx, const PendRec &rec, int status, double q, bool q_final) {
	const bool slot = (int) (threadIdx.x & 63u) == slot_idx;
	union {
		double d;
		unsigned u[2];
	} qq, bb;
	qq.d = q;
	bb.d = rec.best;
	pd.off_lo = put_lane(pd.off_lo, (unsigned) rec.off, slot);
	pd.off_hi = put_lane(pd.off_hi, (unsigned) (rec.off >> 32), slot);
	pd.q_lo = put_lane(pd.q_lo, qq.u[0], slot);
	pd.q_hi = put_lane(pd.q_hi, qq.u[1], slot);
	pd.b_lo = put_lane(pd.b_lo, bb.u[0], slot);
	pd.b_hi = put_lane(pd.b_hi, bb.u[1], slot);
	pd.len_bo = put_lane(pd.len_bo, (unsigned) rec.seqlen | ((unsigned) rec.bo << 16), slot);
	pd.ex_mm = put_lane(pd.ex_mm, (unsigned) rec.examined | ((unsigned) rec.mism << 16), slot);
	pd.dg_fo = put_lane(pd.dg_fo, (unsigned) rec.degen | ((unsigned) rec.fo << 16), slot);
	pd.ro_st = put_lane(pd.ro_st, (unsigned) rec.ro | ((unsigned) status << 16) | ((unsigned) (rec.slow != 0) << 24) | ((unsigned) q_final << 25) | ((unsigned) (rec.phmin > 63 ? 63 : rec.phmin) << 26), slot);
}
// quality from the any-order sum (|error| ~1e-12); the LOWQ verdict is the only consumer that could notice,
// so within 1e-9 of the threshold (or always, for `-a` in exact-sums mode, which has no group finish) the
// three sums are redone in reference order from the staged row (assembler.c:171,226,236,244)
__device__ __forceinline__ double quality_sum(double qacc) {	// the any-order sum over the wave
	return wave_sum_any_order(qacc);
}

// `row`: the pair's log-p row -- staged in LDS (`-a`), or the one this wave has just written to HBM (plain cached loads:
// wave_sync() retires the stores, and a CU's vector L1 is coherent for its own stores, as in finish_group())
__device__ __forceinline__ double fast_quality(const double *row, double qacc, double threshold, int len, int seqlen, int A, int novl, bool always_exact, int lane) {
	double quality = quality_sum(qacc) / (double) len;
	if (always_exact || fabs(quality - threshold) <= 1e-9) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// a row in HBM: its stores must have reached L2
		wave_sync();
		double acc = 0.0;
		int lo = 0, hi = 0;
		if (lane == 0) {
			hi = A;
		} else if (lane == 1) {
			lo = A + novl;
			hi = seqlen;
		} else if (lane == 2) {
			lo = A;
			hi = A + novl;
		}
		for (int j = lo; j < hi; j++)
			acc += row[j];
		const double fq = lane_f64(acc, 0), rq = lane_f64(acc, 1), oq = lane_f64(acc, 2);
		quality = (fq + rq + oq) / (double) len;	// assembler.c:244
	}
	return quality;
}

// the registers a wave carries from one pair to the next (software prefetch)
struct Carry {
	int flen_v, rlen_v;
	unsigned primers_v;	// `-p` / `-q`: what the primer kernel found in this pair (pamd_primer.hip)
	uint4 f, r;
};

// Pins the carried registers (the compiler must treat them as written here).  With four bases per lane a row is two
// words per lane: the upper two words of each uint4 are constant zero and are left to the compiler.
template <int BPL>
__device__ __forceinline__ void settle_carry(Carry &c) {
	if (BPL == 8) {
		asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.primers_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.f.z), "+v"(c.f.w), "+v"(c.r.x), "+v"(c.r.y), "+v"(c.r.z), "+v"(c.r.w));
	} else {
		asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.primers_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.r.x), "+v"(c.r.y));
		c.f.z = c.f.w = c.r.z = c.r.w = 0u;
	}
}

// ---- one pair: everything up to its parked record ----------------------------------------------
template <int ALGO, bool PHRED, int BPL, bool EXACT, bool POST>
__device__ void assemble_pair(const PamdKernelParams &P, const double *ltab, const uint8_t *lprimers, unsigned *lctr, const WaveLds &L, WaveArena &arena, PendRec *slot_rec, Pend64 &pd, int slot_idx, unsigned wave_gid, u64 pair, int lane, const int flen, const int rlen, Carry &cur, const Carry &nxt
#ifdef PAMD_PHASE_TIMING
	, u64 *ph_acc
#endif
) {
	const uint4 fregs = cur.f, rregs = cur.r;
	const unsigned primers_found = (unsigned) uni((int) cur.primers_v);
	PH_DECL;
	bool rotated = false;
	double quality_fast = 0.0;
	bool quality_final_flag = POST;	// quality_fast is the finished quality, not the sum still to be divided
	const int minov = P.minoverlap;
	const unsigned hmask = P.hash_slots - 1u;
	const unsigned hshift = P.hash_shift;

	PendRec rec;
	rec.pair = pair;
	rec.off = ~0ull;
	rec.best = 0.0;
	rec.st = PAMD_ST_OK;
	rec.seqlen = 0;
	rec.A = 0;
	rec.novl = 0;
	rec.len = 1;
	rec.bo = 0;
	rec.examined = 0;
	rec.mism = 0;
	rec.degen = 0;
	rec.fo = 0;
	rec.ro = 0;
	rec.slow = 0;
	rec.phmin = 127;
	rec.score_n = 0;
	int status = PAMD_ST_OK;

	do {
		if (flen < 2 || rlen < 2) {	// assembler.c:255-258
			status = PAMD_ST_BADR;
			break;
		}
		// previous pair's readers are done with this wave's LDS
		wave_sync();
		const int uf = stage_read<true, BPL>(fregs, flen, lane, L.F, L.fc, L.fn);
		const int ur = stage_read<false, BPL>(rregs, rlen, lane, L.R, L.rc, L.rn);
		// clear the k-mer table (aliases the log-p staging area) and the candidate bitmap
		for (unsigned s4 = lane; s4 < P.hash_slots / 4; s4 += 64)
			reinterpret_cast<uint4 *>(L.hash)[s4] = make_uint4(HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY);
		if (lane < 32)
			L.bits[lane] = 0;
		wave_sync();
		PH_MARK(0);

		// ---- primer offsets / trims (assembler.c:262-288) ----
		constexpr bool post = POST;	// `-a`: primers are looked for in the merged sequence instead (own kernel instantiation)
		int fo, ro;
		if (post) {
			fo = 0;
		} else if (P.fprimer_len > 0) {
			const int r = (int) (primers_found & 0xFFFFu);	// found by the primer kernel (pamd_primer.hip), prefetched with the reads
			if (r == 0) {
				status = PAMD_ST_NOFP;
				break;
			}
			fo = r - 1;
		} else {
			fo = P.ftrim;
		}
		rec.fo = fo;
		if (post) {
			ro = 0;
		} else if (P.rprimer_len > 0) {
			const int r = (int) (primers_found >> 16);
			if (r == 0) {
				status = PAMD_ST_NORP;
				break;
			}
			ro = r - 1;
		} else {
			ro = P.rtrim;
		}
		rec.ro = ro;
		if ((flen < rlen ? flen : rlen) < minov) {	// assembler.c:289-292
			status = PAMD_ST_BADR;
			break;
		}

		// ---- align(): bounds (assembler.c:59,73-84) ----
		if (minov + fo >= flen || minov + ro >= rlen) {
			status = PAMD_ST_NOALGN;
			break;
		}
		// size_t arithmetic of assembler.c:59 done in 64 bits (it can wrap when offsets are huge)
		u64 maxov64 = (u64) flen + (u64) rlen - (u64) minov - (u64) fo - (u64) ro - 1ull;
		if (P.maxoverlap == 0)
			maxov64 = (u64) (flen < rlen ? flen : rlen);
		else if (maxov64 > (u64) P.maxoverlap)
			maxov64 = (u64) P.maxoverlap;
		const int maxov = (int) maxov64;	// <= 2 * PAMD_MAX_LEN by construction
		const int size = minov <= maxov ? maxov - minov + 1 : 1;

		PH_MARK(1);
		// ---- seed: forward k-mers into the LDS table (assembler.c:92-101) ----
		// The reference keeps, per k-mer, the first two forward positions.  We insert every
		// (k-mer, position) into a linear-probing multiset and let the reverse probe pick the two
		// smallest positions of its k-mer, which is the same set.
		for (int base = 0; base < flen; base += 64) {
			const int p = base + lane;
			const bool inr = p >= 8 && p < flen;
			// both windows are read unconditionally (clamped) so the two LDS round trips overlap
			const int sn = inr ? p - 8 : 0;	// no N among the 9 positions ending at p (misc.h:41)
			const unsigned nwin = __builtin_amdgcn_alignbit(L.fn[(sn >> 4) + 1], L.fn[sn >> 4], 2u * ((unsigned) sn & 15u)) & 0x2AAAAu;	// N = high bit of an entry
			// the k-mer is the eight codes ending at p: entries 1..8 of the same nine-entry window (one address computation)
			unsigned kmer = (__builtin_amdgcn_alignbit(L.fc[(sn >> 4) + 1], L.fc[sn >> 4], 2u * ((unsigned) sn & 15u)) >> 2) & 0xFFFFu;
			asm volatile("" : "+v"(kmer));	// pins the second window read above the branch (the compiler sinks it into it otherwise)
			if (inr && nwin == 0) {
				const unsigned word = (kmer << 16) | (unsigned) p;
				// A chain starts on the k-mer's slot PAIR (the probe reads pairs); which of the pair's two slots is tried
				// first is one more hash bit, so two lanes that meet in a pair collide only half the time; then the
				// other slot of the pair, then slot by slot onward.
				unsigned slot = kmer_hash(kmer, hshift);
				if (atomicCAS(&L.hash[slot], HASH_EMPTY, word) != HASH_EMPTY && atomicCAS(&L.hash[slot ^ 1u], HASH_EMPTY, word) != HASH_EMPTY) {
					slot = ((slot | 1u) + 1u) & hmask;
					while (atomicCAS(&L.hash[slot], HASH_EMPTY, word) != HASH_EMPTY)
						slot = (slot + 1) & hmask;
				}
			}
		}
		wave_sync();
		PH_MARK(2);
		// ---- seed: reverse k-mers flag candidate overlaps (assembler.c:104-110) ----
		for (int base = 0; base < rlen; base += 64) {
			const int p = base + lane;
			const bool inr = p + 8 < rlen;	// p <= rlen - 9
			const int sp = inr ? p : 0;	// no N among positions p .. p+8
			const unsigned nwin = __builtin_amdgcn_alignbit(L.rn[(sp >> 4) + 1], L.rn[sp >> 4], 2u * ((unsigned) sp & 15u)) & 0x15555u;	// reverse entries are bit-swapped
			unsigned x = __builtin_amdgcn_alignbit(L.rc[(sp >> 4) + 1], L.rc[sp >> 4], 2u * ((unsigned) sp & 15u)) & 0xFFFFu;
			asm volatile("" : "+v"(x));	// as above
			if (inr && nwin == 0) {
				const unsigned kmer = __builtin_bitreverse32(x) >> 16;
				const unsigned kk = kmer << 16;
				unsigned slot = kmer_hash(kmer, hshift) & ~1u;
				// Walk the probe chain one aligned slot pair per step and keep the two smallest forward
				// positions of this k-mer, which are the reference's first two.  entry ^ (kmer << 16) is the
				// bare position (< 0x10000) for an entry of this k-mer and >= 0x10000 for any other k-mer;
				// an empty slot gives >= 0xFFFF, above every real position, so plain unsigned minima do it.
				// first step peeled: with nothing found yet the two-smallest network is one min and one max
				const uint2 w0 = *reinterpret_cast<const uint2 *>(&L.hash[slot]);
				unsigned m1 = (w0.x ^ kk) < (w0.y ^ kk) ? (w0.x ^ kk) : (w0.y ^ kk);
				unsigned m2 = (w0.x ^ kk) < (w0.y ^ kk) ? (w0.y ^ kk) : (w0.x ^ kk);
				slot = (slot + 2) & hmask;
				bool more = w0.x != HASH_EMPTY && w0.y != HASH_EMPTY;
				while (more) {
					const uint2 w = *reinterpret_cast<const uint2 *>(&L.hash[slot]);
					const unsigned x0 = w.x ^ kk, x1 = w.y ^ kk;
					const unsigned lo0 = x0 < m1 ? x0 : m1, hi0 = x0 < m1 ? m1 : x0;
					const unsigned lo1 = x1 < lo0 ? x1 : lo0, hi1 = x1 < lo0 ? lo0 : x1;
					const unsigned h = hi0 < hi1 ? hi0 : hi1;
					m2 = h < m2 ? h : m2;
					m1 = lo1;
					slot = (slot + 2) & hmask;
					more = w.x != HASH_EMPTY && w.y != HASH_EMPTY;
				}
				if (m1 < 0xFFFFu) {
					const int idx = flen + rlen - p - (int) m1 - minov - 1;	// assembler.c:107
					if (idx >= 0 && idx < size)
						atomicOr(&L.bits[idx >> 5], 1u << (idx & 31));
					if (m2 < 0xFFFFu) {
						const int idx2 = flen + rlen - p - (int) m2 - minov - 1;
						if (idx2 >= 0 && idx2 < size)
							atomicOr(&L.bits[idx2 >> 5], 1u << (idx2 & 31));
					}
				}
			}
		}
		wave_sync();
		PH_MARK(3);

		// ---- scan + argmax (assembler.c:118-143) ----
		const int lastw = size >> 5;
		unsigned mybits = 0;
		const unsigned keep = lane < lastw ? 0xFFFFFFFFu : (lane == lastw ? ((1u << (size & 31)) - 1u) : 0u);
		if (lane <= lastw && lane < 32)
			mybits = L.bits[lane] & keep;
		if (ballot(mybits != 0) == 0)
			mybits = keep;	// ALL_BITS_IF_NONE
		wave_sync();	// the scan below reuses the staging area the k-mer table lived in

		double best = P.qual_nn * (double) (flen + rlen);	// assembler.c:60
		int bo = -1;
		int examined = 0;
		int score_parked = 0;	// pear / rdp_mle: the winner's terms sit in L.ps and must be parked
		int ncand = 0;
		constexpr int MANY = counting_algo<ALGO>() ? 24 : 4;	// from here on a candidate per lane pays
		{
			u64 nz = ballot(mybits != 0);
			while (nz != 0 && ncand <= MANY) {	// only "more than MANY?" matters
				const int w = __builtin_ctzll(nz);
				nz &= nz - 1;
				ncand += __popc((unsigned) __builtin_amdgcn_readlane((int) mybits, w));
			}
		}
		bool per_lane_done = false;
		if (counting_algo<ALGO>() && ncand > MANY) {
			// Many candidates: 64 at a time, one per lane (scan_many_counting).  An IUPAC ambiguity code inside
			// a compared range abandons that path: the pair is then scanned by the step path below.
			const ManyScan ms = scan_many_counting<ALGO>(L.fc, L.rc, L.fn, L.rn, reinterpret_cast<uint16_t *>(L.ps), mybits, minov, flen, rlen, P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, best, lane);
			if (uni(ms.done)) {	// a function's results come back in vector registers: make them wave-uniform again
				per_lane_done = true;
				best = uni_f64(ms.best);
				bo = uni(ms.bo);
				examined = uni(ms.examined);
			}
		}
		if (counting_algo<ALGO>() && ncand == 1) {
			// The usual case, one seeded overlap: scored by one DPP row (pair of rows for long reads) without
			// the candidate bookkeeping below.  An IUPAC ambiguity code in range falls through to it.
			constexpr int ROWS1 = BPL == 4 ? 1 : 2;
			const int w = __builtin_ctzll(ballot(mybits != 0));
			const unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
			const int ov = w * 32 + __builtin_ctz(m) + minov;
			bool odd;
			const double sc = score_sb_codes<ALGO, ROWS1>(P, L, flen, rlen, ov, lane < 16 * ROWS1, lane, odd);
			if (ballot(odd) == 0) {
				const double pr = lane_f64(sc, 0);
				if (pr > best) {
					best = pr;
					bo = ov;
				}
				examined = 1;
				per_lane_done = true;
			}
		}
		if (counting_algo<ALGO>() && !per_lane_done) {
			// Candidates are scored PER_STEP at a time from the packed code streams, one candidate per
			// DPP row (pair of rows for long reads), and judged in ascending order.
			constexpr int ROWS = BPL == 4 ? 1 : 2;
			constexpr int PER_STEP = 4 / ROWS;
			const int row = lane / (16 * ROWS);
			int myov = 0;	// this row's candidate
			int pending = 0;
			u64 words = ballot(mybits != 0);	// non-empty 32-candidate words, ascending
			while (words != 0) {
				const int w = __builtin_ctzll(words);
				words &= words - 1;
				unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
				const bool lastword = words == 0;
				while (m != 0) {
					const int b = __builtin_ctz(m);
					m &= m - 1;
					const int ov = w * 32 + b + minov;
					myov = row == pending ? ov : myov;
					pending++;
					if (pending == PER_STEP || (lastword && m == 0)) {
						bool odd;
						const double sc = score_sb_codes<ALGO, ROWS>(P, L, flen, rlen, myov, row < pending, lane, odd);
						const bool slowstep = ballot(odd) != 0;	// an nt that is neither A C G T nor N: base by base
						for (int r = 0; r < pending; r++) {
							const int ovr = __builtin_amdgcn_readlane(myov, 16 * ROWS * r);
							const double pr = slowstep ? score_overlap<ALGO>(P, ltab, L, flen, rlen, ovr, lane) : lane_f64(sc, 16 * ROWS * r);
							if (pr > best) {	// strict: ascending scan keeps the smallest overlap on ties
								best = pr;
								bo = ovr;
							}
						}
						examined += pending;
						pending = 0;
					}
				}
			}
		} else if (!counting_algo<ALGO>() && ncand > 4) {
			// Many candidates: list them in ascending order (u16 overlaps in the staging area), then
			// up to 128 at a time, two per lane, each lane summing its own candidates in reference order.
			uint16_t *cand = reinterpret_cast<uint16_t *>(L.ps);
			int total = 0;
			u64 words = ballot(mybits != 0);
			while (words != 0) {
				const int w = __builtin_ctzll(words);
				words &= words - 1;
				const unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
				if (lane < 32 && ((m >> lane) & 1u))
					cand[total + __popc(m & ((1u << lane) - 1u))] = (uint16_t) (w * 32 + lane + minov);
				total += __popc(m);
			}
			// the reads' digest for the walk, behind the list (score_two_overlaps_serial)
			uint32_t *const sf = reinterpret_cast<uint32_t *>(cand + ((total + 1) & ~1));
			uint16_t *const sr = reinterpret_cast<uint16_t *>(sf + flen);
			for (int i = lane; i < flen; i += 64) {
				const unsigned e = L.F[i];
				const unsigned q = (unsigned) clampq(qual_of(e));
				sf[i] = (e & 0xFFu) | ((q * PAMD_NQ) << 8) | (q << 20);
			}
			for (int i = lane; i < rlen; i += 64) {
				const unsigned e = L.R[i];
				sr[i] = (uint16_t) ((e & 0xFFu) | ((unsigned) clampq(qual_of(e)) << 8));
			}
			// Copies of the special-base maps in which a base of low quality counts as an N (behind the digest, a word of
			// slack either side as code_word_counts reads): what they call "unknown" is then a position whose term is bounded
			// by many_all only, a match among the rest by many_match, a mismatch among the rest by many_strong.  The quality
			// that counts is the one the term is looked up with: the reverse base's for rdp_mle, the FORWARD read's at the
			// reverse index for pear (algo_pear.c:52,54; 0 past the read).
			uint32_t *const fnw = reinterpret_cast<uint32_t *>(sr + ((rlen + 1) & ~1)) + 1;
			uint32_t *const rnw = fnw + 34;
			wave_sync();	// (the digest is read below)
			{
				const int w = lane & 31;
				const bool fwd = lane < 32;
				unsigned v = fwd ? L.fn[w] : L.rn[w];
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const int at = 16 * w + k;
					unsigned q = 0;	// (past the read: nothing there counts anyway)
					if (fwd || ALGO == PAMD_ALGO_PEAR)
						q = at < flen ? sf[at] >> 20 : 0u;
					else
						q = at < rlen ? (unsigned) sr[at] >> 8 : 0u;
					if (q < (unsigned) PAMD_STRONG_Q)
						v |= 3u << (2 * k);
				}
				(fwd ? fnw : rnw)[w] = v;
				if (w == 0) {
					(fwd ? fnw : rnw)[-1] = 0xFFFFFFFFu;
					(fwd ? fnw : rnw)[32] = 0xFFFFFFFFu;
				}
			}
			WaveLds LW = L;
			LW.fn = fnw;
			LW.rn = rnw;
			wave_sync();
			double nrb = -P.pear_random_base, nn = P.qual_nn;
			asm volatile("" : "+v"(nrb), "+v"(nn));	// register copies: keep the table reads pure LDS loads
			const bool inside = (int) cand[total - 1] <= (flen < rlen ? flen : rlen);	// the longest candidate
			const double nothing_yet = best;
			// an upper bound of candidate ov's score from three counts (16 positions per step, whole-word arithmetic)
			auto bound_of = [&](int ov) -> double {
				int lo_i = ov - flen;
				lo_i = lo_i < 0 ? 0 : lo_i;
				const int hi_i = ov < rlen ? ov : rlen;
				unsigned packed = 0;
				bool any_odd = false;
				for (int i0 = lo_i & ~15; i0 < hi_i; i0 += 16) {
					bool o;
					packed += code_word_counts(LW, flen, rlen, ov, i0, true, o);
					any_odd |= o;
				}
				const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
				const int walked = hi_i > lo_i ? hi_i - lo_i : 0;
				const double b = (double) unknowns * P.many_all + (double) matches * P.many_match + (double) (walked - unknowns - matches) * P.many_strong;
				return any_odd ? __builtin_inf() : b;	// (an ambiguity code: the codes say nothing about agreement)
			};
			// (the sixteen shortest candidates first, by themselves: they cost next to nothing, and what they find is what
			// the bounds of all the others are held against)
			for (int c0 = 0, group = 16; c0 < total; c0 += group, group = 128) {
				const bool mine = 2 * lane < group;
				const int ja = c0 + lane, jb = c0 + group - 1 - lane;
				int ova = (mine && ja < total) ? (int) cand[ja] : 0, ovb = (mine && jb < total) ? (int) cand[jb] : 0;
				if (best != nothing_yet) {
					// something has been found: a candidate whose bound is clearly below it cannot win, nor tie
					if (ovb != 0 && bound_of(ovb) < best - 1e-6)
						ovb = 0;
					if (ova != 0 && bound_of(ova) < best - 1e-6) {
						ova = ovb;
						ovb = 0;
					}
				}
				if (ballot(ova != 0) == 0)
					continue;
				double pra, prb;
				if (inside)
					score_two_overlaps_serial<ALGO, true>(ltab, sf, sr, flen, rlen, ova, ovb, nrb, nn, pra, prb);
				else
					score_two_overlaps_serial<ALGO, false>(ltab, sf, sr, flen, rlen, ova, ovb, nrb, nn, pra, prb);
				// first maximum of these 128: larger score wins, equal scores -> smaller overlap
				double pr = ova != 0 ? pra : -__builtin_inf();
				int bov = ova != 0 ? ova : 0x7FFFFFFF;
				if (ovb != 0 && prb > pr) {	// (its own second candidate is the longer one)
					pr = prb;
					bov = ovb;
				}
				for (int o = 32; o > 0; o >>= 1) {
					const double opr = __shfl_xor(pr, o);
					const int oov = __shfl_xor(bov, o);
					if (opr > pr || (opr == pr && oov < bov)) {
						pr = opr;
						bov = oov;
					}
				}
				pr = uni_f64(pr);
				bov = uni(bov);
				if (pr > best) {	// groups are in ascending overlap order, strict > keeps the earlier one
					best = pr;
					bo = bov;
				}
			}
			examined = total;
			wave_sync();
		} else if (!counting_algo<ALGO>()) {
			// Few candidates.  simple_bayesian scores are exact integers-in-a-formula.  pear / rdp_mle
			// scores are ordered f64 chains, ~150 dependent adds each: the winner is DECIDED from
			// any-order sums (error < 1e-9) whenever every comparison is clearer than 1e-6, and its exact
			// ordered sum is left to the group finish as a fourth chain over the parked terms.  Any
			// comparison closer than that falls back to summing every candidate in reference order here.
			bool ambiguous = EXACT && P.score_scratch == nullptr;
			if (!ambiguous) {
				int last_ov = -1;
				u64 words = ballot(mybits != 0);
				while (words != 0) {
					const int w = __builtin_ctzll(words);
					words &= words - 1;
					unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
					while (m) {
						const int b = __builtin_ctz(m);
						m &= m - 1;
						const int ov = w * 32 + b + minov;
						score_terms<ALGO>(P, ltab, L, flen, rlen, ov, lane);
						const double pr = sum_terms_fast(L, ov, lane);
						if (pr > best + 1e-6) {
							best = pr;
							bo = ov;
						} else if (!(pr < best - 1e-6)) {
							ambiguous = true;
						}
						last_ov = ov;
						examined++;
					}
				}
				if (!ambiguous && bo >= 0) {
					if (EXACT) {
						if (last_ov != bo)
							score_terms<ALGO>(P, ltab, L, flen, rlen, bo, lane);
						score_parked = bo;	// stored to the scratch row once the prefetch registers are rotated in
					} else {
						// fast sums: `best` (any-order, |error| < 1e-9) is what gets reported; the one
						// decision that still depends on it is min_overlapbits, re-done exactly if close
						bool redo = post && P.exact_sums != 0;	// `-a` has no group finish: exact sums are made here
						for (int mi = 0; mi < P.n_modules; mi++) {
							const PamdModule md = P.modules[mi];
							redo |= md.kind == 2 && fabs(md.nats - best) <= 1e-6;
						}
						if (redo) {
							score_terms<ALGO>(P, ltab, L, flen, rlen, bo, lane);
							best = sum_terms_exact(L, bo, lane);
						}
					}
				}
			}
			if (ambiguous) {
				best = P.qual_nn * (double) (flen + rlen);
				bo = -1;
				examined = 0;
				u64 words = ballot(mybits != 0);	// non-empty 32-candidate words, ascending
				while (words != 0) {
					const int w = __builtin_ctzll(words);
					words &= words - 1;
					unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mybits, w);
					while (m) {
						const int b = __builtin_ctz(m);
						m &= m - 1;
						const int ov = w * 32 + b + minov;
						const double pr = score_overlap<ALGO>(P, ltab, L, flen, rlen, ov, lane);
						if (pr > best) {	// strict: ascending scan keeps the smallest overlap on ties
							best = pr;
							bo = ov;
						}
						examined++;
					}
				}
			}
		}
		rec.examined = examined;
		if ((u64) examined == maxov64 - (u64) minov + 1ull)	// assembler.c:135, unsigned wrap included
			rec.slow = 1;
		if (bo < 0) {
			status = PAMD_ST_NOALGN;
			break;
		}

		// Rotate the prefetched registers in BEFORE this pair's stores are issued: vmcnt counts loads
		// and stores in one in-order queue, so waiting for the (long finished) loads here is free,
		// while waiting for them at the top of the next iteration would also wait for the stores.
		cur = nxt;
		settle_carry<BPL>(cur);
		rotated = true;

		if (score_parked > 0) {
			// park the winner's terms for the finish: 16-byte stores into this wave's scratch row
			double *dst = P.score_scratch + ((size_t) wave_gid * group_size<ALGO>() + (size_t) slot_idx) * SCORE_STRIDE;
			for (int j0 = 0; j0 < score_parked; j0 += 128) {
				const int j = j0 + lane * 2;
				if (j < score_parked)
					reinterpret_cast<double2 *>(dst)[j >> 1] = reinterpret_cast<const double2 *>(L.ps)[j >> 1];
			}
			rec.score_n = score_parked;
			wave_sync();
		}
		PH_MARK(4);
		// ---- merge (assembler.c:146-247), written straight to the pair's output row ----
		const int len = flen - fo - bo + rlen - ro + 1;
		if (len <= 0 || len > 2 * PAMD_MAX_LEN) {
			status = PAMD_ST_NOALGN;
			break;
		}
		const int seqlen = len - 1;
		const int df = flen - fo - bo;
		const int dr = rlen - ro - bo;
		const int A = df > 0 ? df : 0;		// VEEZ(df): forward-only bases
		const int wdf = df < 0 ? df : 0;	// WEDGEZ(df)
		const int wdr = dr < 0 ? dr : 0;
		const int novl = bo + wdf + wdr;	// overlap bases
		const int Bn = dr > 0 ? dr : 0;		// reverse-only bases; they start at index df + bo = A + novl
		rec.seqlen = seqlen;
		rec.A = A;
		rec.novl = novl;
		rec.len = len;
		rec.bo = bo;
		rec.best = best;

		// the row: fixed stride, or the next piece of this wave's arena chunk (rows of pairs that are
		// rejected later simply stay behind as holes)
		const u64 need = ((u64) seqlen + 15ull) & ~15ull;
		u64 off;
		if (P.seq_stride != 0) {
			off = pair * P.seq_stride;
		} else {
			if (arena.cur + need > arena.end) {
				u64 got = 0;
				if (lane == 0)
					got = atomicAdd(P.seq_cursor, P.arena_chunk);
				union {
					u64 v;
					int i[2];
				} x;
				x.v = got;
				x.i[0] = uni(x.i[0]);
				x.i[1] = uni(x.i[1]);
				arena.cur = x.v;
				arena.end = x.v + P.arena_chunk;
			}
			off = arena.cur;
			arena.cur += need;
			if (off + need > P.seq_capacity)
				off = ~0ull;
		}
		rec.off = off;
		const bool store = off != ~0ull;
		uint8_t *row_nt = P.seq_nt + (store ? off : 0);
		double *row_p = P.seq_p + (store ? off : 0);
		constexpr bool want_phred = PHRED;	// a min_phred module is loaded (separate kernel instantiation)
		int mism = 0, degen = 0, phmin = 127;
		double qacc = 0.0;	// fast sums: this lane's share of the quality sum, any order
		bool low_p = false;
		double pmin = 0.0;	// every log-p is <= 0
		const double qs0 = want_phred ? ltab[PAMD_LT_SCORE] : 0.0;
		// The merge is branch-free: a lane past the end of its range reads staged bytes it ignores, takes its log-p
		// from a table slot that holds 0.0 (PAMD_LT_ZERO) and writes where a later range writes again or, past the
		// row's end, into the 64 entries of slack each staging array has.  Two rounds of 64 bases are in flight
		// at a time, written out in the order reads, table reads, writes: LDS operations complete in order and the
		// compiler keeps a read behind an earlier write it cannot tell apart, so this order is what lets one wait
		// cover both rounds' round trips (the wave is parked on those waits for half its time).
		// The log-p goes straight to the pair's row in HBM (8 bytes per lane, 512 contiguous bytes per round) unless
		// the row is still to be cut (`-a`), which needs it staged; the nt bytes are staged either way and leave
		// in 16-byte stores.  `in`: this lane's base exists (always true for the first round of a pair of rounds).
		auto put_base = [&](int at, unsigned nt, int qi, double q, bool in) {	// one merged base
			L.nts[at] = (uint8_t) nt;
			if (post) {
				L.ps[at] = q;
				L.pidx[at] = (uint16_t) qi;
			} else if (store && in) {
				row_p[at] = q;
			}
			if (!EXACT)
				qacc += q;
			// panda_result_phred (nt.c:126-150) is 1 for p <= qual_score[0] and non-decreasing above
			// it (checked over every breakpoint in tests/test_oracle_cpu.py), so the smallest phred of
			// the row follows from "any p at or below qual_score[0]" and the smallest p above it
			if (want_phred) {
				low_p |= q <= qs0;
				pmin = (q > qs0 && q < pmin) ? q : pmin;
			}
		};
		// bases of one read only: forward-only (assembler.c:162-173) at [0, A), reverse-only (assembler.c:231-243)
		// at [A + novl, seqlen); `entry(i)` is the staged read position of the i-th of them
		auto single_read = [&](const uint16_t *S, int n, int at0, auto entry) {
			int j0 = 0;
			for (; j0 + 64 < n; j0 += 128) {
				const int ia = j0 + lane, ib = ia + 64;
				const bool inb = ib < n;
				const unsigned ea = S[entry(ia)], eb = S[inb ? entry(ib) : 0];
				const int qia = PAMD_LT_SCORE + clampq(qual_of(ea)), qib = inb ? PAMD_LT_SCORE + clampq(qual_of(eb)) : PAMD_LT_ZERO;
				const double qa = ltab[qia], qb = ltab[qib];
				put_base(at0 + ia, ea & 0xFFu, qia, qa, true);
				put_base(at0 + ib, eb & 0xFFu, qib, qb, inb);
				degen += __popcll(ballot(__popc(ea & 0xFFu) != 1)) + __popcll(ballot(inb) & ballot(__popc(eb & 0xFFu) != 1));	// PANDA_NT_IS_DEGN
			}
			if (j0 < n) {
				const int i = j0 + lane;
				const bool in = i < n;
				const unsigned e = S[in ? entry(i) : 0];
				const int qi = in ? PAMD_LT_SCORE + clampq(qual_of(e)) : PAMD_LT_ZERO;
				const double q = ltab[qi];
				put_base(at0 + i, e & 0xFFu, qi, q, in);
				degen += __popcll(ballot(in) & ballot(__popc(e & 0xFFu) != 1));
			}
		};
		single_read(L.F, A, 0, [&](int j) { return j + fo; });
		// overlapping bases (assembler.c:181-228)
		auto overlap_index = [&](unsigned ef, unsigned er, int fi, int ri, bool in, unsigned &nt) -> int {
			const unsigned a = ef & 0xFFu, b = er & 0xFFu;
			const int fq = qual_of(ef), rq = qual_of(er);
			const bool match = (a & b) != 0;
			nt = match ? (a & b) : (fq < rq ? b : a);	// assembler.c:212-220
			const bool fmask = fi >= uf, rmask = ri >= ur;	// B-cliff
			// one table read, its index chosen without a branch (slot 47 of the score block holds qual_nn)
			const int cf = clampq(fq), cr = clampq(rq);
			int qi;
			if (ALGO == PAMD_ALGO_EA_UTIL) {	// algo_ea_util.c:58-67
				qi = PAMD_LT_SCORE + (fq > rq ? cf : cr);
			} else if (ALGO == PAMD_ALGO_FLASH) {	// algo_flash.c:63-80
				int sc = cf - cr;
				sc = sc < 0 ? -sc : sc;
				sc = sc < 2 ? 2 : sc;
				qi = PAMD_LT_SCORE + (match ? (fq > rq ? cf : cr) : sc);
			} else if (ALGO == PAMD_ALGO_RDP_MLE) {	// algo_rdp_mle.c:29-41
				qi = match ? PAMD_LT_SCORE + clampq(fq >= rq ? fq : rq) : PAMD_LT_C + cf * PAMD_NQ + cr;
			} else {	// algo_simple_bayes.c:68-75, algo_pear.c:61-68, algo_uparse.c:68-75, algo_stitch.c:59-67
				qi = (match ? PAMD_LT_A : PAMD_LT_B) + cf * PAMD_NQ + cr;
			}
			qi = rmask ? PAMD_LT_SCORE + cf : qi;	// B-cliff (assembler.c:192-207)
			qi = fmask ? PAMD_LT_SCORE + (rmask ? 47 : cr) : qi;
			return in ? qi : PAMD_LT_ZERO;
		};
		{
			int i0 = 0;
			for (; i0 + 64 < novl; i0 += 128) {
				const int ia = i0 + lane, ib = ia + 64;
				const bool inb = ib < novl;
				const int fia = fo + A + ia, fib = fia + 64;
				const int ria = rlen - ia - 1 + wdf, rib = inb ? ria - 64 : 0;
				const unsigned efa = L.F[fia], era = L.R[ria], efb = L.F[fib], erb = L.R[rib];
				unsigned nta, ntb;
				const int qia = overlap_index(efa, era, fia, ria, true, nta), qib = overlap_index(efb, erb, fib, rib, inb, ntb);
				const double qa = ltab[qia], qb = ltab[qib];
				put_base(A + ia, nta, qia, qa, true);
				put_base(A + ib, ntb, qib, qb, inb);
				const u64 inmask = ballot(inb);
				mism += __popcll(ballot((efa & era & 0xFFu) == 0)) + __popcll(inmask & ballot((efb & erb & 0xFFu) == 0));
				degen += __popcll(ballot(__popc(nta) != 1)) + __popcll(inmask & ballot(__popc(ntb) != 1));
			}
			if (i0 < novl) {
				const int i = i0 + lane;
				const bool in = i < novl;
				const int fi = fo + A + i;
				const int ri = in ? rlen - i - 1 + wdf : 0;
				const unsigned ef = L.F[fi], er = L.R[ri];
				unsigned nt;
				const int qi = overlap_index(ef, er, fi, ri, in, nt);
				const double q = ltab[qi];
				put_base(A + i, nt, qi, q, in);
				const u64 inmask = ballot(in);
				mism += __popcll(inmask & ballot((ef & er & 0xFFu) == 0));
				degen += __popcll(inmask & ballot(__popc(nt) != 1));
			}
		}
		single_read(L.R, Bn, A + novl, [&](int i) { return rlen - bo - i - 1; });
		wave_sync();
		int out_len = seqlen;	// bases the row keeps
		if (post) {
			// ---- `-a`: find the primers in the merged sequence and cut them off it (assembler.c:300-333) ----
			if (!EXACT)
				quality_fast = fast_quality(L.ps, qacc, P.threshold, len, seqlen, A, novl, P.exact_sums != 0, lane);	// needs the whole row: before it is cut
			int pfo = P.ftrim, pro = P.rtrim;
			if (P.fprimer_len > 0) {
				const int r = primer_offset_result(P, L, seqlen, lprimers, P.fprimer_len, false, lane);
				if (r == 0) {
					status = PAMD_ST_NOFP;
					break;
				}
				pfo = r - 1;
			}
			rec.fo = pfo;
			if (P.rprimer_len > 0) {
				const int r = primer_offset_result(P, L, seqlen, lprimers + 456, P.rprimer_len, true, lane);
				if (r == 0) {
					status = PAMD_ST_NORP;
					break;
				}
				pro = r - 1;
			}
			rec.ro = pro;
			if (seqlen <= pfo + pro) {	// assembler.c:323-327: counted as a missing forward primer
				status = PAMD_ST_NOFP;
				break;
			}
			out_len = seqlen - pfo - pro;
			rec.seqlen = out_len;
			// move the kept bases to the front of the staged row, 64 at a time, reads before writes
			if (pfo > 0) {
				for (int j0 = 0; j0 < out_len; j0 += 64) {
					const int j = j0 + lane;
					unsigned nt = 0;
					double q = 0.0;
					if (j < out_len) {
						nt = L.nts[j + pfo];
						q = L.ps[j + pfo];
					}
					wave_sync();
					if (j < out_len) {
						L.nts[j] = (uint8_t) nt;
						L.ps[j] = q;
					}
				}
			}
			wave_sync();
			if (want_phred) {	// the smallest phred of the KEPT bases
				low_p = false;
				pmin = 0.0;
				for (int j = lane; j < out_len; j += 64) {
					const double q = L.ps[j];
					low_p |= q <= qs0;
					pmin = (q > qs0 && q < pmin) ? q : pmin;
				}
			}
		}
		// the row leaves LDS in full 16-byte stores (16 nt or 2 log-p per lane); rows are 16-base aligned
		if (store) {
			// the LDS reads are unconditional and come first, two rounds of log-p at a time, so that one wait covers
			// them all; only the stores are predicated (a read past the row's end returns staged bytes nobody keeps)
			const uint4 nt16 = reinterpret_cast<const uint4 *>(L.nts)[lane];
			const double2 *src = reinterpret_cast<const double2 *>(L.ps);
			double2 *dst = reinterpret_cast<double2 *>(row_p);
			if (lane * 16 < out_len)
				reinterpret_cast<uint4 *>(row_nt)[lane] = nt16;
			for (int j0 = 0; post && j0 < out_len; j0 += 256) {
				const int ja = j0 + lane * 2, jb = ja + 128;
				const double2 pa = src[ja >> 1], pb = src[jb >> 1];
				if (ja < out_len)
					dst[ja >> 1] = pa;
				if (jb < out_len)
					dst[jb >> 1] = pb;
			}
		}
		if (want_phred) {
			for (int o = 32; o > 0; o >>= 1) {
				const double other = __shfl_xor(pmin, o);
				pmin = other < pmin ? other : pmin;
			}
			phmin = result_phred(ltab + PAMD_LT_SCORE, uni_f64(pmin));
			if (ballot(low_p) != 0 && phmin > 1)
				phmin = 1;
			rec.phmin = out_len > 0 ? phmin : 127;	// an empty row passes any threshold
		}
		rec.mism = mism;
		rec.degen = degen;
		if (!EXACT && !post) {
			quality_fast = quality_sum(qacc);	// divided by len in finish64() ...
			// ... unless the quotient would land within 1e-9 of the LOWQ threshold (tested without dividing):
			// then the three sums are redone here, in reference order, from the staged row
			// (a row the arena had no room for was not written: the launch reports the overflow and its results are void)
			if (store && fabs(quality_fast - P.threshold * (double) len) <= 1e-9 * (double) len) {
				quality_fast = fast_quality(row_p, qacc, P.threshold, len, seqlen, A, novl, true, lane);
				quality_final_flag = true;
			}
		}
	} while (0);
	if (!rotated) {
		cur = nxt;
		settle_carry<BPL>(cur);
	}
	rec.st = status;
	PH_MARK_END(7);
	if (EXACT) {
		if (lane == 0)
			*slot_rec = rec;	// park it; finish_group() does the rest
		return;
	}
	// fast sums: the pair's scalars wait in lane `slot_idx` of the wave's pending registers (finish64)
	park64(pd, slot_idx, rec, status, quality_fast, quality_final_flag);
}

template <int ALGO, bool PHRED, int BPL, bool EXACT, bool POST>
__global__ __launch_bounds__(1024) void pamd_assemble_kernel(const PamdKernelParams P) {
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	double *ltab = reinterpret_cast<double *>(lds);
	unsigned *lctr = reinterpret_cast<unsigned *>(lds + P.lds_ctr_off);
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = uni(tid >> 6);
	const int nwaves = blockDim.x >> 6;

	for (unsigned i = tid; i < P.lds_table_doubles; i += blockDim.x)
		ltab[i] = P.tables[i];
	for (unsigned i = tid; i < PAMD_NUM_COUNTERS; i += blockDim.x)
		lctr[i] = 0;
	uint8_t *lprimers = lds + P.lds_primer_off;	// both primers, staged once per workgroup
	if (P.fprimer_len > 0 || P.rprimer_len > 0)
		for (unsigned i = tid; i < 912; i += blockDim.x)
			lprimers[i] = P.primers[i];
	__syncthreads();

	unsigned char *wb = lds + P.lds_wave_off + (size_t) wave * P.lds_wave_bytes;
	WaveLds L;
	L.F = reinterpret_cast<uint16_t *>(wb + P.w_fwd);
	L.R = reinterpret_cast<uint16_t *>(wb + P.w_rev);
	L.fc = reinterpret_cast<uint32_t *>(wb + P.w_fc);
	L.rc = reinterpret_cast<uint32_t *>(wb + P.w_rc);
	L.fn = reinterpret_cast<uint32_t *>(wb + P.w_fn);
	L.rn = reinterpret_cast<uint32_t *>(wb + P.w_rn);
	L.bits = reinterpret_cast<uint32_t *>(wb + P.w_bits);
	L.nts = wb + P.w_nt;
	L.ps = reinterpret_cast<double *>(wb + P.w_p);
	L.hash = reinterpret_cast<uint32_t *>(wb + P.w_p);
	L.pidx = reinterpret_cast<uint16_t *>(wb + P.w_pidx);
	PendRec *pend = reinterpret_cast<PendRec *>(wb + P.w_pend);

	WaveArena arena;
	arena.cur = 0;
	arena.end = 0;
	const u64 total_waves = (u64) gridDim.x * (u64) nwaves;
	const unsigned wave_gid = blockIdx.x * (unsigned) nwaves + (unsigned) wave;
#ifdef PAMD_PHASE_TIMING
	u64 ph_acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	const u64 ph_begin = __builtin_amdgcn_s_memtime();
#endif
	// Persistent loop with a one-pair software prefetch: the four loads of pair k+1 (two lengths,
	// two read rows) are issued before pair k is assembled and are first touched when k is done.
	// No branch surrounds them (indices are clamped instead), so the waits stay counted.
	// `pair` walks work items: pair indices, or -- second launch of a two-kernel run -- positions in the list of
	// pairs the lean kernel deferred (P.pair_list, *P.pair_count of them); `pid` is the pair index itself.
	const uint32_t *const plist = P.pair_list;
	const u64 n_work = plist != nullptr ? (u64) *P.pair_count : P.n_pairs;
	// Work is handed out from a device counter, not dealt in fixed shares: a pair that scans every overlap costs twenty to a
	// hundred times an ordinary one, and a fixed share per wave leaves the launch waiting for its unluckiest wave (46 % over
	// the mean on the lean kernel's list, 58 mostly such pairs per wave; 5 % on a whole 50 M-pair batch).  A list goes out
	// pair by pair, a whole batch sixteen pairs at a time (one same-address atomic per pair would be the bottleneck).
	const bool queued = P.pair_cursor != nullptr;
	const unsigned take_n = plist != nullptr ? 1u : 16u;
	u64 took = 0;
	unsigned took_left = 0;
	auto take = [&]() -> u64 {
		if (took_left == 0) {
			unsigned got = 0;
			if (lane == 0)
				got = atomicAdd(P.pair_cursor, take_n);
			took = (u64) (unsigned) uni((int) got);
			took_left = take_n;
		}
		took_left--;
		return took++;
	};
	u64 pair = queued ? take() : (u64) blockIdx.x * (u64) nwaves + (u64) wave;
	const u64 last = n_work != 0 ? n_work - 1 : 0;	// an empty list: the prologue reads pair 0 and the loop does not run
	u64 q = pair < n_work ? pair : last;
	u64 pid = (plist != nullptr && n_work != 0) ? (u64) plist[q] : q;
	Carry cur;
	cur.flen_v = P.flen[pid];
	cur.rlen_v = P.rlen[pid];
	const uint32_t *const found = P.primer_offsets != nullptr ? P.primer_offsets : reinterpret_cast<const uint32_t *>(P.tables);	// (else any readable word)
	cur.primers_v = found[P.primer_offsets != nullptr ? pid : 0];
	cur.f = load_read<BPL>(P.fwd + pid * P.fwd_row_bytes, P.fwd_row_bytes, lane);
	cur.r = load_read<BPL>(P.rev + pid * P.rev_row_bytes, P.rev_row_bytes, lane);
	// settle the prologue loads before the loop so that, at the loop header, the carried registers
	// are known-ready on both incoming edges
	settle_carry<BPL>(cur);
	int npend = 0;
	Pend64 pd = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	u64 group_first = pair;	// fast sums: the pair parked in lane 0 of the pending registers
	while (pair < n_work) {
		const u64 next = queued ? take() : pair + total_waves;
		q = next < n_work ? next : last;
		const u64 npid = plist != nullptr ? (u64) plist[q] : q;
		if (!EXACT && queued)	// which pair this is, parked with its scalars (exact sums park the pair's index with its record)
			pd.pid = put_lane(pd.pid, (unsigned) pid, lane == npend);
		Carry nxt;
		nxt.flen_v = P.flen[npid];
		nxt.rlen_v = P.rlen[npid];
		nxt.primers_v = found[P.primer_offsets != nullptr ? npid : 0];
		nxt.f = load_read<BPL>(P.fwd + npid * P.fwd_row_bytes, P.fwd_row_bytes, lane);
		nxt.r = load_read<BPL>(P.rev + npid * P.rev_row_bytes, P.rev_row_bytes, lane);
		const int flen = uni(cur.flen_v), rlen = uni(cur.rlen_v);
		// The primers-after variants run out of registers; what the compiler would otherwise keep across pairs in scratch
		// (every constant derived from the lane number) is cheaper recomputed, because a scratch reload waits in the same
		// in-order queue as the next pair's rows.
		int lane_now = lane;
		if constexpr (POST)
			asm volatile("" : "+v"(lane_now));
#ifdef PAMD_PHASE_TIMING
		assemble_pair<ALGO, PHRED, BPL, EXACT, POST>(P, ltab, lprimers, lctr, L, arena, &pend[EXACT ? npend : 0], pd, npend, wave_gid, pid, lane_now, flen, rlen, cur, nxt, ph_acc);
#else
		assemble_pair<ALGO, PHRED, BPL, EXACT, POST>(P, ltab, lprimers, lctr, L, arena, &pend[EXACT ? npend : 0], pd, npend, wave_gid, pid, lane_now, flen, rlen, cur, nxt);
#endif
		pid = npid;
		npend++;
		if (!EXACT && npend == 64) {
			if (queued)	// (the wave's own record says which pair is behind which parked record)
				finish64<true, false, false, true>(P, lctr, pd, npend, 0, 0, lane);
			else if (plist != nullptr)
				finish64<true, false>(P, lctr, pd, npend, group_first, total_waves, lane);
			else
				finish64<false, false>(P, lctr, pd, npend, group_first, total_waves, lane);
			npend = 0;
			group_first = next;
		}
		if (EXACT && npend == group_size<ALGO>()) {
#ifdef PAMD_PHASE_TIMING
			const u64 t0 = __builtin_amdgcn_s_memtime();
#endif
			finish_group<ALGO>(P, ltab, lctr, pend, npend, wave_gid, lane);
#ifdef PAMD_PHASE_TIMING
			ph_acc[8] += __builtin_amdgcn_s_memtime() - t0;
#endif
			npend = 0;
		}
		pair = next;
	}
	if (EXACT && npend > 0)
		finish_group<ALGO>(P, ltab, lctr, pend, npend, wave_gid, lane);
	if (!EXACT && npend > 0) {
		if (queued)
			finish64<true, false, false, true>(P, lctr, pd, npend, 0, 0, lane);
		else if (plist != nullptr)
			finish64<true, false>(P, lctr, pd, npend, group_first, total_waves, lane);
		else
			finish64<false, false>(P, lctr, pd, npend, group_first, total_waves, lane);
	}
#ifdef PAMD_PHASE_TIMING
	ph_acc[9] = __builtin_amdgcn_s_memtime() - ph_begin;
	if (lane == 0 && P.phase_cycles != nullptr)
		for (int k = 0; k < 10; k++)
			atomicAdd(&P.phase_cycles[k], ph_acc[k]);
#endif
	__syncthreads();
	if (P.counters != nullptr) {
		for (unsigned i = tid; i < PAMD_NUM_COUNTERS; i += blockDim.x) {
			const unsigned v = lctr[i];
			if (v != 0)
				atomicAdd(&P.counters[i], (unsigned long long) v);
		}
	}
}

// ---- synthetic pairs (pandaseq_amd/synth.py is the host twin) ---------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {
	x ^= x >> 30;
	x *= 0xBF58476D1CE4E5B9ull;
	x ^= x >> 27;
	x *= 0x94D049BB133111EBull;
	x ^= x >> 31;
	return x;
}

__device__ __forceinline__ u64 draw(u64 key, u64 slot) {
	return mix64(key + slot * 0xD1B54A32D192ED03ull);
}

__global__ __launch_bounds__(256) void pamd_synth_kernel(const PamdSynthParams S) {
	const int lane = threadIdx.x & 63;
	const u64 wave = ((u64) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const u64 nw = ((u64) gridDim.x * blockDim.x) >> 6;
	const int L = (int) S.spec.read_len;
	const int qrepl[6] = { 2, 8, 15, 22, 27, 33 };
	for (u64 idx = wave; idx < S.n; idx += nw) {
		const u64 pair = S.first + idx;
		const u64 key = mix64(S.spec.seed + pair * 0x9E3779B97F4A7C15ull);
		const unsigned ku = (unsigned) (draw(key, 0) % 100ull);
		const int kind = ku < 97 ? 0 : (ku < 98 ? 1 : 2);
		const int tlen = (int) (S.spec.tmin + draw(key, 1) % (u64) (S.spec.tmax - S.spec.tmin + 1));
		const int unit = (int) (2 + draw(key, 2) % 11ull);
		int prefix = 0;
		if (S.spec.fprimer_len > 0 && kind == 0) {
			const u64 pu = draw(key, 5);
			if (pu % 10ull == 0)
				prefix = 1 + (int) ((pu >> 32) % 3ull);
		}
		for (int r = 0; r < 2; r++) {
			const u64 ut = draw(key, 3 + r);
			const int tail = (ut % 20ull == 0) ? 1 + (int) ((ut >> 32) % 30ull) : 0;
			uint8_t *row = (r == 0 ? S.fwd : S.rev) + idx * (u64) S.stride * 2ull;
			for (int pos = lane; pos < (int) S.stride; pos += 64) {
				unsigned nt = 0;
				int q = 0;
				if (pos < L) {
					int b;
					if (r == 1 && kind == 2) {
						b = (int) (draw(key, 64 + 1024 + pos) & 3ull);
					} else {
						int t = r == 0 ? pos : tlen - 1 - pos;
						if (kind == 1)
							t = t % unit;
						b = (int) (draw(key, 64 + t) & 3ull);
					}
					if (S.spec.fprimer_len > 0 && kind == 0) {
						if (r == 0) {
							const int off = pos - prefix;
							if (off >= 0 && off < (int) S.spec.fprimer_len)
								b = S.spec.fprimer[off];
						} else if (pos < (int) S.spec.rprimer_len) {
							b = 3 - S.spec.rprimer[pos];
						}
					}
					const u64 slot = 4096ull + (u64) r * 4096ull + (u64) pos * 4ull;
					const u64 u0 = draw(key, slot), u1 = draw(key, slot + 1), u2 = draw(key, slot + 2);
					if (S.spec.novaseq) {
						const int v = (int) (u0 % 100ull);
						q = v < 85 ? 37 : (v < 93 ? 23 : (v < 98 ? 12 : 2));
					} else {
						q = S.spec.qhi - (pos * 12) / L;
						if (u0 % 10ull == 0)
							q = qrepl[(u0 >> 32) % 6ull];
					}
					if (pos >= L - tail)
						q = 2;
					if ((unsigned) (u1 & 0xFFFFFFFFull) < S.thr[q])
						b = (b + 1 + (int) ((u1 >> 32) % 3ull)) & 3;
					nt = 1u << b;
					if (q == 2 && (u2 % 10ull) < 3)
						nt = NT_N;
				}
				row[pos * 2] = (uint8_t) nt;
				row[pos * 2 + 1] = (uint8_t) (int8_t) q;
			}
		}
		if (lane == 0) {
			S.flen[idx] = (uint16_t) L;
			S.rlen[idx] = (uint16_t) L;
		}
	}
}

}	// namespace

int pamd_launch_assemble(const PamdKernelParams &p, int workgroups, int waves_per_wg, size_t lds_bytes, void *stream) {
	const dim3 grid((unsigned) workgroups), block((unsigned) waves_per_wg * 64u);
	hipStream_t st = static_cast<hipStream_t>(stream);
	void (*kern)(const PamdKernelParams) = nullptr;
	const bool ph = p.has_min_phred != 0;
	const bool narrow = p.smax <= 256;	// 4 bases per lane while staging
// `-a` always runs the instantiation without the deferred group finish (the finish re-reads rows by the
	// ranges of the untrimmed merge); with exact sums asked for it makes them in the wave instead
#define PAMD_PICK3(A, PH, B) (p.post_primers ? pamd_assemble_kernel<A, PH, B, false, true> : (p.exact_sums ? pamd_assemble_kernel<A, PH, B, true, false> : pamd_assemble_kernel<A, PH, B, false, false>))
#define PAMD_PICK(A) (ph ? (narrow ? PAMD_PICK3(A, true, 4) : PAMD_PICK3(A, true, 8)) : (narrow ? PAMD_PICK3(A, false, 4) : PAMD_PICK3(A, false, 8)))
	switch (p.algo) {
	case PAMD_ALGO_SIMPLE_BAYES:
		kern = PAMD_PICK(PAMD_ALGO_SIMPLE_BAYES);
		break;
	case PAMD_ALGO_PEAR:
		kern = PAMD_PICK(PAMD_ALGO_PEAR);
		break;
	case PAMD_ALGO_RDP_MLE:
		kern = PAMD_PICK(PAMD_ALGO_RDP_MLE);
		break;
	case PAMD_ALGO_UPARSE:
		kern = PAMD_PICK(PAMD_ALGO_UPARSE);
		break;
	case PAMD_ALGO_FLASH:
		kern = PAMD_PICK(PAMD_ALGO_FLASH);
		break;
	case PAMD_ALGO_EA_UTIL:
		kern = PAMD_PICK(PAMD_ALGO_EA_UTIL);
		break;
	case PAMD_ALGO_STITCH:
		kern = PAMD_PICK(PAMD_ALGO_STITCH);
		break;
	default:
		pamd_set_error("unknown algorithm %d", p.algo);
		return PAMD_EINVAL;
	}
#undef PAMD_PICK
#undef PAMD_PICK3
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
	if (e != hipSuccess) {
		pamd_set_error("hipFuncSetAttribute(%zu bytes LDS): %s", lds_bytes, hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
	e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("assemble kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}

int pamd_launch_synth(const PamdSynthParams &p, void *stream) {
	hipStream_t st = static_cast<hipStream_t>(stream);
	u64 waves = p.n < 65536 ? p.n : 65536;
	if (waves == 0)
		return PAMD_OK;
	const unsigned blocks = (unsigned) ((waves + 3) / 4);
	hipLaunchKernelGGL(pamd_synth_kernel, dim3(blocks), dim3(256), 0, st, p);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("synth kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
