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

namespace {

// k-mer table slots per wave: a power of two >= 512 (> 442 = most forward k-mers a 450-base read
// can emit), as many as fit in the log-p staging area it shares (PamdKernelParams::hash_slots)
constexpr unsigned HASH_EMPTY = 0xFFFFFFFFu;
constexpr unsigned NT_N = 15;
constexpr unsigned PLANE_STRIDE = 80;	// one zero dword in front, 64 bytes of bits (512 positions), zero tail
constexpr unsigned PLANE_BYTES = 4 * PLANE_STRIDE;
constexpr unsigned long long ARENA_CHUNK = 16384;	// bases a wave reserves from the output arena at a time

typedef unsigned long long u64;

// Optional per-phase cycle accounting (build with -DPAMD_PHASE_TIMING; see tools/phase_timing.py).
// Costs a few % itself, so it is never compiled into the shipped library.
#ifdef PAMD_PHASE_TIMING
#define PH_DECL u64 ph_t0 = __builtin_amdgcn_s_memtime(), ph_t1
#define PH_MARK(k)                                   \
	do {                                         \
		ph_t1 = __builtin_amdgcn_s_memtime(); \
		ph_acc[k] += ph_t1 - ph_t0;          \
		ph_t0 = ph_t1;                       \
	} while (0)
#else
#define PH_DECL
#define PH_MARK(k)
#endif

__device__ __forceinline__ void wave_sync() {
	// Lanes of one wavefront exchange data through LDS.  The LDS unit executes one wave's DS
	// operations in issue order, so ordering the compiler is all that is needed.
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int v) {
	return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ double uni_f64(double v) {
	union {
		double d;
		int i[2];
	} x;
	x.d = v;
	x.i[0] = __builtin_amdgcn_readfirstlane(x.i[0]);
	x.i[1] = __builtin_amdgcn_readfirstlane(x.i[1]);
	return x.d;
}

__device__ __forceinline__ double lane_f64(double v, int src) {
	union {
		double d;
		int i[2];
	} x;
	x.d = v;
	x.i[0] = __builtin_amdgcn_readlane(x.i[0], src);
	x.i[1] = __builtin_amdgcn_readlane(x.i[1], src);
	return x.d;
}

__device__ __forceinline__ int clampq(int q) {	// PHREDCLAMP, prob.h:23
	return q > 46 ? 46 : (q < 0 ? 0 : q);
}

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
	uint32_t *fn;	// forward N bitmap
	uint32_t *rn;	// reverse N bitmap
	uint8_t *fpl;	// forward one-hot bit planes A,C,G,T: 4 x PLANE_STRIDE bytes, bit p = base p has that bit
	uint8_t *rpl;	// reverse one-hot bit planes (natural read order)
	uint32_t *bits;	// candidate overlaps (BITS_INIT, assembler.c:41,84)
	uint8_t *nts;	// merged nt staging
	double *ps;	// merged log-p staging; the k-mer table lives here while seeding
	uint32_t *hash;
};

// ---- load one read, stage it, derive its packed k-mer inputs --------------------------------
// Returns the B-cliff length: read length minus the trailing run of raw quality 2 (assembler.c:176-177).
// Branch-free so the compiler can count it in s_waitcnt vmcnt(N): every lane loads 16 B of the row,
// lanes past the row end re-read its last 16 bytes (rows are whole multiples of 16 B).
__device__ __forceinline__ uint4 load_read(const uint8_t *row, unsigned row_bytes, int lane) {
	unsigned off = (unsigned) lane * 16u;
	off = off < row_bytes ? off : row_bytes - 16u;
	return *reinterpret_cast<const uint4 *>(row + off);
}

template <bool FORWARD>
__device__ __forceinline__ int stage_read(const uint4 v, int len, int lane, uint16_t *dst, uint32_t *codes, uint32_t *nmap, uint8_t *planes) {
	if (lane * 8 < len)
		reinterpret_cast<uint4 *>(dst)[lane] = v;
	const unsigned w[4] = { v.x, v.y, v.z, v.w };
	// Branch-free on purpose (straight-line bit arithmetic; ?: chains here compile to exec-mask
	// branches per base).  nt is 0..15 in the device domain.
	//   2-bit code (misc.h:41: T=3 G=2 C=1, anything else 0) = two bits of a constant indexed by nt
	//   N          = (nt + 1) >> 4
	//   qual == 2  = bit 8 of ((qual ^ 2) - 1) taken as an unsigned byte
	//   one-hot planes: bit b of nt, gathered over the lane's 8 bases into one byte per plane
	unsigned code16 = 0, is2 = 0, pa = 0, pc = 0, pg = 0, pt = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const unsigned e = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
		const unsigned nt = e & 0xFFu;
		const unsigned c = (0x00030210u >> ((nt & 15u) * 2u)) & 3u;
		code16 |= FORWARD ? (c << (14 - 2 * k)) : (c << (2 * k));
		pa |= (nt & 1u) << k;
		pc |= ((nt >> 1) & 1u) << k;
		pg |= ((nt >> 2) & 1u) << k;
		pt |= ((nt >> 3) & 1u) << k;
		is2 |= (((((e >> 8) ^ 2u) & 0xFFu) - 1u) >> 8 & 1u) << k;
	}
	unsigned nbits = pa & pc & pg & pt;	// nt == 15
	// bases at or beyond the read length contribute nothing
	int cnt = len - lane * 8;
	cnt = cnt < 0 ? 0 : (cnt > 8 ? 8 : cnt);
	const unsigned inmask = (1u << cnt) - 1u;	// bit k set <=> base k of this lane exists
	nbits &= inmask;
	planes[4 + lane] = (uint8_t) (pa & inmask);
	planes[4 + PLANE_STRIDE + lane] = (uint8_t) (pc & inmask);
	planes[4 + 2 * PLANE_STRIDE + lane] = (uint8_t) (pg & inmask);
	planes[4 + 3 * PLANE_STRIDE + lane] = (uint8_t) (pt & inmask);
	const unsigned not2 = ~is2 & inmask;
	const unsigned cmask = FORWARD ? ~(0xFFFFu >> (2 * cnt)) & 0xFFFFu : (1u << (2 * cnt)) - 1u;
	code16 &= cmask;
	// 16 codes per u32.  Forward reads are packed big-endian inside the word so that a k-mer
	// extracted from them has its OLDEST base in the top bits like FOREACH_KMER's rolling value;
	// reverse reads are packed little-endian, which gives FOREACH_KMER_REVERSE's value directly.
	reinterpret_cast<uint16_t *>(codes)[FORWARD ? (lane ^ 1) : lane] = (uint16_t) code16;
	reinterpret_cast<uint8_t *>(nmap)[lane] = (uint8_t) nbits;
	const int mine = not2 ? lane * 8 + (31 - __builtin_clz(not2)) + 1 : 0;
	const u64 has = __ballot(not2 != 0);
	if (has == 0)
		return 0;
	const int top = 63 - __builtin_clzll(has);
	return __builtin_amdgcn_readlane(mine, top);
}

// ---- primer offset (offset.c:47-112) ---------------------------------------------------------
// Lane s owns the alignment of the primer starting at read position s and accumulates its
// log-probability in primer order x = 0..plen-1, which is the order in which offset.c:79-87 adds
// into bucket s.  That bucket is judged at index = s + plen (offset.c:72-77).
// Returns what computeoffset() returns: 0 = not found, else 1 + number of bases to clip.
__device__ int primer_offset(const PamdKernelParams &P, const double *ltab, const uint16_t *S, int len, const uint8_t *primer, int plen, int lane) {
	if (plen > len)
		return 0;
	const bool logdomain = P.primer_penalty == 0.0;
	// bestpr = exp(primerlen * threshold); with a zero penalty every candidate is exp(x) of some x
	// and exp is monotone, so we compare x (see DESIGN.md "exp in the primer scan").
	double best = logdomain ? (double) plen * P.threshold : exp((double) plen * P.threshold);
	int besti = 0;
	for (int base = 0; base + plen <= len - 1; base += 64) {
		const int s = base + lane;
		const int index = s + plen;
		if (index <= len - 1) {
			double acc = 0.0;
			for (int x = 0; x < plen; x++) {
				const unsigned pn = primer[x];
				if (pn != NT_N) {
					const unsigned e = S[s + x];
					const int q = clampq(qual_of(e));
					acc += ((nt_of(e) & pn) != 0) ? ltab[PAMD_LT_SCORE + q] : ltab[PAMD_LT_ERR + q];
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
	// first-maximum across lanes: larger value wins, equal values -> smaller index
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

// ---- simple_bayesian overlap_probability for up to 4 candidates at once (algo_simple_bayes.c:33-66) ----
// The three counts are popcounts over one-hot bit planes.  Lane (row, c): row = lane >> 4 picks the
// candidate, c = lane & 15 the 32-position chunk i in [32c, 32c + 32) of the overlap walk
// (fi = flen - ov + i ascending in the forward planes, ri = rlen - 1 - i descending in the reverse
// planes, so the reverse chunk is extracted at its mirrored position and bit-reversed).
__device__ __forceinline__ unsigned plane_word(const uint8_t *plane, int start_bit) {
	// 32 bits of a plane starting at bit `start_bit` (>= -32; the dword in front of the plane is zero)
	const uint32_t *w = reinterpret_cast<const uint32_t *>(plane) + 1 + (start_bit >> 5);
	return __builtin_amdgcn_alignbit(w[1], w[0], (unsigned) start_bit & 31u);
}

__device__ __forceinline__ double score_sb_planes(const PamdKernelParams &P, const WaveLds &L, int flen, int rlen, int ov, bool active, int lane) {
	const int c = lane & 15;
	const int i0 = 32 * c;
	// positions that exist on both reads: max(0, ov - flen) <= i < min(ov, rlen)
	int lo_i = ov - flen;
	lo_i = lo_i < 0 ? 0 : lo_i;
	const int hi_i = ov < rlen ? ov : rlen;
	int lo = lo_i - i0, hi = hi_i - i0;
	lo = lo < 0 ? 0 : (lo > 32 ? 32 : lo);
	hi = hi < 0 ? 0 : (hi > 32 ? 32 : hi);
	const unsigned below_hi = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
	const unsigned below_lo = lo >= 32 ? 0xFFFFFFFFu : ((1u << lo) - 1u);
	unsigned vmask = active ? (below_hi & ~below_lo) : 0u;
	unsigned packed = 0;
	if (vmask != 0) {
		const int fs = flen - ov + i0;	// >= -31 here
		const int rs = rlen - 32 - i0;	// >= -31 here
		const unsigned fa = plane_word(L.fpl, fs), fc = plane_word(L.fpl + PLANE_STRIDE, fs);
		const unsigned fg = plane_word(L.fpl + 2 * PLANE_STRIDE, fs), ft = plane_word(L.fpl + 3 * PLANE_STRIDE, fs);
		const unsigned ra = __builtin_bitreverse32(plane_word(L.rpl, rs)), rc = __builtin_bitreverse32(plane_word(L.rpl + PLANE_STRIDE, rs));
		const unsigned rg = __builtin_bitreverse32(plane_word(L.rpl + 2 * PLANE_STRIDE, rs)), rt = __builtin_bitreverse32(plane_word(L.rpl + 3 * PLANE_STRIDE, rs));
		const unsigned match = (fa & ra) | (fc & rc) | (fg & rg) | (ft & rt);
		const unsigned isn = (fa & fc & fg & ft) | (ra & rc & rg & rt);
		const unsigned unknowns = __popc(vmask & isn);
		const unsigned matches = __popc(vmask & ~isn & match);
		const unsigned mismatches = __popc(vmask & ~isn & ~match);
		packed = unknowns | (matches << 10) | (mismatches << 20);	// each count <= 450 < 1024
	}
	// sum over the 16 lanes of the row (rotate-and-add inside the DPP row)
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x128, 0xF, 0xF, false);	// row_ror:8
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x124, 0xF, 0xF, false);	// row_ror:4
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x122, 0xF, 0xF, false);	// row_ror:2
	packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x121, 0xF, 0xF, false);	// row_ror:1
	const unsigned unknowns = packed & 1023u, matches = (packed >> 10) & 1023u, mismatches = packed >> 20;
	// size_t arithmetic inside the parenthesis (algo_simple_bayes.c:61-65), wrap included
	const u64 free_bases = (ov >= flen && ov >= rlen) ? (u64) unknowns : (u64) flen + (u64) rlen - 2ull * (u64) ov + (u64) unknowns;
	return P.qual_nn * (double) free_bases + (double) matches * P.sb_pmatch + (double) mismatches * P.sb_pmismatch;
}

// ---- overlap_probability, one candidate ------------------------------------------------------
template <int ALGO>
__device__ __forceinline__ double score_overlap(const PamdKernelParams &P, const double *ltab, const WaveLds &L, int flen, int rlen, int ov, int lane) {
	if (ALGO == PAMD_ALGO_SIMPLE_BAYES) {
		// algo_simple_bayes.c:33-66: three integer counts, then a closed form
		int unknowns = 0, matches = 0, mismatches = 0;
		for (int i0 = 0; i0 < ov; i0 += 64) {
			const int i = i0 + lane;
			const int fi = flen + i - ov;
			const int ri = rlen - i - 1;
			const bool in = i < ov && fi >= 0 && ri >= 0;
			const unsigned a = in ? (L.F[fi] & 0xFFu) : 0u;
			const unsigned b = in ? (L.R[ri] & 0xFFu) : 0u;
			const bool isn = a == NT_N || b == NT_N;
			const bool m = (a & b) != 0;
			unknowns += __popcll(__ballot(in && isn));
			matches += __popcll(__ballot(in && !isn && m));
			mismatches += __popcll(__ballot(in && !isn && !m));
		}
		// size_t arithmetic inside the parenthesis (algo_simple_bayes.c:61-65), wrap included
		const u64 free_bases = (ov >= flen && ov >= rlen) ? (u64) unknowns : (u64) flen + (u64) rlen - 2ull * (u64) ov + (u64) unknowns;
		return P.qual_nn * (double) free_bases + (double) matches * P.sb_pmatch + (double) mismatches * P.sb_pmismatch;
	} else {
		// algo_pear.c:32-59 / algo_rdp_mle.c:43-74: a sequential f64 sum over i = 0..ov-1.
		// Lanes produce the terms, lane 0 adds them in order (skipped positions add +0.0, which
		// leaves an f64 sum that started at +0.0 unchanged).
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
							term = -P.pear_random_base;	// "probability -= random_base"
						} else {
							// second index: the FORWARD read's quality at rindex (algo_pear.c:52,54);
							// past the forward read the reference is undefined, we define quality 0
							const int qb = ri < flen ? clampq(qual_of(L.F[ri])) : 0;
							term = ltab[((a & b) != 0 ? PAMD_LT_A : PAMD_LT_B) + qa * PAMD_NQ + qb];
						}
					} else {
						const int qb = clampq(qual_of(er));
						term = ltab[((a & b) != 0 ? PAMD_LT_A : PAMD_LT_B) + qa * PAMD_NQ + qb] - P.qual_nn;
					}
				}
				L.ps[i] = term;
			}
		}
		wave_sync();
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
}

// panda_result_phred (nt.c:126-150) on the staged score table
__device__ __forceinline__ int result_phred(const double *score, double p) {
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

struct WaveArena {
	u64 cur;
	u64 end;
};

// ---- one pair ---------------------------------------------------------------------------------
struct PairOut {
	pamd_pair_result rec;
	int status;
	int reject;
	int seqlen;
};

// Everything up to (not including) the HBM stores of one pair.
template <int ALGO>
__device__ void assemble_pair(const PamdKernelParams &P, const double *ltab, const uint8_t *lprimers, const WaveLds &L, int lane, const int flen, const int rlen, const uint4 fregs, const uint4 rregs, PairOut &po
#ifdef PAMD_PHASE_TIMING
	, u64 *ph_acc
#endif
) {
	PH_DECL;
	const int minov = P.minoverlap;
	const unsigned hmask = P.hash_slots - 1u;
	const unsigned hshift = P.hash_shift;

	pamd_pair_result &out = po.rec;
	out.quality = 0.0;
	out.est_overlap_prob = 0.0;
	out.seq_offset = ~0ull;
	out.seq_len = 0;
	out.overlap = 0;
	out.examined = 0;
	out.mismatches = 0;
	out.degenerates = 0;
	out.fwd_offset = 0;
	out.rev_offset = 0;
	out.status = PAMD_ST_OK;
	out.reject_module = -1;
	out.slow = 0;
#pragma unroll
	for (int k = 0; k < 7; k++)
		out.reserved[k] = 0;

	int status = PAMD_ST_OK;
	int reject = -1;
	int seqlen = 0;

	do {
		if (flen < 2 || rlen < 2) {	// assembler.c:255-258
			status = PAMD_ST_BADR;
			break;
		}
		// previous pair's readers are done with this wave's LDS
		wave_sync();
		const int uf = stage_read<true>(fregs, flen, lane, L.F, L.fc, L.fn, L.fpl);
		const int ur = stage_read<false>(rregs, rlen, lane, L.R, L.rc, L.rn, L.rpl);
		// clear the k-mer table (aliases the log-p staging area) and the candidate bitmap
		for (unsigned s4 = lane; s4 < P.hash_slots / 4; s4 += 64)
			reinterpret_cast<uint4 *>(L.hash)[s4] = make_uint4(HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY);
		if (lane < 32)
			L.bits[lane] = 0;
		wave_sync();
		PH_MARK(0);

		// ---- primer offsets / trims (assembler.c:262-288) ----
		int fo, ro;
		if (P.fprimer_len > 0) {
			const int r = primer_offset(P, ltab, L.F, flen, lprimers, P.fprimer_len, lane);
			if (r == 0) {
				status = PAMD_ST_NOFP;
				break;
			}
			fo = r - 1;
		} else {
			fo = P.ftrim;
		}
		out.fwd_offset = (uint16_t) fo;
		if (P.rprimer_len > 0) {
			const int r = primer_offset(P, ltab, L.R, rlen, lprimers + 456, P.rprimer_len, lane);
			if (r == 0) {
				status = PAMD_ST_NORP;
				break;
			}
			ro = r - 1;
		} else {
			ro = P.rtrim;
		}
		out.rev_offset = (uint16_t) ro;
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
			bool ok = p >= 8 && p < flen;
			if (ok) {	// no N among the 9 positions ending at p (misc.h:41)
				const int s = p - 8;
				const u64 win = ((u64) L.fn[(s >> 5) + 1] << 32) | (u64) L.fn[s >> 5];
				ok = ((win >> (s & 31)) & 0x1FFull) == 0;
			}
			if (ok) {
				const int s = p - 7;
				const u64 win = ((u64) L.fc[s >> 4] << 32) | (u64) L.fc[(s >> 4) + 1];
				const unsigned kmer = (unsigned) (win >> (48 - 2 * (s & 15))) & 0xFFFFu;
				const unsigned word = (kmer << 16) | (unsigned) p;
				unsigned slot = kmer_hash(kmer, hshift);
				while (atomicCAS(&L.hash[slot], HASH_EMPTY, word) != HASH_EMPTY)
					slot = (slot + 1) & hmask;
			}
		}
		wave_sync();
		PH_MARK(2);
		// ---- seed: reverse k-mers flag candidate overlaps (assembler.c:104-110) ----
		for (int base = 0; base < rlen; base += 64) {
			const int p = base + lane;
			bool ok = p + 8 < rlen;	// p <= rlen - 9
			if (ok) {	// no N among positions p .. p+8
				const u64 win = ((u64) L.rn[(p >> 5) + 1] << 32) | (u64) L.rn[p >> 5];
				ok = ((win >> (p & 31)) & 0x1FFull) == 0;
			}
			if (ok) {
				const u64 win = ((u64) L.rc[(p >> 4) + 1] << 32) | (u64) L.rc[p >> 4];
				const unsigned kmer = (unsigned) (win >> (2 * (p & 15))) & 0xFFFFu;
				unsigned slot = kmer_hash(kmer, hshift);
				unsigned m1 = 0xFFFFu, m2 = 0xFFFFu;
				unsigned wd;
				while ((wd = L.hash[slot]) != HASH_EMPTY) {
					if ((wd >> 16) == kmer) {
						const unsigned pos = wd & 0xFFFFu;
						if (pos < m1) {
							m2 = m1;
							m1 = pos;
						} else if (pos < m2) {
							m2 = pos;
						}
					}
					slot = (slot + 1) & hmask;
				}
				if (m1 != 0xFFFFu) {
					const int idx = flen + rlen - p - (int) m1 - minov - 1;	// assembler.c:107
					if (idx >= 0 && idx < size)
						atomicOr(&L.bits[idx >> 5], 1u << (idx & 31));
				}
				if (m2 != 0xFFFFu) {
					const int idx = flen + rlen - p - (int) m2 - minov - 1;
					if (idx >= 0 && idx < size)
						atomicOr(&L.bits[idx >> 5], 1u << (idx & 31));
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
		if (__ballot(mybits != 0) == 0)
			mybits = keep;	// ALL_BITS_IF_NONE
		wave_sync();	// the scan below reuses the staging area the k-mer table lived in

		double best = P.qual_nn * (double) (flen + rlen);	// assembler.c:60
		int bo = -1;
		int examined = 0;
		if (ALGO == PAMD_ALGO_SIMPLE_BAYES) {
			// four candidates per step, one per 16-lane row, still judged in ascending overlap order
			const int row = lane >> 4;
			int myov = 0;	// this row's candidate
			int pending = 0;
			for (int w = 0; w <= lastw + 1; w++) {
				unsigned m = w <= lastw ? (unsigned) __builtin_amdgcn_readlane((int) mybits, w) : 0u;
				const bool flush = w > lastw;
				while (m != 0 || (flush && pending > 0)) {
					if (m != 0) {
						const int b = __builtin_ctz(m);
						m &= m - 1;
						const int ov = w * 32 + b + minov;
						myov = row == pending ? ov : myov;
						pending++;
					}
					if (pending == 4 || (flush && m == 0)) {
						const double sc = score_sb_planes(P, L, flen, rlen, myov, row < pending, lane);
						for (int r = 0; r < pending; r++) {
							const double pr = lane_f64(sc, 16 * r);
							const int ov = __builtin_amdgcn_readlane(myov, 16 * r);
							if (pr > best) {	// strict: ascending scan keeps the smallest overlap on ties
								best = pr;
								bo = ov;
							}
						}
						examined += pending;
						pending = 0;
					}
				}
			}
		} else {
			for (int w = 0; w <= lastw; w++) {
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
		out.examined = (uint16_t) examined;
		if ((u64) examined == maxov64 - (u64) minov + 1ull)	// assembler.c:135, unsigned wrap included
			out.slow = 1;
		if (bo < 0) {
			status = PAMD_ST_NOALGN;
			break;
		}

		PH_MARK(4);
		// ---- merge (assembler.c:146-247) ----
		const int len = flen - fo - bo + rlen - ro + 1;
		if (len <= 0 || len > 2 * PAMD_MAX_LEN) {
			status = PAMD_ST_NOALGN;
			break;
		}
		seqlen = len - 1;
		const int df = flen - fo - bo;
		const int dr = rlen - ro - bo;
		const int A = df > 0 ? df : 0;		// VEEZ(df): forward-only bases
		const int wdf = df < 0 ? df : 0;	// WEDGEZ(df)
		const int wdr = dr < 0 ? dr : 0;
		const int novl = bo + wdf + wdr;	// overlap bases
		const int Bn = dr > 0 ? dr : 0;		// reverse-only bases
		const int rbase = df + bo;		// first reverse-only index (assembler.c:232)
		int mism = 0, degen = 0;
		for (int j0 = 0; j0 < seqlen; j0 += 64) {
			const int j = j0 + lane;
			unsigned nt = 0;
			double q = 0.0;
			bool ismm = false;
			const bool in = j < seqlen;
			if (in) {
				if (j < A) {	// assembler.c:162-173
					const unsigned e = L.F[j + fo];
					nt = e & 0xFFu;
					q = ltab[PAMD_LT_SCORE + clampq(qual_of(e))];
				} else if (j < A + novl) {	// assembler.c:181-228
					const int i = j - A;
					const int fi = fo + A + i;
					const int ri = rlen - i - 1 + wdf;
					const unsigned ef = L.F[fi];
					const unsigned er = L.R[ri];
					const unsigned a = ef & 0xFFu, b = er & 0xFFu;
					const int fq = qual_of(ef), rq = qual_of(er);
					const bool match = (a & b) != 0;
					ismm = !match;
					const bool fmask = fi >= uf, rmask = ri >= ur;	// B-cliff
					if (fmask && rmask) {
						q = P.qual_nn;
					} else if (fmask) {
						q = ltab[PAMD_LT_SCORE + clampq(rq)];
					} else if (rmask) {
						q = ltab[PAMD_LT_SCORE + clampq(fq)];
					} else if (ALGO == PAMD_ALGO_RDP_MLE) {	// algo_rdp_mle.c:29-41
						if (match) {
							const int mx = fq >= rq ? fq : rq;
							q = ltab[PAMD_LT_SCORE + clampq(mx)];
						} else {
							q = ltab[PAMD_LT_C + clampq(fq) * PAMD_NQ + clampq(rq)];
						}
					} else {	// algo_simple_bayes.c:68-75, algo_pear.c:61-68
						q = ltab[(match ? PAMD_LT_A : PAMD_LT_B) + clampq(fq) * PAMD_NQ + clampq(rq)];
					}
					nt = match ? (a & b) : (fq < rq ? b : a);	// assembler.c:212-220
				} else {	// assembler.c:231-243
					const int i = j - rbase;
					const int ri = rlen - bo - i - 1;
					const unsigned e = L.R[ri];
					nt = e & 0xFFu;
					q = ltab[PAMD_LT_SCORE + clampq(qual_of(e))];
				}
				L.nts[j] = (uint8_t) nt;
				L.ps[j] = q;
			}
			mism += __popcll(__ballot(in && ismm));
			degen += __popcll(__ballot(in && __popc(nt) != 1));	// PANDA_NT_IS_DEGN
		}
		(void) Bn;
		wave_sync();
		PH_MARK(5);
		// ordered sums: lane 0 forward-only, lane 1 reverse-only, lane 2 overlap
		double acc = 0.0;
		{
			int lo = 0, hi = 0;
			if (lane == 0) {
				lo = 0;
				hi = A;
			} else if (lane == 1) {
				lo = A + novl;
				hi = seqlen;
			} else if (lane == 2) {
				lo = A;
				hi = A + novl;
			}
			// same left-to-right order as the reference; the loads of a block are issued together so
			// the chain pays one LDS round trip per 8 adds instead of one per add
			int j = lo;
			for (; j + 8 <= hi; j += 8) {
				const double v0 = L.ps[j], v1 = L.ps[j + 1], v2 = L.ps[j + 2], v3 = L.ps[j + 3];
				const double v4 = L.ps[j + 4], v5 = L.ps[j + 5], v6 = L.ps[j + 6], v7 = L.ps[j + 7];
				acc += v0;
				acc += v1;
				acc += v2;
				acc += v3;
				acc += v4;
				acc += v5;
				acc += v6;
				acc += v7;
			}
			for (; j < hi; j++)
				acc += L.ps[j];
		}
		PH_MARK(6);
		const double fquality = lane_f64(acc, 0);
		const double rquality = lane_f64(acc, 1);
		const double oquality = lane_f64(acc, 2);
		const double quality = (fquality + rquality + oquality) / (double) len;	// assembler.c:244
		out.quality = quality;
		out.overlap = (uint16_t) bo;
		out.est_overlap_prob = best;
		out.mismatches = (uint16_t) mism;
		out.degenerates = (uint16_t) degen;
		out.seq_len = (uint16_t) seqlen;

		if (quality < P.threshold) {	// assembler.c:334-338
			status = PAMD_ST_LOWQ;
			break;
		}
		// ---- check modules in order (module.c:124-136) ----
		for (int mi = 0; mi < P.n_modules && reject < 0; mi++) {
			const PamdModule md = P.modules[mi];
			if (md.kind == 1) {	// plugin_min_phred.c:8-22
				bool bad = false;
				for (int j0 = 0; j0 < seqlen; j0 += 64) {
					const int j = j0 + lane;
					if (j < seqlen && result_phred(ltab + PAMD_LT_SCORE, L.ps[j]) < md.phred)
						bad = true;
				}
				if (__ballot(bad) != 0)
					reject = mi;
			} else if (md.kind == 2) {	// plugin_min_overlapbits.c:17-23
				if (!(md.nats <= best))
					reject = mi;
			}
		}
		if (reject >= 0) {
			status = PAMD_ST_REJECT;
			break;
		}
	} while (0);

	po.status = status;
	po.reject = reject;
	po.seqlen = seqlen;
	out.status = (uint8_t) status;
	out.reject_module = (int8_t) reject;
	PH_MARK(7);
}

// The HBM stores of one pair: merged row (accepted pairs only), 48-byte record, workgroup counters.
__device__ void emit_pair(const PamdKernelParams &P, unsigned *lctr, const WaveLds &L, WaveArena &arena, u64 pair, int lane, PairOut &po) {
	pamd_pair_result &out = po.rec;
	const int status = po.status, reject = po.reject, seqlen = po.seqlen;
	if (status == PAMD_ST_OK) {
		const u64 need = ((u64) seqlen + 15ull) & ~15ull;
		u64 off;
		bool fits = true;
		if (P.seq_stride != 0) {
			off = pair * P.seq_stride;
		} else {
			if (arena.cur + need > arena.end) {
				u64 got = 0;
				if (lane == 0)
					got = atomicAdd(P.seq_cursor, ARENA_CHUNK);
				union {
					u64 v;
					int i[2];
				} x;
				x.v = got;
				x.i[0] = uni(x.i[0]);
				x.i[1] = uni(x.i[1]);
				arena.cur = x.v;
				arena.end = x.v + ARENA_CHUNK;
			}
			off = arena.cur;
			arena.cur += need;
			fits = off + need <= P.seq_capacity;
		}
		if (fits) {
			out.seq_offset = off;
			// 16 nt bytes per lane, 2 log-p per lane: full 16-byte stores, rows are 16-base aligned
			if (lane * 16 < seqlen)
				reinterpret_cast<uint4 *>(P.seq_nt + off)[lane] = reinterpret_cast<const uint4 *>(L.nts)[lane];
			for (int j0 = 0; j0 < seqlen; j0 += 128) {
				const int j = j0 + lane * 2;
				if (j < seqlen)
					reinterpret_cast<double2 *>(P.seq_p + off)[j >> 1] = reinterpret_cast<const double2 *>(L.ps)[j >> 1];
			}
		} else if (lane == 0) {
			atomicAdd(P.error_flag, 1ull);
		}
	}
	if (lane == 0) {
		P.results[pair] = out;
		// workgroup-local counters (assembler.h:59-67,75); flushed once per workgroup
		atomicAdd(&lctr[PAMD_CTR_COUNT], 1u);
		if (out.slow)
			atomicAdd(&lctr[PAMD_CTR_SLOW], 1u);
		switch (status) {
		case PAMD_ST_OK:
			atomicAdd(&lctr[PAMD_CTR_OK], 1u);
			atomicAdd(&lctr[PAMD_CTR_OVERLAP0 + out.overlap], 1u);
			break;
		case PAMD_ST_BADR:
			atomicAdd(&lctr[PAMD_CTR_BADR], 1u);
			break;
		case PAMD_ST_NOFP:
			atomicAdd(&lctr[PAMD_CTR_NOFP], 1u);
			break;
		case PAMD_ST_NORP:
			atomicAdd(&lctr[PAMD_CTR_NORP], 1u);
			break;
		case PAMD_ST_NOALGN:
			atomicAdd(&lctr[PAMD_CTR_NOALGN], 1u);
			break;
		case PAMD_ST_LOWQ:
			atomicAdd(&lctr[PAMD_CTR_LOWQ], 1u);
			break;
		default:
			atomicAdd(&lctr[PAMD_CTR_REJECTED0 + reject], 1u);
			break;
		}
	}
}

template <int ALGO>
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
	L.fpl = wb + P.w_fpl;
	L.rpl = wb + P.w_rpl;
	for (unsigned i = lane; i < 2 * PLANE_BYTES / 4; i += 64)
		reinterpret_cast<uint32_t *>(wb + P.w_fpl)[i] = 0;	// plane pads stay zero for the whole kernel
	L.bits = reinterpret_cast<uint32_t *>(wb + P.w_bits);
	L.nts = reinterpret_cast<uint8_t *>(wb + P.w_nt);
	L.ps = reinterpret_cast<double *>(wb + P.w_p);
	L.hash = reinterpret_cast<uint32_t *>(wb + P.w_p);

	WaveArena arena;
	arena.cur = 0;
	arena.end = 0;
	const u64 total_waves = (u64) gridDim.x * (u64) nwaves;
#ifdef PAMD_PHASE_TIMING
	u64 ph_acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	const u64 ph_begin = __builtin_amdgcn_s_memtime();
#endif
	// Persistent loop with a one-pair software prefetch: the four loads of pair k+1 (two lengths,
	// two read rows) are issued before pair k is assembled and are first touched when k is done.
	// No branch surrounds them (indices are clamped instead), so the waits stay counted.
	u64 pair = (u64) blockIdx.x * (u64) nwaves + (u64) wave;
	const u64 last = P.n_pairs - 1;
	u64 q = pair < P.n_pairs ? pair : last;
	int flen_v = P.flen[q];
	int rlen_v = P.rlen[q];
	uint4 fregs = load_read(P.fwd + q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
	uint4 rregs = load_read(P.rev + q * P.rev_row_bytes, P.rev_row_bytes, lane);
	// settle the prologue loads before the loop so that, at the loop header, the carried registers
	// are known-ready on both incoming edges (otherwise every use inside the loop gets a vmcnt wait
	// that also waits for the previous pair's stores)
	asm volatile("" : "+v"(flen_v), "+v"(rlen_v), "+v"(fregs.x), "+v"(fregs.y), "+v"(fregs.z), "+v"(fregs.w), "+v"(rregs.x), "+v"(rregs.y), "+v"(rregs.z), "+v"(rregs.w));
	while (pair < P.n_pairs) {
		const u64 next = pair + total_waves;
		q = next < P.n_pairs ? next : last;
		const int nflen_v = P.flen[q];
		const int nrlen_v = P.rlen[q];
		const uint4 nf = load_read(P.fwd + q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
		const uint4 nr = load_read(P.rev + q * P.rev_row_bytes, P.rev_row_bytes, lane);
		const int flen = uni(flen_v), rlen = uni(rlen_v);
		PairOut po;
#ifdef PAMD_PHASE_TIMING
		assemble_pair<ALGO>(P, ltab, lprimers, L, lane, flen, rlen, fregs, rregs, po, ph_acc);
#else
		assemble_pair<ALGO>(P, ltab, lprimers, L, lane, flen, rlen, fregs, rregs, po);
#endif
		// Rotate the prefetched registers in BEFORE this pair's stores are issued: vmcnt counts loads
		// and stores in one in-order queue, so waiting for the (long finished) loads here is free,
		// while waiting for them at the top of the next iteration would also wait for the stores.
		flen_v = nflen_v;
		rlen_v = nrlen_v;
		fregs = nf;
		rregs = nr;
		asm volatile("" : "+v"(flen_v), "+v"(rlen_v), "+v"(fregs.x), "+v"(fregs.y), "+v"(fregs.z), "+v"(fregs.w), "+v"(rregs.x), "+v"(rregs.y), "+v"(rregs.z), "+v"(rregs.w));
		emit_pair(P, lctr, L, arena, pair, lane, po);
		pair = next;
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
	switch (p.algo) {
	case PAMD_ALGO_SIMPLE_BAYES:
		kern = pamd_assemble_kernel<PAMD_ALGO_SIMPLE_BAYES>;
		break;
	case PAMD_ALGO_PEAR:
		kern = pamd_assemble_kernel<PAMD_ALGO_PEAR>;
		break;
	case PAMD_ALGO_RDP_MLE:
		kern = pamd_assemble_kernel<PAMD_ALGO_RDP_MLE>;
		break;
	default:
		pamd_set_error("unknown algorithm %d", p.algo);
		return PAMD_EINVAL;
	}
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
