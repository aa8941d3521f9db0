// pamd_fast.hip -- the lean assemble kernel for gfx950 (CDNA4): ordinary read pairs, as few instructions as possible.
//
// pamd_kernels.hip assembles ANY pair under ANY configuration; its wave spends ~900 vector instructions and
// two dozen exposed LDS round trips on a 2x250 pair.  Nearly every pair of a real run is ordinary: reads of
// at most 256 bases that hold nothing but A C G T N with qualities 0..46, a handful of k-mer-seeded overlaps,
// no primers.  This kernel assembles exactly those pairs -- same path, same reference lines (assemble_seq(),
// assembler.c:252-348; align(), assembler.c:48-250), bit-identical results -- and hands every other pair, by
// index, to the general kernel through a device list (finish64<.., DEFER>): a pair is deferred, never
// approximated.  One 64-lane wavefront per pair, persistent workgroups, tables staged in LDS once.
//
//   phase        what is different from the general kernel
//   stage        4 bases per lane; the same byte-parallel packing also proves "only A C G T N, qualities 0..46"
//   seed         k-mers come from the neighbouring lanes' packed codes (DPP wave shifts), not from LDS windows;
//                forward k-mers go into four direct-mapped LDS tables by ds_min (the loser of a slot moves to
//                the next table), so all inserts of a level are in flight together and nothing loops; a
//                reverse k-mer reads its slot of every table at once (assembler.c:92-110, misc.h:41-45)
//   scan         at most four candidates, one per 16-lane DPP row, from whole words of the 2-bit code streams
//                (algo_simple_bayes.c:33-66 and the other counting scorers)
//   merge        no clamps, no IUPAC cases; B-cliff handling only when a read has a quality-2 tail
//                (assembler.c:146-247)
//   finish       64 pairs at a time from registers (finish64), quality sums in any order with the LOWQ verdict
//                guarded as in the general kernel's fast-sums mode
//
// Compiled with -ffp-contract=off like the rest (no FMA contraction: Makefile.am:62-66 builds the reference without).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "pamd_internal.h"
#include "pamd_device.h"

namespace {

using namespace pamd_dev;

constexpr unsigned EMPTY = 0xFFFFFFFFu;

// ---- per-wave LDS (bytes) ------------------------------------------------------------------------------
// reads as panda_qual rows; 2-bit code streams and N maps (16 bases per word; the scan reads up to one word either side
// of a stream: the neighbouring arrays, masked); candidate flags, one byte per overlap; merged-nt staging (+64: lanes past
// a row's end write into slack); the four k-mer tables
constexpr unsigned FK_F = 0, FK_R = 512, FK_FC = 1024, FK_RC = 1088, FK_FN = 1152, FK_RN = 1216, FK_FLAGS = 1296, FK_NT = 1552;
// the first k-mer table has 2^T1B slots: 1024 where the scorer's tables leave room (36 KB of them per workgroup), 512 for
// rdp_mle (54 KB: a third matrix); the other three are fixed
constexpr unsigned FK_T2_SLOTS = 256, FK_T3_SLOTS = 64, FK_T4_SLOTS = 32;
constexpr unsigned FK_T1 = 2128;
template <int T1B>
struct Geo {
	static constexpr unsigned T1_SLOTS = 1u << T1B;
	static constexpr unsigned T2 = FK_T1 + 4u * T1_SLOTS, T3 = T2 + 4u * FK_T2_SLOTS, T4 = T3 + 4u * FK_T3_SLOTS;
	static constexpr unsigned WAVE_BYTES = T4 + 4u * FK_T4_SLOTS;
	static_assert(WAVE_BYTES % 16 == 0, "16-byte LDS accesses");
};
static_assert(FK_NT % 16 == 0 && FK_T1 % 16 == 0 && FK_FLAGS % 16 == 0, "16-byte LDS accesses");
template <int ALGO>
__device__ __host__ constexpr int t1_bits() {
	return ALGO == PAMD_ALGO_RDP_MLE ? 9 : 10;	// (512 slots for every scorer: measured 2 % slower on config 1)
}

// k-mer -> slot of table 1..4: bit fields of one 24-bit multiplicative hash (a second one for the rarely used last table)
constexpr unsigned HASH_C = 0x9E3779u, HASH_C4 = 0x85EBCBu;

__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) {
	return (unsigned) __umul24(a, b);	// (this ROCm's __umul24 yields a SIGNED value: a plain >> on it is an arithmetic shift)
}
template <int T1B>
__device__ __forceinline__ unsigned slot1(unsigned k) {
	return mul24(k, HASH_C) >> (32 - T1B);
}
__device__ __forceinline__ unsigned slot2(unsigned k) {
	return (mul24(k, HASH_C) >> 14) & (FK_T2_SLOTS - 1u);
}
__device__ __forceinline__ unsigned slot3(unsigned k) {
	return (mul24(k, HASH_C) >> 8) & (FK_T3_SLOTS - 1u);
}
__device__ __forceinline__ unsigned slot4(unsigned k) {
	return (mul24(k, HASH_C4) >> 20) & (FK_T4_SLOTS - 1u);
}

// lane l receives lane l-1's value (lane 0: `none`) / lane l+1's value (lane 63: `none`): DPP wave shifts, no LDS
template <unsigned NONE>
__device__ __forceinline__ unsigned from_lane_below(unsigned v) {
	return (unsigned) __builtin_amdgcn_update_dpp((int) NONE, (int) v, 0x138, 0xF, 0xF, false);	// wave_shr:1
}
template <unsigned NONE>
__device__ __forceinline__ unsigned from_lane_above(unsigned v) {
	return (unsigned) __builtin_amdgcn_update_dpp((int) NONE, (int) v, 0x130, 0xF, 0xF, false);	// wave_shl:1
}

// word w (0..3) of the 16-entry byte table nt -> k-mer code (bits 0..1) | N (bit 2) | neither A C G T nor N (bit 3).
// misc.h:41: T=3 G=2 C=1, everything else 0; the reverse read stores C=2 G=1 (see stage).
template <bool FORWARD>
__device__ __forceinline__ constexpr unsigned lut_word(int w) {
	unsigned v = 0;
	for (int k = 0; k < 4; k++) {
		const int nt = 4 * w + k;
		const unsigned code = ((FORWARD ? 0x00030210u : 0x00030120u) >> (2 * nt)) & 3u;
		const unsigned isn = nt == 15 ? 1u : 0u;
		const unsigned other = (nt == 1 || nt == 2 || nt == 4 || nt == 8 || nt == 15) ? 0u : 1u;
		v |= (code | (isn << 2) | (other << 3)) << (8 * k);
	}
	return v;
}

// ---- stage one read: four bases per lane -------------------------------------------------------------------
// Writes the row, its 2-bit codes and its N map to LDS; returns the B-cliff length (read length minus the trailing
// run of raw quality 2, assembler.c:176-177).  codes: base k of this lane at bits 2k..2k+1.  nmap: bit 2k (forward) or
// 2k+1 (reverse) set when base k is N -- or does not exist: past the end of a read everything reads as N, which is
// what keeps k-mer windows inside the read.  weird: non-zero when one of this lane's bases is outside the lean domain.
// The reverse stream stores every code with its two bits swapped (C=2, G=1) and its N flags on the odd bits: a
// bit-reversed window of it then reads as forward codes in reversed base order, which is how a reverse k-mer meets a
// forward one (base t against base 7-t) and how the scan pairs forward position fi with reverse position ri.
template <bool FORWARD>
__device__ __forceinline__ int stage(const uint2 v, const int len, int lane, unsigned char *wb, unsigned &codes, unsigned &nmap, unsigned &weird) {
	reinterpret_cast<uint2 *>(wb + (FORWARD ? FK_F : FK_R))[lane] = v;
	const unsigned n4 = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);	// the four nt bytes
	const unsigned q4 = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);	// the four quality bytes
	const unsigned sel = n4 & 0x07070707u;
	const unsigned b3 = (n4 >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;	// 0xFF where nt >= 8
	const unsigned lo = __builtin_amdgcn_perm(lut_word<FORWARD>(1), lut_word<FORWARD>(0), sel);
	const unsigned hi = __builtin_amdgcn_perm(lut_word<FORWARD>(3), lut_word<FORWARD>(2), sel);
	const unsigned r4 = (hi & m3) | (lo & ~m3);
	int cnt = len - 4 * lane;
	cnt = cnt < 0 ? 0 : (cnt > 4 ? 4 : cnt);	// how many of this lane's bases exist
	const unsigned bm = (unsigned) ((0xFFFFFFFFull << (8 * cnt)) >> 32);	// their bytes
	const unsigned wq = (((q4 & 0x7F7F7F7Fu) + 0x51515151u) | q4) & 0x80808080u;	// a quality outside 0..46
	weird |= ((r4 & 0x08080808u) | wq | (n4 & 0xF0F0F0F0u)) & bm;
	const unsigned c4 = r4 & 0x03030303u, isn4 = (r4 >> 2) & 0x01010101u;
	const unsigned cy = c4 | (c4 >> 6), ny = isn4 | (isn4 >> 6);
	const unsigned in2 = (1u << (2 * cnt)) - 1u;
	codes = (cy | (cy >> 12)) & in2;
	const unsigned nb = ((ny | (ny >> 12)) | ~in2) & 0x55u;
	nmap = FORWARD ? nb : nb << 1;
	wb[(FORWARD ? FK_FC : FK_RC) + lane] = (unsigned char) codes;
	wb[(FORWARD ? FK_FN : FK_RN) + lane] = (unsigned char) nmap;
	// the usual read does not end in quality 2: one lane read settles it
	const unsigned lastq = (unsigned) __builtin_amdgcn_readlane((int) q4, (len - 1) >> 2);
	if (((lastq >> (8 * ((len - 1) & 3))) & 0xFFu) != 2u)
		return len;
	const unsigned x = q4 ^ 0x02020202u;
	const unsigned t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);	// 0x80 where quality == 2
	const unsigned is2 = (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
	const unsigned not2 = ~is2 & ((1u << cnt) - 1u);
	const int mine = not2 ? lane * 4 + (31 - __builtin_clz(not2)) + 1 : 0;
	const u64 has = ballot(not2 != 0);
	if (has == 0)
		return 0;
	return __builtin_amdgcn_readlane(mine, 63 - __builtin_clzll(has));
}

// counts over the 16 overlap positions i0 .. i0+15 of candidate `ov` (ov <= both read lengths, so every position
// below ov exists on both reads): unknowns (an N on either side) in bits 0..9, matching bases in bits 10..
// (algo_simple_bayes.c:40-59).  Forward bases flen-ov+i ascending; reverse bases rlen-1-i descending: the reverse words
// are taken at the mirrored position and bit-reversed.  Reads outside a stream stay inside the wave's LDS and are masked.
__device__ __forceinline__ unsigned word_counts(const unsigned char *wb, int flen, int rlen, int ov, int i0, bool active) {
	int hi = ov - i0;
	hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
	const unsigned valid = active ? ((unsigned) ((1ull << (2 * hi)) - 1ull) & 0x55555555u) : 0u;
	const int fb = 2 * (flen - ov + i0), rb = 2 * (rlen - 16 - i0);
	const uint32_t *fcw = reinterpret_cast<const uint32_t *>(wb + FK_FC) + (fb >> 5), *fnw = reinterpret_cast<const uint32_t *>(wb + FK_FN) + (fb >> 5);
	const uint32_t *rcw = reinterpret_cast<const uint32_t *>(wb + FK_RC) + (rb >> 5), *rnw = reinterpret_cast<const uint32_t *>(wb + FK_RN) + (rb >> 5);
	const unsigned fc0 = fcw[0], fc1 = fcw[1], fn0 = fnw[0], fn1 = fnw[1];
	const unsigned rc0 = rcw[0], rc1 = rcw[1], rn0 = rnw[0], rn1 = rnw[1];
	const unsigned f = __builtin_amdgcn_alignbit(fc1, fc0, (unsigned) fb & 31u);
	const unsigned fN = __builtin_amdgcn_alignbit(fn1, fn0, (unsigned) fb & 31u);
	const unsigned r = __builtin_bitreverse32(__builtin_amdgcn_alignbit(rc1, rc0, (unsigned) rb & 31u));
	const unsigned rN = __builtin_bitreverse32(__builtin_amdgcn_alignbit(rn1, rn0, (unsigned) rb & 31u));
	const unsigned x = f ^ r;
	const unsigned same = ~(x | (x >> 1));	// bit 2k: the two codes are equal
	const unsigned isn = fN | rN;		// bit 2k: an N on either side
	return (unsigned) __popc(valid & isn) | ((unsigned) __popc(valid & same & ~isn) << 10);
}

// ---- many candidates: pairs without a seeded overlap (every overlap is a candidate) and tandem repeats -----------------
// Every lane scores up to four candidates by itself, 16 positions per step with the same whole-word arithmetic as above and
// no cross-lane traffic.  Candidates are dealt so that the lanes finish together: with the candidates in ascending
// order, lane l takes numbers l and 127 - l (first half of the steps) and 128 + l and 255 - l (second half) -- a short
// overlap always travels with a long one.  Returns what the ascending scan of assembler.c:120-133 returns: the largest
// score above `best`, the smallest overlap among equals.  Out of line: it runs for three pairs in a hundred.
struct ManyResult {
	double best;
	int bo, examined;
};

template <int ALGO>
__device__ __attribute__((noinline)) ManyResult scan_many(unsigned char *wb, int all, int size, int minov, int flen, int rlen, double qual_nn, double pmatch, double pmismatch, const double *count_table, double best, int lane) {
	// the candidates in ascending order, one byte each (overlap - minov < 256), where the k-mer tables were
	uint8_t *list = wb + FK_T1;
	int total = size;
	if (!all) {
		const unsigned mine = reinterpret_cast<const unsigned *>(wb + FK_FLAGS)[lane];	// flag bytes are 0 or 1
		const unsigned c = (mine * 0x01010101u) >> 24;	// how many of this lane's four overlaps are flagged
		unsigned before = 0;
		total = 0;
#pragma unroll
		for (int b = 0; b < 3; b++) {
			const u64 m = ballot(((c >> b) & 1u) != 0);
			before += (unsigned) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u)) << b;
			total += __popcll(m) << b;
		}
		wave_sync();
#pragma unroll
		for (int k = 0; k < 4; k++) {
			if ((mine >> (8 * k)) & 1u) {
				list[before] = (uint8_t) (4 * lane + k);
				before++;
			}
		}
		wave_sync();
	}
	int ov[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int c = k == 0 ? lane : (k == 1 ? 127 - lane : (k == 2 ? 128 + lane : 255 - lane));
		const int idx = all ? c : (int) list[c < total ? c : 0];
		ov[k] = c < total ? idx + minov : 0;
	}
	unsigned acc[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
	for (int half = 0; half < 2; half++) {
		const int ova = ov[2 * half], ovb = ov[2 * half + 1];
		const int wa = (ova + 15) >> 4;
		int steps = wa + ((ovb + 15) >> 4);
		for (int o = 32; o > 0; o >>= 1) {
			const int other = __shfl_xor(steps, o);
			steps = other > steps ? other : steps;
		}
		steps = uni(steps);
		unsigned a = 0, b = 0;
		for (int s = 0; s < steps; s++) {
			const bool first = s < wa;
			const int o = first ? ova : ovb;
			const int i0 = 16 * (first ? s : s - wa);
			const unsigned cnt = word_counts(wb, flen, rlen, o, i0, i0 < o);
			a += first ? cnt : 0u;
			b += first ? 0u : cnt;
		}
		acc[2 * half] = a;
		acc[2 * half + 1] = b;
	}
	// this lane's best, its candidates in ascending order; then the wave's
	double pr = -__builtin_inf();
	int bov = 0x7FFFFFFF;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const double sc = count_score<ALGO>(qual_nn, pmatch, pmismatch, count_table, flen, rlen, ov[k], (int) (acc[k] & 1023u), (int) (acc[k] >> 10), ov[k]);
		if (ov[k] != 0 && sc > pr) {
			pr = sc;
			bov = ov[k];
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		const double opr = __shfl_xor(pr, o);
		const int oov = __shfl_xor(bov, o);
		if (opr > pr || (opr == pr && oov < bov)) {
			pr = opr;
			bov = oov;
		}
	}
	ManyResult out;
	out.best = best;
	out.bo = -1;
	out.examined = total;
	if (pr > best) {	// strict, as every step of the ascending scan is
		out.best = pr;
		out.bo = bov;
	}
	return out;
}

// LDS by byte address: the compiler then sees plain integer arithmetic (no symbolic base to add per access) and folds
// constant steps into the instructions' offset field
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
typedef __attribute__((address_space(3))) const double lds_cf64;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(1))) double gbl_f64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gbl_u128;
__device__ __forceinline__ unsigned lds_read_u16(unsigned addr) {
	return *reinterpret_cast<lds_cu16 *>(addr);
}
__device__ __forceinline__ double lds_read_f64(unsigned addr) {
	return *reinterpret_cast<lds_cf64 *>(addr);
}
__device__ __forceinline__ void lds_write_u8(unsigned addr, unsigned v) {
	*reinterpret_cast<lds_u8 *>(addr) = (uint8_t) v;
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
	return (unsigned) (uintptr_t) p;	// a flat pointer into LDS carries the LDS byte address in its low half
}

// (x << S) + y in one instruction (the compiler prefers a shift, a mask and an add when it sees (x >> a) << S)
template <int S>
__device__ __forceinline__ unsigned lshl_add(unsigned x, unsigned y) {
	unsigned r;
	asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(S), "v"(y));
	return r;
}
__device__ __forceinline__ unsigned lds_read_u32(unsigned addr) {
	return *reinterpret_cast<__attribute__((address_space(3))) const unsigned *>(addr);
}

struct WaveArena {
	u64 cur;
	u64 end;
};

// the registers a wave carries from one pair to the next (software prefetch)
struct Carry {
	int flen_v, rlen_v;
	uint2 f, r;
};

__device__ __forceinline__ void settle(Carry &c) {	// the compiler must treat the carried registers as written here
	asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.r.x), "+v"(c.r.y));
}

__device__ __forceinline__ uint2 load_row(const uint8_t *row, unsigned row_bytes, int lane) {
	unsigned off = (unsigned) lane * 8u;
	off = off < row_bytes ? off : row_bytes - 8u;	// lanes past the row's end re-read its last bytes (rows are whole multiples of 16 B)
	return *reinterpret_cast<const uint2 *>(row + off);
}

// one level of the forward insert: the smaller of (slot, carry) stays in the slot, the larger one moves on
// (EMPTY = nothing moves on).  The order of the words is irrelevant; ds_min only makes "who stays" independent of the
// order in which lanes and instructions reach a slot, so no lane ever has to retry.
__device__ __forceinline__ unsigned insert_level(unsigned *table, unsigned slot, unsigned carry) {
	if (carry != EMPTY) {	// lanes with nothing to insert stay out of the LDS (they would all meet on one slot)
		const unsigned old = atomicMin(&table[slot], carry);
		carry = old > carry ? old : carry;
	}
	return carry;
}
__device__ __forceinline__ unsigned insert_at(unsigned lds_byte_addr, unsigned carry) {	// the same, by LDS byte address
	if (carry != EMPTY) {
		// (through a generic pointer for the overload's sake; the compiler sees where it came from and emits ds_min_rtn_u32)
		const unsigned old = atomicMin((unsigned *) reinterpret_cast<__attribute__((address_space(3))) unsigned *>(lds_byte_addr), carry);
		carry = old > carry ? old : carry;
	}
	return carry;
}

// why pairs are deferred, counted in P.phase_cycles[0..5] (read back with pamd_debug_phase_cycles): 0 an nt or quality
// outside the lean domain, 1 a k-mer found no slot, (2, 3: unused) 4 quality too close to the LOWQ threshold; (debug builds) 5 counts pairs that used the last two k-mer tables
__device__ __forceinline__ void note(const PamdKernelParams &P, int what, int lane) {
	if (lane == 0 && P.phase_cycles != nullptr)
		atomicAdd(&P.phase_cycles[what], 1ull);
}

// -DPAMD_MARK: phase names as comments in the ISA (make asm EXTRA=-DPAMD_MARK; tools/isa_phases.py counts between them)
#ifdef PAMD_MARK
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name)
#endif

// -DFK_STOP_AFTER=k: every pair is abandoned (as NOALGN) after phase k, so build k minus build k-1 is what phase k
// costs under real occupancy (tools/fk_phases.sh); never in a shipped build
#ifdef FK_STOP_AFTER
#define STOP(k)                          \
	if ((k) == FK_STOP_AFTER) {      \
		status = PAMD_ST_NOALGN; \
		break;                   \
	}
#else
#define STOP(k)
#endif

#ifdef PAMD_FK_DEBUG
__device__ __forceinline__ void debug_dump(const PamdKernelParams &P, const unsigned char *wb, u64 pair, int lane, int where, int a, int b, int c, int d, int e, int f) {
	if (P.debug_buf == nullptr || pair != P.debug_pair)
		return;
	unsigned char *dst = P.debug_buf + (size_t) where * 8192u;
	for (unsigned i = lane; i < Geo<10>::WAVE_BYTES / 4; i += 64)
		reinterpret_cast<unsigned *>(dst)[i] = reinterpret_cast<const unsigned *>(wb)[i];
	if (lane == 0) {
		int *s = reinterpret_cast<int *>(dst + 7680);
		s[0] = a; s[1] = b; s[2] = c; s[3] = d; s[4] = e; s[5] = f;
	}
}
#define DEBUG_DUMP(...) debug_dump(__VA_ARGS__)
#else
#define DEBUG_DUMP(...)
#endif

template <int ALGO, bool EXACT, bool PHRED>
__device__ __forceinline__ void fast_pair(const PamdKernelParams &P, const double *ltab, unsigned char *wb, WaveArena &arena, Pend64 &pd, int slot_idx, bool &deep_dirty, u64 pair, int lane, const int flen, const int rlen, Carry &cur, const Carry &nxt) {
	const uint2 fv = cur.f, rv = cur.r;
	bool rotated = false;
	int status = PAMD_ST_OK;
	u64 off = ~0ull;
	double best = 0.0, qsum = 0.0;
	int seqlen = 0, bo = -1, examined = 0, mism = 0, degen = 0, phmin = 127;
	bool slow = false;
	const int minov = P.minoverlap;
	constexpr int T1B = t1_bits<ALGO>();
	typedef Geo<T1B> G;
	unsigned *const t1 = reinterpret_cast<unsigned *>(wb + FK_T1), *const t2 = reinterpret_cast<unsigned *>(wb + G::T2);
	unsigned *const t3 = reinterpret_cast<unsigned *>(wb + G::T3), *const t4 = reinterpret_cast<unsigned *>(wb + G::T4);
	uint8_t *const flags = wb + FK_FLAGS;
	uint8_t *const nts = wb + FK_NT;

	do {
		if (flen < 2 || rlen < 2) {	// assembler.c:255-258
			status = PAMD_ST_BADR;
			break;
		}
		if ((flen < rlen ? flen : rlen) < minov) {	// assembler.c:289-292 (no primers, no trims here)
			status = PAMD_ST_BADR;
			break;
		}
		if (minov >= flen || minov >= rlen) {	// assembler.c:73-82
			status = PAMD_ST_NOALGN;
			break;
		}
		wave_sync();	// the previous pair's readers are done with this wave's LDS
		MARK("stage");
		// ---- stage; clear the k-mer tables and the candidate flags ----
		unsigned fcodes, fnm, rcodes, rnm, weird = 0;
		const int uf = stage<true>(fv, flen, lane, wb, fcodes, fnm, weird);
		const int ur = stage<false>(rv, rlen, lane, wb, rcodes, rnm, weird);
		{
			const unsigned n16 = (deep_dirty ? G::WAVE_BYTES - FK_T1 : G::T3 - FK_T1) / 16u;
			uint4 *h = reinterpret_cast<uint4 *>(wb + FK_T1);
			for (unsigned s = lane; s < n16; s += 64)
				h[s] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
			if (lane < 16)
				reinterpret_cast<uint4 *>(flags)[lane] = make_uint4(0u, 0u, 0u, 0u);
			deep_dirty = false;
		}
		if (ballot(weird != 0) != 0) {	// an IUPAC ambiguity code, nt 0, a quality outside 0..46: the general kernel's pair
			note(P, 0, lane);
			status = PAMD_ST_DEFER;
			break;
		}
		wave_sync();
		STOP(0);
		MARK("seed-fwd");

		// ---- seed: forward k-mers (assembler.c:92-101) ----
		// This lane's four positions p = 4*lane + j.  The k-mer ending at p covers bases p-7 .. p and is emitted iff
		// p >= 8 and none of the nine bases p-8 .. p is N (misc.h:41): both windows lie in the codes of lanes l-2, l-1, l
		// (what lies before the read, like what lies behind it, reads as N).
		const int maxov = flen < rlen ? flen : rlen;	// assembler.c:59 with no configured maximum
		const int size = maxov - minov + 1;		// minov <= maxov here
		bool deep = false;
		{
			const unsigned x0 = fcodes | (fnm << 8);
			const unsigned x1 = from_lane_below<0x5500u>(x0), x2 = from_lane_below<0x5500u>(x1);
			const unsigned w = (x2 & 0xFFu) | ((x1 & 0xFFu) << 8) | ((x0 & 0xFFu) << 16);		// base 4(l-2)+b at bits 2b..2b+1
			const unsigned nw = ((x2 >> 8) & 0xFFu) | (x1 & 0xFF00u) | ((x0 & 0xFF00u) << 8);	// its N flag at bit 2b
			unsigned carry[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				// the k-mer's eight codes are bits 2+2j .. 17+2j of w: shifted to the top half of the table word
				const unsigned word = ((w << (14 - 2 * j)) & 0xFFFF0000u) | (unsigned) (4 * lane + j);
				carry[j] = (nw & (0x15555u << (2 * j))) == 0 ? word : EMPTY;
			}
			const unsigned l1 = lds_addr(t1), l2 = lds_addr(t2);
#pragma unroll
			for (int j = 0; j < 4; j++)
				carry[j] = insert_at(lshl_add<2>(mul24(carry[j] >> 16, HASH_C) >> (32 - T1B), l1), carry[j]);
#pragma unroll
			for (int j = 0; j < 4; j++)
				carry[j] = insert_at(lshl_add<2>(__builtin_amdgcn_ubfe(mul24(carry[j] >> 16, HASH_C), 14, 8), l2), carry[j]);
			bool lost = false;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (ballot(carry[j] != EMPTY) != 0) {	// rare from here on: one or two words per pair
					deep = true;
					carry[j] = insert_level(t3, slot3(carry[j] >> 16), carry[j]);
					carry[j] = insert_level(t4, slot4(carry[j] >> 16), carry[j]);
					if (ballot(carry[j] != EMPTY) != 0) {
						// A word found no slot.  The words of one k-mer that did are always its smallest positions (a slot
						// keeps its minimum), so a k-mer with two of them in the tables loses nothing the reference keeps
						// (assembler.c:94-99 drops a third position as well): tandem repeats pass here.
						const unsigned k = carry[j] >> 16, kk = carry[j] & 0xFFFF0000u;
						const int kept = (int) ((t1[slot1<T1B>(k)] ^ kk) < 0x10000u) + (int) ((t2[slot2(k)] ^ kk) < 0x10000u) + (int) ((t3[slot3(k)] ^ kk) < 0x10000u) + (int) ((t4[slot4(k)] ^ kk) < 0x10000u);
						lost |= ballot(carry[j] != EMPTY && kept < 2) != 0;
					}
				}
			}
			deep_dirty = deep;
			DEBUG_DUMP(P, wb, pair, lane, 0, uf, ur, (int) deep, (int) lost, (int) carry[0], (int) carry[1]);
#ifdef PAMD_FK_DEBUG
			if (deep)
				note(P, 5, lane);	// (two pairs in three: far too many same-address atomics for a shipped build)
#endif
			if (lost) {	// a k-mer found no slot (tandem repeats do this): the general kernel's unbounded table
				note(P, 1, lane);
				status = PAMD_ST_DEFER;
				break;
			}
		}
		wave_sync();
		STOP(1);
		MARK("seed-rev");
		// ---- seed: reverse k-mers flag candidate overlaps (assembler.c:104-110) ----
		// The k-mer at p covers bases p .. p+7 and is emitted iff p <= rlen-9 and none of p .. p+8 is N: the codes of
		// lanes l, l+1, l+2.  Its two smallest forward positions are the reference's first two (kmerseen[2k], [2k+1]).
		{
			const unsigned x0 = rcodes | (rnm << 8);
			const unsigned x1 = from_lane_above<0xAA00u>(x0), x2 = from_lane_above<0xAA00u>(x1);
			const unsigned w = (x0 & 0xFFu) | ((x1 & 0xFFu) << 8) | ((x2 & 0xFFu) << 16);		// base 4l+b at bits 2b..2b+1
			const unsigned nw = ((x0 >> 8) & 0xFFu) | (x1 & 0xFF00u) | ((x2 & 0xFF00u) << 8);	// its N flag at bit 2b+1
			const unsigned br = __builtin_bitreverse32(w);	// base b now at bits 30-2b..31-2b, codes read as forward codes
			const unsigned top = (unsigned) (flen + rlen - minov - 1 - 4 * lane);	// assembler.c:107: index = top - j - forward position
			unsigned m1[4], m2[4], xk[4];
			bool ok[4];
			const unsigned l1 = lds_addr(t1), l2 = lds_addr(t2);
#pragma unroll
			for (int j = 0; j < 4; j++) {
				xk[j] = br << (2 * j);	// this k-mer in the top half
				ok[j] = (nw & (0x2AAAAu << (2 * j))) == 0;
				const unsigned prod = mul24(xk[j] >> 16, HASH_C), kk = xk[j] & 0xFFFF0000u;
				const unsigned e1 = lds_read_u32(lshl_add<2>(prod >> (32 - T1B), l1)) ^ kk;
				const unsigned e2 = lds_read_u32(lshl_add<2>(__builtin_amdgcn_ubfe(prod, 14, 8), l2)) ^ kk;
				// an entry of this k-mer reads as its bare position (< 0x10000), anything else as >= 0x10000
				m1[j] = e1 < e2 ? e1 : e2;
				m2[j] = e1 < e2 ? e2 : e1;
			}
			if (deep) {	// wave-uniform: the last two tables hold something
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const unsigned kmer = xk[j] >> 16, kk = xk[j] & 0xFFFF0000u;
					const unsigned e3 = t3[slot3(kmer)] ^ kk, e4 = t4[slot4(kmer)] ^ kk;
					const unsigned l34 = e3 < e4 ? e3 : e4, h34 = e3 < e4 ? e4 : e3;
					const unsigned hi = m1[j] < l34 ? l34 : m1[j];
					m1[j] = m1[j] < l34 ? m1[j] : l34;
					const unsigned h = m2[j] < h34 ? m2[j] : h34;
					m2[j] = hi < h ? hi : h;
				}
			}
			bool second = false;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const unsigned idx = top - (unsigned) j - m1[j];	// no match: m1 >= 0x10000 wraps far beyond size
				if (ok[j] && idx < (unsigned) size)
					flags[idx] = 1;
				second |= ok[j] && m2[j] < 0xFFFFu;	// (an empty slot reads as 0xFFFF for the k-mer 0xFFFF)
			}
			if (ballot(second) != 0) {	// a k-mer that occurs twice in the forward read
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const unsigned idx = top - (unsigned) j - m2[j];
					if (ok[j] && idx < (unsigned) size)
						flags[idx] = 1;
				}
			}
		}
		wave_sync();
		STOP(2);
		MARK("candidates");

		// ---- the candidates (assembler.c:118-143) ----
		int ncand = 0, ov0 = 0, ov1 = 0, ov2 = 0, ov3 = 0;
		bool all = false;
		{
			const unsigned mine = reinterpret_cast<const unsigned *>(flags)[lane];	// overlaps minov + 4*lane + 0..3
			u64 nz = ballot(mine != 0);
			all = nz == 0;	// ALL_BITS_IF_NONE: every overlap is a candidate
			while (nz != 0 && ncand <= 4) {
				const int w = __builtin_ctzll(nz);
				nz &= nz - 1;
				unsigned m = (unsigned) __builtin_amdgcn_readlane((int) mine, w);
				while (m != 0 && ncand <= 4) {
					const int k = __builtin_ctz(m) >> 3;
					m &= ~(0xFFu << (8 * k));
					const int ov = 4 * w + k + minov;
					if (ncand == 0)
						ov0 = ov;
					else if (ncand == 1)
						ov1 = ov;
					else if (ncand == 2)
						ov2 = ov;
					else
						ov3 = ov;
					ncand++;
				}
			}
		}
		MARK("scan");
		best = P.qual_nn * (double) (flen + rlen);	// assembler.c:60
		if (!counting_algo<ALGO>()) {
			// ---- pear / rdp_mle: an overlap's score is a sum of per-base table values (algo_pear.c:32-59,
			// algo_rdp_mle.c:43-74).  Lanes produce the terms.  Exact sums: lane 0 adds them in the reference's order.
			// Any-order sums: every lane adds its own, the wave adds the lanes (|error| < 1e-9), and a comparison closer
			// than 1e-6 -- or a min_overlapbits verdict that close -- sends the pair to the general kernel, which sums in
			// reference order.  Pairs with many candidates go there too (a candidate per lane, each summed in order).
			if (all || ncand > 4) {
				note(P, 3, lane);
				status = PAMD_ST_DEFER;
				break;
			}
			const unsigned lF = lds_addr(wb + FK_F), lR = lds_addr(wb + FK_R), lT = lds_addr(ltab);
			double *const ps = reinterpret_cast<double *>(wb + FK_T1);	// the terms of one candidate, where the k-mer tables were
			deep_dirty = deep_dirty || EXACT;
			const double nrb = -P.pear_random_base, nn = P.qual_nn;
			bool ambiguous = false;
			for (int c = 0; c < ncand; c++) {
				const int ov = c == 0 ? ov0 : (c == 1 ? ov1 : (c == 2 ? ov2 : ov3));
				double part = 0.0;
#pragma unroll
				for (int r = 0; r < 4; r++) {	// reads of at most 256 bases
					const int i = 64 * r + lane;
					if (64 * r < ov && i < ov) {
						const int fi = flen - ov + i, ri = rlen - 1 - i;
						const unsigned ef = lds_read_u16(lF + 2u * (unsigned) fi), er = lds_read_u16(lR + 2u * (unsigned) ri);
						const unsigned a = ef & 0xFFu, b = er & 0xFFu, qa = ef >> 8;
						const unsigned base = (a & b) != 0 ? PAMD_LT_A : PAMD_LT_B;
						double term;
						if (ALGO == PAMD_ALGO_PEAR) {
							// the second index is the FORWARD read's quality at the reverse index (algo_pear.c:52,54, sic);
							// past the forward read the reference reads beyond the array: defined as quality 0 here
							const unsigned qb = ri < flen ? lds_read_u16(lF + 2u * (unsigned) ri) >> 8 : 0u;
							term = (a == 15u || b == 15u) ? nrb : lds_read_f64(lT + 8u * (base + qa * PAMD_NQ + qb));
						} else {
							term = lds_read_f64(lT + 8u * (base + qa * PAMD_NQ + (er >> 8))) - nn;
						}
						if (EXACT)
							ps[i] = term;
						else
							part += term;
					}
				}
				double pr;
				if (EXACT) {
					wave_sync();
					double acc = 0.0;
					if (lane == 0) {	// the reference's own order, eight reads in flight
						int i = 0;
						for (; i + 8 <= ov; i += 8) {
							const double v0 = ps[i], v1 = ps[i + 1], v2 = ps[i + 2], v3 = ps[i + 3];
							const double v4 = ps[i + 4], v5 = ps[i + 5], v6 = ps[i + 6], v7 = ps[i + 7];
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
							acc += ps[i];
					}
					wave_sync();
					pr = uni_f64(acc);
					if (pr > best) {	// strict: the ascending scan keeps the smallest overlap on ties
						best = pr;
						bo = ov;
					}
				} else {
					pr = wave_sum_any_order(part);
					if (pr > best + 1e-6) {
						best = pr;
						bo = ov;
					} else if (!(pr < best - 1e-6)) {
						ambiguous = true;
					}
				}
			}
			if (!EXACT) {
				for (int mi = 0; mi < P.n_modules; mi++)	// plugin_min_overlapbits.c:17-23 reads this sum
					ambiguous |= P.modules[mi].kind == 2 && fabs(P.modules[mi].nats - best) <= 1e-6;
				if (ambiguous) {
					note(P, 2, lane);
					status = PAMD_ST_DEFER;
					break;
				}
			}
			bo = uni(bo);
			best = uni_f64(best);
		} else if (all || ncand > 4) {
			// ---- pairs without a seeded overlap, tandem repeats: a candidate per lane (three pairs in a hundred) ----
			const ManyResult mr = scan_many<ALGO>(wb, all ? 1 : 0, size, minov, flen, rlen, P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, best, lane);
			best = uni_f64(mr.best);	// a function's results come back in vector registers
			bo = uni(mr.bo);
			ncand = uni(mr.examined);
			deep_dirty = true;	// the candidate list lay over the k-mer tables
		} else {
			// ---- scan + argmax: one candidate per 16-lane row, 16 positions per lane ----
			const int row = lane >> 4;
			const int myov = row == 0 ? ov0 : (row == 1 ? ov1 : (row == 2 ? ov2 : ov3));
			unsigned packed = word_counts(wb, flen, rlen, myov, 16 * (lane & 15), row < ncand);
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x128, 0xF, 0xF, false);	// row_ror:8
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x124, 0xF, 0xF, false);	// row_ror:4
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x122, 0xF, 0xF, false);	// row_ror:2
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x121, 0xF, 0xF, false);	// row_ror:1
			const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
			const double sc = count_score<ALGO>(P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, flen, rlen, myov, unknowns, matches, myov);
#pragma unroll
			for (int r = 0; r < 4; r++) {
				if (r < ncand) {
					const double pr = lane_f64(sc, 16 * r);
					if (pr > best) {	// strict: the ascending scan keeps the smallest overlap on ties
						best = pr;
						bo = r == 0 ? ov0 : (r == 1 ? ov1 : (r == 2 ? ov2 : ov3));
					}
				}
			}
			// wave-uniform by construction; tell the compiler, so that everything derived from them is scalar
			bo = uni(bo);
			best = uni_f64(best);
		}
		examined = ncand;
		slow = examined == size;	// assembler.c:135
		DEBUG_DUMP(P, wb, pair, lane, 2, uf, ur, ncand, ov0, ov1, bo);
		if (bo < 0) {
			status = PAMD_ST_NOALGN;
			best = 0.0;	// (the record of a pair that did not align carries no score, as in the general kernel)
			break;
		}

		STOP(3);
		MARK("rotate+alloc");
		// Rotate the prefetched registers in BEFORE this pair's stores are issued (vmcnt is one in-order queue for
		// loads and stores: waiting for the loads at the top of the next pair would also wait for the stores).
		cur = nxt;
		settle(cur);
		rotated = true;

		// ---- merge (assembler.c:146-247): no offsets, so df = flen - bo >= 0 and dr = rlen - bo >= 0 ----
		const int A = flen - bo, Bn = rlen - bo, novl = bo;
		seqlen = A + novl + Bn;	// len - 1; len = seqlen + 1 is in 3 .. 511
		const u64 need = ((u64) seqlen + 15ull) & ~15ull;
		// (the launch parameters this block needs, read together: the compiler otherwise re-reads each from the kernel
		// argument segment, one exposed scalar-memory round trip per use, in whichever branch it is used)
		u64 k_stride = P.seq_stride, k_capacity = P.seq_capacity;
		uint8_t *k_nt = P.seq_nt;
		double *k_p = P.seq_p;
		asm volatile("" : "+s"(k_stride), "+s"(k_capacity), "+s"(k_nt), "+s"(k_p));
		if (k_stride != 0) {
			off = pair * k_stride;
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
			if (off + need > k_capacity)
				off = ~0ull;
		}
		const bool store = off != ~0ull;
		uint8_t *const row_nt = k_nt + (store ? off : 0);
		double *const row_p = k_p + (store ? off : 0);
		double qacc = 0.0;	// this lane's share of the quality sum, any order
		double pmin = 0.0;	// PHRED: this lane's smallest log-p (every log-p is <= 0)
		// One pass over the output row, 64 bases per round: lane l of round r makes base j = 64r + l, from forward base j
		// (j < flen: the forward-only bases [0, A), assembler.c:162-173, and the overlap [A, A + novl), assembler.c:181-228) and
		// from reverse base rlen - 1 - (j - A) (j >= A: the overlap and the reverse-only bases [A + novl, seqlen),
		// assembler.c:231-243): the same two index formulas in all three regions, so a 380-base row takes six rounds and every
		// round writes 512 aligned bytes of log-p.  The staged bases of all rounds are read first (one wait); then, two rounds
		// per trip, each round by its kind -- wave-uniform: all forward-only, all reverse-only (a shift and an add), or mixed
		// (the overlap's table index and base, with per-lane selects at the region borders) -- table reads, then the writes.
		// A lane past the end of the row reads whatever lies there, inside the LDS, and its base is forced to 0.
		const unsigned lF = lds_addr(wb + FK_F), lR = lds_addr(wb + FK_R), lT = lds_addr(ltab), lN = lds_addr(nts);
		// a pair whose row does not fit stores nothing (it ends as NOSPACE whatever its sums are)
#ifdef FK_NO_PSTORE	/* experiment: what do the log-p stores cost? (results are wrong) */
		const int lim = 0;
		qacc = -1.0;
#else
		const int lim = store ? seqlen : 0;
#endif
		// the B-cliff cases (assembler.c:192-207) only when a read ends in a run of quality 2
		const bool cliff = uf < flen || ur < rlen;
		const int AF = A + novl;	// forward bases end here
		const unsigned nN = lN + (unsigned) lane;
		// (explicitly global: through a generic pointer these would be flat_store, which every later LDS wait waits for as well)
		gbl_f64 *const gP = (gbl_f64 *) (row_p + lane);
		MARK("merge");
		unsigned ef[8], er[8];
		{
			const unsigned aF = lF + 2u * (unsigned) lane;					// round r: F[64r + lane] ...
			const unsigned aR = lR + 2u * (unsigned) (rlen - 1 + A - lane) - 128u * 7u;	// ... and R[rlen - 1 - (64r + lane - A)]
#pragma unroll
			for (int r = 0; r < 8; r++) {
				ef[r] = lds_read_u16(aF + 128u * r);
				er[r] = lds_read_u16(aR + 128u * (7 - r));
			}
		}
		// one round: table index and base of output base 64 R + lane
		auto round = [&](auto rc, unsigned &qi_out, unsigned &nt_out) __attribute__((always_inline)) {
			constexpr int R = decltype(rc)::value;
			const unsigned e_f = ef[R], e_r = er[R];
			if (64 * (R + 1) <= A) {	// all forward-only (assembler.c:162-173)
				qi_out = PAMD_LT_SCORE + (e_f >> 8);
				nt_out = e_f;
				return;
			}
			const bool ok = lane < lim - 64 * R;
			if (64 * R >= AF) {	// all reverse-only (assembler.c:231-243)
				qi_out = PAMD_LT_SCORE + (e_r >> 8);
				nt_out = ok ? e_r : 0u;
				return;
			}
			const int j = 64 * R + lane;
			const bool inF = j < AF, inR = j >= A;
			const unsigned ab = e_f & e_r;	// low byte: the bases the two reads agree on
			const unsigned fq = e_f >> 8, rq = e_r >> 8;
			const bool match = (ab & 0xFFu) != 0;
			unsigned nt = match ? ab : (fq < rq ? e_r : e_f);	// assembler.c:212-220 (low byte)
			unsigned qi;
			if (ALGO == PAMD_ALGO_EA_UTIL) {	// algo_ea_util.c:58-67
				qi = PAMD_LT_SCORE + (fq > rq ? fq : rq);
			} else if (ALGO == PAMD_ALGO_FLASH) {	// algo_flash.c:63-80
				const unsigned d = fq > rq ? fq - rq : rq - fq;
				qi = PAMD_LT_SCORE + (match ? (fq > rq ? fq : rq) : (d < 2 ? 2 : d));
			} else if (ALGO == PAMD_ALGO_RDP_MLE) {	// algo_rdp_mle.c:29-41
				qi = match ? PAMD_LT_SCORE + (fq > rq ? fq : rq) : PAMD_LT_C + fq * PAMD_NQ + rq;
			} else {	// algo_simple_bayes.c:68-75, algo_pear.c:61-68, algo_uparse.c:68-75, algo_stitch.c:59-67
				qi = (match ? PAMD_LT_A : PAMD_LT_B) + fq * PAMD_NQ + rq;
			}
			if (cliff) {
				asm volatile("; B-cliff");	// (keeps this a branch: as selects it would cost every pair six instructions a round)
				const bool fmask = j >= uf, rmask = rlen - 1 + A - j >= ur;
				qi = rmask ? PAMD_LT_SCORE + fq : qi;
				qi = fmask ? PAMD_LT_SCORE + (rmask ? 47u : rq) : qi;	// slot 47 of the score block holds qual_nn
			}
			qi = inR ? qi : PAMD_LT_SCORE + fq;	// forward-only
			qi = inF ? qi : PAMD_LT_SCORE + rq;	// reverse-only
			nt = inR ? nt : e_f;
			nt = inF ? nt : e_r;
			qi_out = qi;
			nt_out = ok ? nt : 0u;
			mism += __popcll(ballot(inF) & ballot(inR) & ~ballot(match));
		};
		// two rounds per trip, straight-line; trips nest so that nothing but the accumulators lives across them
		auto trip = [&](auto r0c) __attribute__((always_inline)) {
			constexpr int R0 = decltype(r0c)::value;
			unsigned qi[2], ntv[2];
			round(std::integral_constant<int, R0>(), qi[0], ntv[0]);
			round(std::integral_constant<int, R0 + 1>(), qi[1], ntv[1]);
			double pv[2];
#pragma unroll
			for (int k = 0; k < 2; k++)
				pv[k] = lds_read_f64(lT + 8u * qi[k]);
#pragma unroll
			for (int k = 0; k < 2; k++) {
				lds_write_u8(nN + 64u * (R0 + k), ntv[k]);
				if (lane < lim - 64 * (R0 + k)) {
					gP[64 * (R0 + k)] = pv[k];
					qacc += pv[k];
					if (PHRED)
						pmin = pv[k] < pmin ? pv[k] : pmin;
				}
				degen += __popcll(ballot((ntv[k] & 0xFFu) == 15u));	// PANDA_NT_IS_DEGN: only N here
			}
		};
		trip(std::integral_constant<int, 0>());
		if (seqlen > 128) {
			trip(std::integral_constant<int, 2>());
			if (seqlen > 256) {
				trip(std::integral_constant<int, 4>());
				if (seqlen > 384)
					trip(std::integral_constant<int, 6>());
			}
		}
		wave_sync();
		STOP(4);
		DEBUG_DUMP(P, wb, pair, lane, 1, A, novl, Bn, mism, degen, seqlen);
		MARK("row-out+sum");
		// the nt row leaves LDS in 16-byte stores (rows are 16-base aligned)
		if (store && lane * 16 < seqlen)
			((gbl_u128 *) row_nt)[lane] = reinterpret_cast<const u32x4 *>(nts)[lane];
		if (PHRED) {
			// panda_result_phred (nt.c:126-150) is 1 at or below qual_score[0] and does not decrease above it, so the row's
			// smallest phred is that of its smallest log-p: one bisection per pair instead of one per base
			for (int o = 32; o > 0; o >>= 1) {
				const double other = __shfl_xor(pmin, o);
				pmin = other < pmin ? other : pmin;
			}
			const double lowest = uni_f64(pmin);
			phmin = seqlen > 0 ? result_phred(ltab + PAMD_LT_SCORE, lowest) : 127;	// an empty row passes any threshold
		}
		if (!EXACT) {
			qsum = wave_sum_any_order(qacc);	// divided by len in finish64() ...
			// ... unless the quotient could land within 1e-9 of the LOWQ threshold (tested without dividing): the general
			// kernel then makes the three sums in reference order (assembler.c:171,226,236,244,334)
			if (fabs(qsum - P.threshold * (double) (seqlen + 1)) <= 1e-9 * (double) (seqlen + 1)) {
				note(P, 4, lane);
				status = PAMD_ST_DEFER;	// (its row stays behind in the arena as a hole)
				break;
			}
		}
		// (exact sums: finish64 walks the row's three chains in reference order, sixty-four pairs at a time)
	} while (0);
	MARK("park");
	if (!rotated) {
		cur = nxt;
		settle(cur);
	}
	// park the pair's scalars in lane `slot_idx` of the pending registers (finish64)
	const bool mine = lane == slot_idx;
	union {
		double d;
		unsigned u[2];
	} qq, bb;
	qq.d = qsum;
	bb.d = best;
	pd.off_lo = put_lane(pd.off_lo, (unsigned) off, mine);
	pd.off_hi = put_lane(pd.off_hi, (unsigned) (off >> 32), mine);
	pd.q_lo = put_lane(pd.q_lo, qq.u[0], mine);
	pd.q_hi = put_lane(pd.q_hi, qq.u[1], mine);
	pd.b_lo = put_lane(pd.b_lo, bb.u[0], mine);
	pd.b_hi = put_lane(pd.b_hi, bb.u[1], mine);
	pd.len_bo = put_lane(pd.len_bo, (unsigned) seqlen | ((unsigned) (bo < 0 ? 0 : bo) << 16), mine);
	pd.ex_mm = put_lane(pd.ex_mm, (unsigned) examined | ((unsigned) mism << 16), mine);
	pd.dg_fo = put_lane(pd.dg_fo, (unsigned) degen | (EXACT ? (unsigned) (seqlen - rlen) << 16 : 0u), mine);	// exact sums: + the forward-only bases
	pd.ro_st = put_lane(pd.ro_st, ((unsigned) status << 16) | ((unsigned) slow << 24) | ((unsigned) (phmin > 63 ? 63 : phmin) << 26), mine);
}

// PHRED: a min_phred module is loaded, the merge tracks the row's smallest log-p (plugin_min_phred.c:8-22)
#ifdef FK_NO_W4	/* A/B builds only */
#define FK_WAVES_ATTR
#else
#define FK_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
// (one 16-wave workgroup fills a CU's LDS: four waves per SIMD whatever the registers allow, so the compiler is told to
// spend all 128 vector registers on scheduling instead of saving them for waves that cannot come -- 1 % on a 16 M-pair
// launch, nothing measurable at 50 M)
template <int ALGO, bool EXACT, bool PHRED>
__global__ __launch_bounds__(1024) FK_WAVES_ATTR void pamd_fast_kernel(const PamdKernelParams P) {
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
	__syncthreads();

	unsigned char *wb = lds + P.lds_primer_off + (size_t) wave * Geo<t1_bits<ALGO>()>::WAVE_BYTES;	// (no primers here: the per-wave regions start where the general kernel keeps them)
	WaveArena arena;
	arena.cur = 0;
	arena.end = 0;
	// (pair indices are 32-bit here -- pamd_fast_supports() keeps the batch below 2^31 pairs -- and the arena is counted in
	// units of 16 bases: scalar registers are what this kernel runs out of, and every spilled one costs vector
	// instructions, v_writelane / v_readlane, on the way out and back)
	const unsigned total_waves = gridDim.x * (unsigned) nwaves;
	const unsigned n_pairs = (unsigned) P.n_pairs;
	// Persistent loop with a one-pair software prefetch: the four loads of pair k+1 (two lengths, two rows) are
	// issued, branch-free, before pair k is assembled and are first touched when k is done.
	unsigned pair = blockIdx.x * (unsigned) nwaves + (unsigned) wave;
	const unsigned last = n_pairs - 1u;
	unsigned q = pair < n_pairs ? pair : last;
	Carry cur;
	cur.flen_v = P.flen[q];
	cur.rlen_v = P.rlen[q];
	cur.f = load_row(P.fwd + (u64) q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
	cur.r = load_row(P.rev + (u64) q * P.rev_row_bytes, P.rev_row_bytes, lane);
	settle(cur);
	int npend = 0;
	bool deep_dirty = true;	// the first pair clears all four tables
	Pend64 pd = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	unsigned group_first = pair;
	while (pair < n_pairs) {
		const unsigned next = pair + total_waves;
		q = next < n_pairs ? next : last;
		Carry nxt;
		nxt.flen_v = P.flen[q];
		nxt.rlen_v = P.rlen[q];
		nxt.f = load_row(P.fwd + (u64) q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
		nxt.r = load_row(P.rev + (u64) q * P.rev_row_bytes, P.rev_row_bytes, lane);
		const int flen = uni(cur.flen_v), rlen = uni(cur.rlen_v);
		fast_pair<ALGO, EXACT, PHRED>(P, ltab, wb, arena, pd, npend, deep_dirty, pair, lane, flen, rlen, cur, nxt);
		MARK("loop");
		npend++;
		if (npend == 64) {
			finish64<false, true, EXACT>(P, lctr, pd, npend, group_first, total_waves, lane);
			npend = 0;
			group_first = next;
		}
		pair = next;
	}
	if (npend > 0)
		finish64<false, true, EXACT>(P, lctr, pd, npend, group_first, total_waves, lane);
	__syncthreads();
	if (P.counters != nullptr) {
		for (unsigned i = tid; i < PAMD_NUM_COUNTERS; i += blockDim.x) {
			const unsigned v = lctr[i];
			if (v != 0)
				atomicAdd(&P.counters[i], (unsigned long long) v);
		}
	}
}

}	// namespace

// ---- host side: does the lean kernel serve this configuration, and its launch ---------------------------------
int pamd_fast_supports(const PamdKernelParams &p) {
	if (p.smax > 256 || p.post_primers)
		return 0;
	if (p.fprimer_len > 0 || p.rprimer_len > 0 || p.ftrim != 0 || p.rtrim != 0 || p.maxoverlap != 0)
		return 0;
	if (p.n_pairs > 0x7FFFFFFFull)
		return 0;	// 32-bit pair indices in the kernel and in the deferred list
	return 1;
}

size_t pamd_fast_lds_bytes(const PamdKernelParams &p, int waves_per_wg) {
	return (size_t) p.lds_primer_off + (size_t) waves_per_wg * (p.algo == PAMD_ALGO_RDP_MLE ? Geo<9>::WAVE_BYTES : Geo<10>::WAVE_BYTES);
}

int pamd_launch_fast(const PamdKernelParams &p, int workgroups, int waves_per_wg, size_t lds_bytes, void *stream) {
	const dim3 grid((unsigned) workgroups), block((unsigned) waves_per_wg * 64u);
	hipStream_t st = static_cast<hipStream_t>(stream);
	void (*kern)(const PamdKernelParams) = nullptr;
#define FK_PICK2(A, E) (p.has_min_phred ? pamd_fast_kernel<A, E, true> : pamd_fast_kernel<A, E, false>)
#define FK_PICK(A) (p.exact_sums ? FK_PICK2(A, true) : FK_PICK2(A, false))
	switch (p.algo) {
	case PAMD_ALGO_SIMPLE_BAYES:
		kern = FK_PICK(PAMD_ALGO_SIMPLE_BAYES);
		break;
	case PAMD_ALGO_PEAR:
		kern = FK_PICK(PAMD_ALGO_PEAR);
		break;
	case PAMD_ALGO_RDP_MLE:
		kern = FK_PICK(PAMD_ALGO_RDP_MLE);
		break;
	case PAMD_ALGO_UPARSE:
		kern = FK_PICK(PAMD_ALGO_UPARSE);
		break;
	case PAMD_ALGO_FLASH:
		kern = FK_PICK(PAMD_ALGO_FLASH);
		break;
	case PAMD_ALGO_EA_UTIL:
		kern = FK_PICK(PAMD_ALGO_EA_UTIL);
		break;
	case PAMD_ALGO_STITCH:
		kern = FK_PICK(PAMD_ALGO_STITCH);
		break;
	default:
		pamd_set_error("the lean kernel does not serve algorithm %d", p.algo);
		return PAMD_EINVAL;
	}
#undef FK_PICK
#undef FK_PICK2
	hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
	if (e0 != hipSuccess) {
		pamd_set_error("hipFuncSetAttribute(%zu bytes LDS, lean kernel): %s", lds_bytes, hipGetErrorString(e0));
		return PAMD_ENODEVICE;
	}
	hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("lean kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
