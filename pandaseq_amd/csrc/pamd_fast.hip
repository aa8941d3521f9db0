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
#include <utility>
#include "pamd_internal.h"
#include "pamd_device.h"

namespace {

using namespace pamd_dev;

constexpr unsigned EMPTY = 0xFFFFFFFFu;

// ---- per-wave LDS (bytes) ------------------------------------------------------------------------------
// reads as panda_qual rows; 2-bit code streams and N maps (16 bases per word; the scan reads up to one word either side
// of a stream: the neighbouring arrays, masked); candidate flags, one byte per overlap; merged-nt staging (+64: lanes past
// a row's end write into slack); the four k-mer tables
// (16 bytes before the forward N map hold N flags: seeding by windows of the streams starts eight bases before the read)
constexpr unsigned FK_F = 0, FK_R = 512, FK_FC = 1024, FK_RC = 1088, FK_PAD0 = 1152, FK_FN = 1168, FK_RN = 1232, FK_FLAGS = 1312, FK_NT = 1568;
// the first k-mer table has 2^T1B slots: 1024 where the scorer's tables leave room (36 KB of them per workgroup), 512 for
// rdp_mle (54 KB: a third matrix); the other three are fixed
constexpr unsigned FK_T2_SLOTS = 256, FK_T3_SLOTS = 64, FK_T4_SLOTS = 32;
constexpr unsigned FK_T1 = 2144;
// Two geometries.  LONG = false: reads of at most 256 bases, four bases per lane (the offsets above).  LONG = true: reads of
// at most 320 bases (2x300), primers and trims: rows arrive eight bases per lane, seeding runs five bases per lane on
// windows of the code streams, and what is only needed before seeding (the primer scan's nibble streams and scratch) or
// after it (the merged-nt staging) lies over the first k-mer table, which is cleared for every pair anyway.
template <bool LONG>
struct Lay;
template <>
struct Lay<false> {
	static constexpr unsigned F = FK_F, R = FK_R, FC = FK_FC, RC = FK_RC, PAD0 = FK_PAD0, FN = FK_FN, RN = FK_RN, FLAGS = FK_FLAGS, NT = FK_NT, T1 = FK_T1;
	static constexpr int BPL = 4;		// bases per lane while seeding (3 for reads of at most 192 bases: NARROW)
	static constexpr int MAXLEN = 256;
	static constexpr int NR = 8;		// rounds of 64 bases a merged row can span
	static constexpr int SR = 4;		// rounds of 64 positions an overlap can span
};
template <>
struct Lay<true> {
	// (a pad word before the forward N map: what lies before a read reads as N; see the kernel's prologue)
	static constexpr unsigned F = 0, R = 640, FC = 1280, RC = 1376, PAD0 = 1472, FN = 1488, RN = 1584, PAD1 = 1680, FLAGS = 1696, T1 = 2080;
	static constexpr unsigned NT = T1;				// merged-nt staging: up to 640 + 64 bytes
	static constexpr unsigned CAREF = T1 + 16, CARER = T1 + 128, SCR = T1 + 256;	// primer scan: "counts" maps (96 bytes each), 512 bytes of f64 terms
	// `-a` (POST): beside the merged nt row, the table index of every merged base's log-p (u16; 640 + 64 of them), and the merged
	// sequence's own 2-bit code stream and "counts" map (160 bytes each + what a window reads past them)
	static constexpr unsigned PIDX = NT + 704, MC = PIDX + 1408, MK = MC + 176, POST_END = MK + 176;
	static constexpr int BPL = 5;
	static constexpr int MAXLEN = 320;
	static constexpr int NR = 10;
	static constexpr int SR = 5;
};
template <bool LONG, int T1B>
struct Geo {
	static constexpr unsigned T1_SLOTS = 1u << T1B;
	static constexpr unsigned T2 = Lay<LONG>::T1 + 4u * T1_SLOTS, T3 = T2 + 4u * FK_T2_SLOTS, T4 = T3 + 4u * FK_T3_SLOTS;
	static constexpr unsigned WAVE_BYTES = T4 + 4u * FK_T4_SLOTS;
	static_assert(WAVE_BYTES % 16 == 0, "16-byte LDS accesses");
};
static_assert(FK_NT % 16 == 0 && FK_T1 % 16 == 0 && FK_FLAGS % 16 == 0, "16-byte LDS accesses");
static_assert(Lay<true>::T1 % 16 == 0 && Lay<true>::FLAGS % 16 == 0 && Lay<true>::R % 16 == 0, "16-byte LDS accesses");
static_assert(Lay<true>::PIDX % 16 == 0 && Lay<true>::MC % 16 == 0 && Lay<true>::MK % 16 == 0 && Lay<true>::POST_END <= Geo<true, 9>::T3, "`-a` staging lies over the first two k-mer tables");
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) {
	return (unsigned) __umul24(a, b);	// (this ROCm's __umul24 yields a SIGNED value: a plain >> on it is an arithmetic shift)
}
// (Round 5 built a geometry of two ten-wave workgroups per CU -- the two matrices staged as lower triangles, a 512-slot first k-mer
// table, 96 and then 80 vector registers -- and measured it: a SIMD's throughput is the same with 3, 4, 5 or 6 waves, the kernel is
// bound by vector issue (profiles/r05_dense_geometry_ab.txt).  It shipped compiled out and was removed in
// round 6; `git log -S FK_DENSE` has it.)
template <int ALGO, bool LONG = false>
__device__ __host__ constexpr int t1_bits() {
	return ALGO == PAMD_ALGO_RDP_MLE ? 9 : 10;	// (512 slots for every scorer: measured 2 % slower on config 1; for reads of at most 192 bases only, round 4: 1 % slower on config 4)
}
// the index, in the staged table image, of entry (a, b) of the "match" (A) or "mismatch" (B) matrix
__device__ __forceinline__ unsigned mat_index(bool match, unsigned a, unsigned b) {
	return (match ? (unsigned) PAMD_LT_A : (unsigned) PAMD_LT_B) + a * PAMD_NQ + b;
}

// k-mer -> slot of table 1..4: bit fields of one 24-bit multiplicative hash (a second one for the rarely used last table)
constexpr unsigned HASH_C = 0x9E3779u, HASH_C4 = 0x85EBCBu;

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

// LDS by byte address: the compiler then sees plain integer arithmetic (no symbolic base to add per access) and folds
// constant steps into the instructions' offset field
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
typedef __attribute__((address_space(3))) const double lds_cf64;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(1))) double gbl_f64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u32x4 gbl_u128;
__device__ __forceinline__ unsigned lds_read_u16(unsigned addr) {
	return *reinterpret_cast<lds_cu16 *>(addr);
}
// (volatile: stays where it is written relative to the other volatile statements around it -- used to get a batch of reads
// under way BEFORE a stretch of scalar work the compiler would otherwise leave them behind)
__device__ __forceinline__ unsigned lds_read_u16_early(unsigned addr) {
	return *reinterpret_cast<volatile lds_cu16 *>(addr);
}
// The same as the bare instruction (a volatile 16-bit load is followed by an `and 0xffff` the zero-extending read does not need).
// The compiler does not count this read: whoever uses the value issues lds_wait_all() on it first.
template <int OFF>
__device__ __forceinline__ unsigned lds_read_u16_bare(unsigned addr) {
	unsigned v;
	asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
	return v;
}
template <int... I, class F>
__device__ __forceinline__ void for_each_int(std::integer_sequence<int, I...>, F &&f) {	// f(0_c), f(1_c), ...: a loop whose index is a constant expression
	(f(std::integral_constant<int, I>()), ...);
}
template <int N>
__device__ __forceinline__ void lds_wait_all(unsigned (&a)[N]) {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
	for (int i = 0; i < N; i++)
		asm volatile("" : "+v"(a[i]));	// (volatile statements keep their order: no use of a[i] moves above the wait)
}
__device__ __forceinline__ double lds_read_f64(unsigned addr) {
	return *reinterpret_cast<lds_cf64 *>(addr);
}
__device__ __forceinline__ void lds_write_u8(unsigned addr, unsigned v) {
	*reinterpret_cast<lds_u8 *>(addr) = (uint8_t) v;
}
// The merged rows are written once and read, if at all, by another kernel much later: nontemporal stores (the `nt` bit: no
// claim on the caches the next pairs' reads go through).  Config 1 +1.3 %, config 4 +0.6 % at full size; the same hint on the
// loads of the reads measured as nothing.
template <class T, class P>
__device__ __forceinline__ void store_stream(P *at, T v) {
	__builtin_nontemporal_store(v, at);
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
// the smallest / the middle one of three (one instruction each)
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) {
	unsigned r;
	asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
	unsigned r;
	asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ unsigned lds_read_u32(unsigned addr) {
	return *reinterpret_cast<__attribute__((address_space(3))) const unsigned *>(addr);
}
__device__ __forceinline__ unsigned lds_read_u32_early(unsigned addr) {	// (volatile: see lds_read_u16_early)
	return *reinterpret_cast<volatile __attribute__((address_space(3))) const unsigned *>(addr);
}
#ifndef FK_DEEP_BATCH
#define FK_DEEP_BATCH 1
#endif
// FK_BUCKETS: the first k-mer table as buckets of two slots (the same 4 KB: 2^(T1B-1) buckets): a word goes to its bucket's first
// slot, what that displaces to the second (the same address + 4), what that displaces to the second table.  A bucket keeps the two
// smallest of its words, in order; 250 k-mers overflow 512 buckets of two by 7 words where they overflow 1 024 single slots by 27,
// and the second table's 256 slots then pass on 0.1 words a pair instead of 1.3: the last two tables (two more inserts with a round
// trip each, eight more lookups and fifty-six vector instructions for every lane) are reached by a pair in ten instead of two in
// three, for one more insert level (no address arithmetic: an offset) and a 64-bit lookup where a 32-bit one was.
// 1: the geometries of four and five k-mers per lane; 2: three per lane (NARROW) too
#ifndef FK_BUCKETS
#define FK_BUCKETS 1
#endif
#ifndef FK_LAZY_FLAGS
#define FK_LAZY_FLAGS 1
#endif
#ifndef FK_KARGS
#define FK_KARGS 1
#endif
// FK_EPOCH (short geometry): the k-mer tables are cleared every 127th pair instead of every pair.  A table word is
// epoch (7 bits) | k-mer (16) | position (9); a wave's epoch counts DOWN from 126, so the words of the pair at hand are smaller
// than everything earlier pairs left behind: a stale word loses every ds_min (it is displaced like an empty slot's 0xFFFFFFFF, and
// what is displaced is dropped: "live" is one compare against the epoch's upper limit, where it was one against EMPTY), and it
// never reads as a match (its epoch bits differ).  After epoch 0, or after something else was laid over the tables, they are
// cleared and the count starts over.
#ifndef FK_EPOCH
#define FK_EPOCH 1
#endif

// ---- stage one read: four bases per lane -------------------------------------------------------------------
// Writes the row, its 2-bit codes and its N map to LDS; returns the B-cliff length (read length minus the trailing
// run of raw quality 2, assembler.c:176-177).  codes: base k of this lane at bits 2k..2k+1.  nmap: bit 2k (forward) or
// 2k+1 (reverse) set when base k is N -- or does not exist: past the end of a read everything reads as N, which is
// what keeps k-mer windows inside the read.  weird: non-zero when one of this lane's bases is outside the lean domain.
// The reverse stream stores every code with its two bits swapped (C=2, G=1) and its N flags on the odd bits: a
// bit-reversed window of it then reads as forward codes in reversed base order, which is how a reverse k-mer meets a
// forward one (base t against base 7-t) and how the scan pairs forward position fi with reverse position ri.
// FK_LEN_MASKS: what stage() needs of a read's length alone -- which of the lane's four bases exist: their bytes, their code
// bits -- stays in registers from pair to pair and is made again only when the length changes (eight vector instructions a read)
#ifndef FK_LEN_MASKS
#define FK_LEN_MASKS 1
#endif
struct LenMasks {
	int len;	// (-1: not made yet)
	unsigned bm, in2;
};
__device__ __forceinline__ void len_masks(LenMasks &m, int len, int lane) {
	int cnt = len - 4 * lane;
	cnt = cnt < 0 ? 0 : (cnt > 4 ? 4 : cnt);	// how many of this lane's bases exist
	m.bm = (unsigned) ((0xFFFFFFFFull << (8 * cnt)) >> 32);	// their bytes
	m.in2 = (1u << (2 * cnt)) - 1u;	// their code bits
	m.len = len;
}
template <bool FORWARD>
__device__ __forceinline__ int stage(const uint2 v, const int len, int lane, unsigned char *wb, unsigned &codes, unsigned &nmap, unsigned &weird, const LenMasks &lm) {
	reinterpret_cast<uint2 *>(wb + (FORWARD ? FK_F : FK_R))[lane] = v;
	const unsigned n4 = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);	// the four nt bytes
	const unsigned q4 = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);	// the four quality bytes
	const unsigned sel = n4 & 0x07070707u;
	const unsigned b3 = (n4 >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;	// 0xFF where nt >= 8
	const unsigned lo = __builtin_amdgcn_perm(lut_word<FORWARD>(1), lut_word<FORWARD>(0), sel);
	const unsigned hi = __builtin_amdgcn_perm(lut_word<FORWARD>(3), lut_word<FORWARD>(2), sel);
	const unsigned r4 = (hi & m3) | (lo & ~m3);
#if FK_LEN_MASKS
	const unsigned bm = lm.bm, in2 = lm.in2;
#else
	int cnt = len - 4 * lane;
	cnt = cnt < 0 ? 0 : (cnt > 4 ? 4 : cnt);	// how many of this lane's bases exist
	const unsigned bm = (unsigned) ((0xFFFFFFFFull << (8 * cnt)) >> 32);	// their bytes
	const unsigned in2 = (1u << (2 * cnt)) - 1u;
#endif
	const unsigned wq = (((q4 & 0x7F7F7F7Fu) + 0x51515151u) | q4) & 0x80808080u;	// a quality outside 0..46
	weird |= ((r4 & 0x08080808u) | wq | (n4 & 0xF0F0F0F0u)) & bm;
	const unsigned c4 = r4 & 0x03030303u, isn4 = (r4 >> 2) & 0x01010101u;
	const unsigned cy = c4 | (c4 >> 6), ny = isn4 | (isn4 >> 6);
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
	int have = len - 4 * lane;
	have = have < 0 ? 0 : (have > 4 ? 4 : have);
	const unsigned not2 = ~is2 & ((1u << have) - 1u);
	const int mine = not2 ? lane * 4 + (31 - __builtin_clz(not2)) + 1 : 0;
	const u64 has = ballot(not2 != 0);
	if (has == 0)
		return 0;
	return __builtin_amdgcn_readlane(mine, 63 - __builtin_clzll(has));
}

// ---- LONG geometry: stage one read, eight bases per lane ------------------------------------------------------
// Four bases of a row (x: nt0 q0 nt1 q1, y: nt2 q2 nt3 q3), of which the first `cnt` exist: their k-mer codes (2 bits each),
// N flags (bit 2k forward, 2k+1 reverse; a base that does not exist reads as N), whether one is outside the lean domain;
// the nt and the quality bytes for the caller.  The arithmetic of stage() above.
// (bm: the bytes of the bases that exist, in2: their code bits -- len_masks())
template <bool FORWARD>
__device__ __forceinline__ void pack4m(unsigned x, unsigned y, unsigned bm, unsigned in2, unsigned &codes, unsigned &nmap, unsigned &weird, unsigned &skip4, unsigned &q4) {
	unsigned n4;
	n4 = __builtin_amdgcn_perm(y, x, 0x06040200u);
	q4 = __builtin_amdgcn_perm(y, x, 0x07050301u);
	const unsigned sel = n4 & 0x07070707u;
	const unsigned b3 = (n4 >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;
	const unsigned lo = __builtin_amdgcn_perm(lut_word<FORWARD>(1), lut_word<FORWARD>(0), sel);
	const unsigned hi = __builtin_amdgcn_perm(lut_word<FORWARD>(3), lut_word<FORWARD>(2), sel);
	const unsigned r4 = (hi & m3) | (lo & ~m3);
	const unsigned wq = (((q4 & 0x7F7F7F7Fu) + 0x51515151u) | q4) & 0x80808080u;
	weird |= ((r4 & 0x08080808u) | wq | (n4 & 0xF0F0F0F0u)) & bm;
	const unsigned c4 = r4 & 0x03030303u, isn4 = (r4 >> 2) & 0x01010101u;
	skip4 = isn4 | ~bm;	// (bit 0 of a byte: an N, or no base at all)
	const unsigned cy = c4 | (c4 >> 6), ny = isn4 | (isn4 >> 6);
	codes = (cy | (cy >> 12)) & in2;
	const unsigned nb = ((ny | (ny >> 12)) | ~in2) & 0x55u;
	nmap = FORWARD ? nb : nb << 1;
}
template <bool FORWARD>
__device__ __forceinline__ void pack4(unsigned x, unsigned y, int cnt, unsigned &codes, unsigned &nmap, unsigned &weird, unsigned &skip4, unsigned &q4) {
	pack4m<FORWARD>(x, y, (unsigned) ((0xFFFFFFFFull << (8 * cnt)) >> 32), (1u << (2 * cnt)) - 1u, codes, nmap, weird, skip4, q4);
}

// which of four bases count in the primer scan's filter (bit 2k+1): a definite base of quality >= qthr that exists; an N,
// a low quality or the end of the read agrees with anything there
__device__ __forceinline__ unsigned care4(unsigned skip4, unsigned q4, unsigned qadd) {
	const unsigned ge = ((q4 & 0x7F7F7F7Fu) + qadd) >> 7;	// bit 0 of a byte: quality >= qthr (qadd = (0x80 - qthr) in every byte)
	const unsigned f = ge & ~skip4 & 0x01010101u;
	const unsigned y = f | (f >> 6);
	return ((y | (y >> 12)) & 0x55u) << 1;
}

__device__ __forceinline__ unsigned is2_mask(unsigned q4) {	// bit k: quality byte k == 2
	const unsigned x = q4 ^ 0x02020202u;
	const unsigned t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);	// 0x80 where quality == 2
	return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// (FK_LEN_MASKS, eight bases per lane: the two halves' masks)
struct LenMasks8 {
	int len;
	unsigned bm0, in0, bm1, in1;
};
__device__ __forceinline__ void len_masks8(LenMasks8 &m, int len, int lane) {
	int c0 = len - 8 * lane;
	int c1 = c0 - 4;
	c0 = c0 < 0 ? 0 : (c0 > 4 ? 4 : c0);
	c1 = c1 < 0 ? 0 : (c1 > 4 ? 4 : c1);
	m.bm0 = (unsigned) ((0xFFFFFFFFull << (8 * c0)) >> 32);
	m.in0 = (1u << (2 * c0)) - 1u;
	m.bm1 = (unsigned) ((0xFFFFFFFFull << (8 * c1)) >> 32);
	m.in1 = (1u << (2 * c1)) - 1u;
	m.len = len;
}
// Writes the row, its code stream and N map (and, PRIM, its 4-bit nt stream) to LDS; returns the B-cliff length.
template <bool FORWARD, bool PRIM>
__device__ __forceinline__ int stage8(const uint4 v, const int len, int lane, unsigned char *wb, unsigned &weird, unsigned qadd, const LenMasks8 &lm) {
	typedef Lay<true> L;
	if (lane < 40)
		reinterpret_cast<uint4 *>(wb + (FORWARD ? L::F : L::R))[lane] = v;
	unsigned ca, na, n4a, q4a, cb, nb, n4b, q4b;
	pack4m<FORWARD>(v.x, v.y, lm.bm0, lm.in0, ca, na, weird, n4a, q4a);
	pack4m<FORWARD>(v.z, v.w, lm.bm1, lm.in1, cb, nb, weird, n4b, q4b);
	if (lane < 48) {	// (lanes past the row: all N)
		reinterpret_cast<uint16_t *>(wb + (FORWARD ? L::FC : L::RC))[lane] = (uint16_t) (ca | (cb << 8));
		reinterpret_cast<uint16_t *>(wb + (FORWARD ? L::FN : L::RN))[lane] = (uint16_t) (na | (nb << 8));
		if (PRIM)
			reinterpret_cast<uint16_t *>(wb + (FORWARD ? L::CAREF : L::CARER))[lane] = (uint16_t) (care4(n4a, q4a, qadd) | (care4(n4b, q4b, qadd) << 8));
	}
	// the usual read does not end in quality 2: two lane reads settle it
	const int li = (len - 1) >> 3, lb = (len - 1) & 7;
	const unsigned qa_l = (unsigned) __builtin_amdgcn_readlane((int) q4a, li), qb_l = (unsigned) __builtin_amdgcn_readlane((int) q4b, li);
	if ((((lb < 4 ? qa_l : qb_l) >> (8 * (lb & 3))) & 0xFFu) != 2u)
		return len;
	int c8 = len - 8 * lane;
	c8 = c8 < 0 ? 0 : (c8 > 8 ? 8 : c8);
	const unsigned not2 = ~(is2_mask(q4a) | (is2_mask(q4b) << 4)) & ((1u << c8) - 1u);
	const int mine = not2 ? lane * 8 + (31 - __builtin_clz(not2)) + 1 : 0;
	const u64 has = ballot(not2 != 0);
	if (has == 0)
		return 0;
	return __builtin_amdgcn_readlane(mine, 63 - __builtin_clzll(has));
}

// ---- LONG geometry: the primer scan (assembler.c:262-288 calling panda_compute_offset_qual, offset.c:47-112) ------------
// An alignment of the primer at read position s scores v(s) = (sum of its terms) / (s + plen + 1) -- a term is qual_score[q]
// where read and primer base agree, qual_score_err[q] where they do not, nothing for a primer N -- and the scan returns
// 1 + s + plen of the FIRST alignment whose value is the largest above plen * threshold (zero penalty: exp() is the last,
// monotone step of offset.c:72 and values are compared before it; a comparison so close that exp() could blur it sends the
// pair to the careful path).  Every term is <= 0, and a disagreement at quality >= qthr costs at least -err_t =
// -qual_score_err[qthr]: an alignment with more than K = floor(bound * len / err_t) of them cannot reach `bound`, whatever
// else it holds.  So: (1) lanes are ALIGNMENTS; each counts the disagreements of its first eight primer bases against a
// window of the read's 4-bit nt stream (low-quality bases already read as N there) -- a dozen instructions for 64 alignments;
// (2) the first alignment with at most one is summed exactly, and its value (or the threshold) is the bound; (3) K follows;
// alignments with more than K disagreements -- among the first eight bases when K <= 1, else among all -- are out;
// (4) the few that are left are summed exactly, in the reference's order, and compared as the reference compares them.
// Returns -1 when it cannot settle the read that cheaply (no good alignment, a bound too loose to filter, too many left).

// the wave's largest value by DPP moves (a lane a move does not reach keeps its own value): no ds_bpermute round trips
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64_or_self(double v) {
	union {
		double d;
		int i[2];
	} x, y;
	x.d = v;
	y.i[0] = __builtin_amdgcn_update_dpp(x.i[0], x.i[0], CTRL, ROWS, 0xF, false);
	y.i[1] = __builtin_amdgcn_update_dpp(x.i[1], x.i[1], CTRL, ROWS, 0xF, false);
	return y.d;
}
__device__ __forceinline__ double max_f64(double a, double b) {	// v_max_f64 itself (no NaNs where it is used)
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ double wave_max_f64(double v) {
	v = max_f64(v, dpp_f64_or_self<0xB1, 0xF>(v));
	v = max_f64(v, dpp_f64_or_self<0x4E, 0xF>(v));
	v = max_f64(v, dpp_f64_or_self<0x141, 0xF>(v));
	v = max_f64(v, dpp_f64_or_self<0x140, 0xF>(v));
	v = max_f64(v, dpp_f64_or_self<0x142, 0xA>(v));
	v = max_f64(v, dpp_f64_or_self<0x143, 0xC>(v));
	return lane_f64(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {	// likewise, integers
#define FK_STEP(CTRL, ROWS)                                                                 \
	{                                                                                   \
		const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xF, false);    \
		v = o > v ? o : v;                                                          \
	}
	FK_STEP(0xB1, 0xF) FK_STEP(0x4E, 0xF) FK_STEP(0x141, 0xF) FK_STEP(0x140, 0xF) FK_STEP(0x142, 0xA) FK_STEP(0x143, 0xC)
#undef FK_STEP
	return __builtin_amdgcn_readlane(v, 63);
}
// among the lanes in `holders`, the smallest of their values (usually one lane)
__device__ __forceinline__ int smallest_of(u64 holders, int value) {
	int w = 0x7FFFFFFF;
	while (holders != 0) {
		const int l = __builtin_ctzll(holders);
		holders &= holders - 1;
		const int x = __builtin_amdgcn_readlane(value, l);
		w = x < w ? x : w;
	}
	return w;
}

// the sum of one alignment's terms in primer order (offset.c:79-87): lanes look the terms up, lane 0 adds them
__device__ __forceinline__ double primer_exact(const double *ltab, unsigned char *wb, unsigned row_off, int s, unsigned pbase, int plen, int lane) {
	typedef Lay<true> L;
	double t = 0.0;
	if (lane < plen) {
		const unsigned e = reinterpret_cast<const uint16_t *>(wb + row_off)[s + lane];
		const double v = ltab[(((e & pbase & 0xFu) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + (e >> 8)];
		t = pbase != 15u ? v : 0.0;	// (a primer N adds nothing; + 0.0 leaves the sum as it is)
	}
	double *scr = reinterpret_cast<double *>(wb + L::SCR);
	wave_sync();
	scr[lane] = t;
	wave_sync();
	double acc = 0.0;
	if (lane == 0) {
		for (int c = 0; c < plen; c += 8) {
			const double2 v0 = *reinterpret_cast<const double2 *>(scr + c), v1 = *reinterpret_cast<const double2 *>(scr + c + 2);
			const double2 v2 = *reinterpret_cast<const double2 *>(scr + c + 4), v3 = *reinterpret_cast<const double2 *>(scr + c + 6);
			acc += v0.x;
			acc += v0.y;
			acc += v1.x;
			acc += v1.y;
			acc += v2.x;
			acc += v2.y;
			acc += v3.x;
			acc += v3.y;
		}
	}
	return uni_f64(acc);
}

// ---- the usual read, five alignments per lane ----
// disagreements that count among the primer's first sixteen bases for this lane's alignments s = 5 * lane + j, five bits each
// (31: no such alignment); s0: the first alignment with at most one (-1: none)
__device__ __forceinline__ unsigned primer_tier1(unsigned code_addr, unsigned care_addr, int nal, int lane, unsigned pc0, unsigned pk0, int &s0) {
	const unsigned bit = 10u * (unsigned) lane;
	const unsigned o = 4u * (bit >> 5), sh = bit & 31u;
	const unsigned c0 = lds_read_u32(code_addr + o), c1 = lds_read_u32(code_addr + o + 4u), c2 = lds_read_u32(code_addr + o + 8u);
	const unsigned k0 = lds_read_u32(care_addr + o), k1 = lds_read_u32(care_addr + o + 4u), k2 = lds_read_u32(care_addr + o + 8u);
	const unsigned x0 = __builtin_amdgcn_alignbit(c1, c0, sh), x1 = __builtin_amdgcn_alignbit(c2, c1, sh);
	const unsigned y0 = __builtin_amdgcn_alignbit(k1, k0, sh), y1 = __builtin_amdgcn_alignbit(k2, k1, sh);
	unsigned packed = 0;
	int firstj = 7;
#pragma unroll
	for (int j = 4; j >= 0; j--) {
		const unsigned w = j == 0 ? x0 : __builtin_amdgcn_alignbit(x1, x0, 2u * j), kk = j == 0 ? y0 : __builtin_amdgcn_alignbit(y1, y0, 2u * j);
		const unsigned t = w ^ pc0;
		const int mm = __popc((t | (t << 1)) & kk & pk0);
		packed = (packed << 5) | (unsigned) mm;
		firstj = mm <= 1 ? j : firstj;
	}
	// the alignments this lane has: j < lim (all five but in the lane the read ends in, none above it)
	int lim = nal - 5 * lane;
	lim = lim < 0 ? 0 : (lim > 5 ? 5 : lim);
	packed |= (0x1FFFFFFu << (5 * lim)) & 0x1FFFFFFu;
	firstj = firstj < lim ? firstj : 7;
	const u64 any = ballot(firstj != 7);
	s0 = -1;
	if (any != 0) {
		const int l = __builtin_ctzll(any);
		s0 = 5 * l + __builtin_amdgcn_readlane(firstj, l);
	}
	return packed;
}

// (measured and not kept, round 6: a pass BEFORE the counts that only asks whether an alignment has a disagreement at all -- xor,
// a three-input and, a compare into a scalar pair: three instructions where the count takes nine -- and settles the read when
// exactly one alignment has none and its value leaves every other alignment a budget of zero.  On config 2's reads a primer base
// of quality 2 or two of quality 8 -- a read in three -- leave late alignments a budget of one or two and the pass is wasted:
// 22 vector instructions fewer per pair of 1 236, 125 scalar ones more of 595 (the five masks' validity, counts and index):
// -2.6 %, profiles/r06_experiments_ab.txt.  Scalar work is not free at four waves a SIMD: 0.6-0.8 cycles an instruction beside a
// 4.2-cycle vector one, and a dependent chain of it is latency the wave's vector work queues behind.)
// The sums of one alignment of each read at once, in ANY order (|error| < 1e-12: good for a bound and for a verdict with
// a margin; values that are compared with each other are summed in the reference's order, primer_exact): lanes 0..31 look up
// the terms of the forward read's alignment s0f (lane x: primer base x), lanes 32..63 those of the reverse read's s0r; no
// LDS round trip but the two lookups
__device__ __forceinline__ void primer_sums_any_order(const double *ltab, unsigned char *wb, int s0f, int s0r, int fplen, int rplen, unsigned pbj, int lane, double &sumf, double &sumr) {
	typedef Lay<true> L;
	const int half = lane >> 5, x = lane & 31;
	const int s0 = half ? s0r : s0f, plen = half ? rplen : fplen;
	double t = 0.0;
	if (x < plen && s0 >= 0) {
		const unsigned e = reinterpret_cast<const uint16_t *>(wb + (half ? L::R : L::F))[s0 + x];
		const double v = ltab[(((e & pbj & 0xFu) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + (e >> 8)];
		t = pbj != 15u ? v : 0.0;
	}
	t += dpp_f64<0xB1>(t);	// quad_perm:[1,0,3,2]
	t += dpp_f64<0x4E>(t);	// quad_perm:[2,3,0,1]
	t += dpp_f64<0x141>(t);	// row_half_mirror
	t += dpp_f64<0x140>(t);	// row_mirror: every lane holds its row's sum
	sumf = lane_f64(t, 0) + lane_f64(t, 16);
	sumr = lane_f64(t, 32) + lane_f64(t, 48);
}

// With the alignment s0 summed: the alignments that could still reach its value -- none, usually -- are summed as well and
// compared as offset.c:72-77 compares them (before exp).  >= 0: what the scan returns; -1: more work (primer_settle)
__device__ __forceinline__ int primer_shortcut(const double *ltab, unsigned char *wb, unsigned row_off, unsigned pbase, unsigned packed, int s0, double sum0, int len, int plen, double thr, double inv_err_t, int lane) {
	const int n0 = s0 + plen + 1;
	const double kv = sum0 * (double) __builtin_amdgcn_rcpf((float) n0) * (1.0 + 1e-5);	// a shade below its value (the reciprocal is good to 1e-7): a bound may be anything an alignment reaches
	const double bound = kv > thr ? kv : thr;
	// alignment s may hold K(s) = floor((bound * (s + plen + 1) - 1e-6) / err_t) disagreements and still reach the bound
	// (taken at the lane's last alignment for all five: a budget may be too large, never too small)
	const double xs = bound * inv_err_t * 1.0001, x0 = (double) (plen + 1) * xs - 1e-6 * inv_err_t + 0.001;	// all >= 0
	if (!((double) (len - plen - 1) * xs + x0 < 8.0))
		return -1;
	const int K = (int) ((double) (5 * lane + 4) * xs + x0);
	u64 l0 = ballot((int) (packed & 31u) <= K), l1 = ballot((int) ((packed >> 5) & 31u) <= K), l2 = ballot((int) ((packed >> 10) & 31u) <= K);
	u64 l3 = ballot((int) ((packed >> 15) & 31u) <= K), l4 = ballot((int) ((packed >> 20) & 31u) <= K);
	const int nleft = __popcll(l0) + __popcll(l1) + __popcll(l2) + __popcll(l3) + __popcll(l4);
	// the usual read: only the alignment already summed is left.  (It is among what is left whenever the bound is its own value
	// -- it reaches the bound it set --; when the bound is the threshold, which it does not reach, the one left is another
	// alignment: summed below like any other.)
	if (nleft == 1 && kv > thr) {
		if (sum0 > (thr + 1e-9) * (double) n0)
			return n0;
		if (sum0 < (thr - 1e-9) * (double) n0)
			return 0;
		return -1;
	}
	if (nleft > 4)
		return -1;
	double best = thr;
	int besti = 0;
	bool close = false;
	for (int it = 0; it < nleft; it++) {	// in ascending order
		const int c0 = l0 != 0 ? 5 * __builtin_ctzll(l0) : 0x7FFFFFF0, c1 = l1 != 0 ? 5 * __builtin_ctzll(l1) + 1 : 0x7FFFFFF0, c2 = l2 != 0 ? 5 * __builtin_ctzll(l2) + 2 : 0x7FFFFFF0;
		const int c3 = l3 != 0 ? 5 * __builtin_ctzll(l3) + 3 : 0x7FFFFFF0, c4 = l4 != 0 ? 5 * __builtin_ctzll(l4) + 4 : 0x7FFFFFF0;
		const int m01 = c0 < c1 ? c0 : c1, m23 = c2 < c3 ? c2 : c3, m03 = m01 < m23 ? m01 : m23, s = m03 < c4 ? m03 : c4;
		if (s == c0)
			l0 &= l0 - 1;
		else if (s == c1)
			l1 &= l1 - 1;
		else if (s == c2)
			l2 &= l2 - 1;
		else if (s == c3)
			l3 &= l3 - 1;
		else
			l4 &= l4 - 1;
		const double sum = primer_exact(ltab, wb, row_off, s, pbase, plen, lane);
		const double v = sum / (double) (s + plen + 1);
		if (v > best) {	// strict: the earliest index wins ties (offset.c:74)
			close |= v - best <= 4.5e-16;	// their exponentials might not differ: the reference would keep the earlier one
			best = v;
			besti = s + plen + 1;
		}
	}
	return close ? -1 : besti;
}

// ---- every other read: exact sums with lanes as alignments ----
// One pass: lane sums alignment s (if it has one) in primer order (offset.c:79-87), four primer bases at a time; the pass is
// given up once no lane can reach `bound` any more (every term is <= 0).  false: given up
__device__ __forceinline__ bool primer_lockstep(const double *ltab, const unsigned char *wb, unsigned row_off, int s, bool have, unsigned pbase, int plen, double bound, double &v) {
	const uint16_t *S = reinterpret_cast<const uint16_t *>(wb + row_off) + (have ? s : 0);
	const double floor_acc = bound * (double) (s + plen + 1) - 1e-6;
	double acc = 0.0;
	int x = 0;
	for (; x + 4 <= plen; x += 4) {
		unsigned e[4], pn[4];
		double t[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			e[u] = S[x + u];
			pn[u] = (unsigned) __builtin_amdgcn_readlane((int) pbase, x + u);
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
			t[u] = ltab[(((e[u] & pn[u] & 0xFu) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + (e[u] >> 8)];
#pragma unroll
		for (int u = 0; u < 4; u++)
			if (pn[u] != 15u)
				acc += t[u];
		if (ballot(have && !(acc < floor_acc)) == 0)
			return false;
	}
	for (; x < plen; x++) {
		const unsigned pn = (unsigned) __builtin_amdgcn_readlane((int) pbase, x);
		if (pn != 15u) {
			const unsigned e = S[x];
			acc += ltab[(((e & pn & 0xFu) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + (e >> 8)];
		}
	}
	v = acc / (double) (s + plen + 1);
	return true;
}

// packed: primer_tier1's counts for this lane's alignments 5 lane + j; thr: plen * threshold (offset.c:59, before exp);
// s0, sum0: an alignment already summed, in any order: good for the bound only (s0 < 0: none).
// An alignment's value divides by its index, so with a disagreement or two in the primer proper the winner is usually some
// LATE alignment: when many alignments could still reach the bound, the last 64 are summed first and the best of them is
// the bound for the rest -- few of which are then left; those are gathered into a list and summed 64 at a time.
__device__ __forceinline__ int primer_settle(const double *ltab, unsigned char *wb, unsigned row_off, int len, unsigned pbase, unsigned packed, int plen, double thr, double inv_err_t, int lane, int s0, double sum0) {
	typedef Lay<true> L;
	const int nal = len - plen;	// alignments 0 .. nal-1 are judged (at index s + plen <= len - 1, offset.c:70-77)
	// the bound: a value some alignment reaches, or the threshold every alignment has to beat
	double bound = thr;
	if (s0 >= 0) {
		const double kv = sum0 * (double) __builtin_amdgcn_rcpf((float) (s0 + plen + 1)) * (1.0 + 1e-5);
		bound = kv > thr ? kv : thr;
	}
	// lane l keeps the best of the alignments it has summed, the first of equals; the lanes are compared at the end.
	// Two values may differ and their exponentials not: the reference then keeps the earlier index (the careful path's case)
	double best = thr;
	int besti = 0;
	bool close = false;
	auto consider = [&](double v, int s, bool mine) {
		const int idx = s + plen + 1;
		const bool first = besti == 0;
		const bool better = mine && (v > best || (!first && v == best && idx < besti));	// strict against the threshold (offset.c:74)
		close |= mine && v != best && fabs(v - best) <= 4.5e-16 && (!first || v > best);
		if (better) {
			best = v;
			besti = idx;
		}
	};
	auto raise = [&]() {	// a better bound for what is still to be summed
		const double m = wave_max_f64(besti != 0 ? best : bound);
		bound = m > bound ? m : bound;
	};
	// Alignment s may hold K(s) = floor((bound * (s + plen + 1) - 1e-6) / err_t) disagreements that count and still reach the
	// bound (taken at the lane's last alignment for all five: a budget may be too large, never too small; the counts are
	// those of the primer's first sixteen bases: never too many).  Bit j: alignment 5 lane + j, below `top`, is left
	auto filter = [&](int top) -> unsigned {
		const double xs = bound * inv_err_t * 1.0001, x0 = (double) (plen + 1) * xs - 1e-6 * inv_err_t + 0.001;	// all >= 0
		int K = (int) ((double) (5 * lane + 4) * xs + x0);
		K = K > 30 ? 30 : K;	// (31: no such alignment)
		unsigned m = 0;
#pragma unroll
		for (int j = 0; j < 5; j++)
			m |= ((int) ((packed >> (5 * j)) & 31u) <= K && 5 * lane + j < top) ? 1u << j : 0u;
		return m;
	};
	unsigned surv = filter(nal);
	int n = 0;
#pragma unroll
	for (int j = 0; j < 5; j++)
		n += __popcll(ballot(((surv >> j) & 1u) != 0));
	if (n > 64) {
		const int top = nal - 64;
		const int s = top + lane;
		double v = 0.0;
		if (primer_lockstep(ltab, wb, row_off, s, s >= 0, pbase, plen, bound, v))
			consider(v, s, s >= 0);
		raise();
		surv = filter(top);
	}
	uint16_t *const list = reinterpret_cast<uint16_t *>(wb + L::SCR);
	int count = 0;
#pragma unroll
	for (int j = 0; j < 5; j++) {
		const bool left = ((surv >> j) & 1u) != 0;
		const u64 m = ballot(left);
		if (left)
			list[count + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u))] = (uint16_t) (5 * lane + j);
		count += __popcll(m);
	}
	wave_sync();
	for (int p0 = 0; p0 < count; p0 += 64) {
		const bool have = p0 + lane < count;
		const int s = have ? (int) list[p0 + lane] : 0;
		double v = 0.0;
		if (primer_lockstep(ltab, wb, row_off, s, have, pbase, plen, bound, v))
			consider(v, s, have);
		if (p0 + 64 < count)
			raise();
	}
	// first maximum across lanes: the larger value, the smaller index among equals
	const u64 got = ballot(besti != 0);
	if ((got & (got - 1)) == 0) {	// nobody, or one lane only
		if (ballot(close) != 0)
			return -1;
		return got != 0 ? __builtin_amdgcn_readlane(besti, __builtin_ctzll(got)) : 0;
	}
	// (the wave's largest value by DPP; a lane whose value differs from it by less than exp() can tell apart sends the read to
	// the careful path -- only comparisons with the winner matter: whatever else lies close together loses to it either way --;
	// among the lanes that hold it the smallest index)
	const double top = wave_max_f64(besti != 0 ? best : -__builtin_inf());
	close |= besti != 0 && best != top && fabs(best - top) <= 4.5e-16;
	if (ballot(close) != 0)
		return -1;
	u64 holders = ballot(besti != 0 && best == top);
	int winner = 0x7FFFFFFF;
	while (holders != 0) {
		const int l = __builtin_ctzll(holders);
		holders &= holders - 1;
		const int bi = __builtin_amdgcn_readlane(besti, l);
		winner = bi < winner ? bi : winner;
	}
	return winner;
}

// ---- `-a`: the primers are looked for in the MERGED sequence (assembler.c:300-333, offset.c:114-133) ----------------------
// A merged base that agrees with the primer base adds its log-probability p, one that does not adds panda_log1mexp(p) -- which,
// written for positive arguments and handed p <= 0, is NaN (-inf for p = 0): such an alignment's value is NaN and offset.c:72
// never prefers it.  (pamd_fast_supports() admits `-a` only when the host has found that so for every entry of the table
// image.)  So only alignments WITHOUT a disagreeing base can be returned: lanes are alignments again (primer_tier1 on the
// merged sequence's own code stream), the ones without a disagreement among the primer's first sixteen bases are nearly always
// one -- the primer itself -- or none, and that one is checked over the primer's whole length and summed.  Two or more send the
// pair to the general kernel.

// four merged bases, one nt byte each, of which the first `cnt` exist: 2-bit codes and "counts" flags (bit 2k+1: a definite
// base; a merged N agrees with every primer base)
__device__ __forceinline__ void pack_nt4(unsigned n4, int cnt, unsigned &codes, unsigned &care) {
	const unsigned sel = n4 & 0x07070707u;
	const unsigned b3 = (n4 >> 3) & 0x01010101u, m3 = (b3 << 8) - b3;
	const unsigned lo = __builtin_amdgcn_perm(lut_word<true>(1), lut_word<true>(0), sel);
	const unsigned hi = __builtin_amdgcn_perm(lut_word<true>(3), lut_word<true>(2), sel);
	const unsigned r4 = (hi & m3) | (lo & ~m3);
	const unsigned bm = (unsigned) ((0xFFFFFFFFull << (8 * cnt)) >> 32);
	const unsigned c4 = r4 & 0x03030303u & bm, def4 = ~(r4 >> 2) & 0x01010101u & bm;
	const unsigned cy = c4 | (c4 >> 6), dy = def4 | (def4 >> 6);
	codes = (cy | (cy >> 12)) & 0xFFu;
	care = ((dy | (dy >> 12)) & 0x55u) << 1;
}

// the merged row's code stream and "counts" map from its staged nt bytes: eight bases per lane, 512 per pass
__device__ __forceinline__ void post_streams(unsigned char *wb, int seqlen, int lane) {
	typedef Lay<true> L;
#pragma unroll
	for (int pass = 0; pass < 2; pass++) {
		if (pass == 1 && seqlen <= 512)
			break;
		if (pass == 0 || lane < 16) {
			const int b0 = 512 * pass + 8 * lane;
			const uint2 v = *reinterpret_cast<const uint2 *>(wb + L::NT + b0);
			int c0 = seqlen - b0, c1 = c0 - 4;
			c0 = c0 < 0 ? 0 : (c0 > 4 ? 4 : c0);
			c1 = c1 < 0 ? 0 : (c1 > 4 ? 4 : c1);
			unsigned ca, ka, cb, kb;
			pack_nt4(v.x, c0, ca, ka);
			pack_nt4(v.y, c1, cb, kb);
			reinterpret_cast<uint16_t *>(wb + L::MC)[64 * pass + lane] = (uint16_t) (ca | (cb << 8));
			reinterpret_cast<uint16_t *>(wb + L::MK)[64 * pass + lane] = (uint16_t) (ka | (kb << 8));
		}
	}
}

// The alignments t = 0 .. nal-1 of a primer (codes pc0, counts map pk0: its first sixteen bases) on the merged sequence that
// hold no disagreement there.  Returns their number (2 = two or more) and, for one, which.  skip0: alignment 0 is not one
// (the reverse primer's alignments are counted from the sequence's end: offset.c:70-77 never judges the one that ends there)
__device__ __forceinline__ int post_candidates(unsigned char *wb, int nal, unsigned pc0, unsigned pk0, bool skip0, int lane, int &which) {
	typedef Lay<true> L;
	int found = 0;
	which = -1;
#pragma unroll
	for (int pass = 0; pass < 2; pass++) {
		const int n = nal - 320 * pass;
		if (n <= 0)
			break;
		int s0;
		const unsigned packed = primer_tier1(lds_addr(wb + L::MC) + 80u * pass, lds_addr(wb + L::MK) + 80u * pass, n, lane, pc0, pk0, s0);
		unsigned z = 0;
#pragma unroll
		for (int j = 0; j < 5; j++)
			z |= ((packed >> (5 * j)) & 31u) == 0 ? 1u << j : 0u;
		if (skip0 && pass == 0 && lane == 0)
			z &= ~1u;
		const u64 any = ballot(z != 0);
		if (any == 0)
			continue;
		const int l = __builtin_ctzll(any);
		const unsigned zl = (unsigned) __builtin_amdgcn_readlane((int) z, l);
		if ((any & (any - 1)) != 0 || (zl & (zl - 1u)) != 0 || found != 0)
			return 2;
		found = 1;
		which = 320 * pass + 5 * l + __builtin_ctz(zl);
	}
	return found;
}

// The alignment that starts at merged base s (reverse: at base s counted from the sequence's end, walking towards its start):
// every primer base that is not N against its merged base (offset.c:79-87); a disagreement anywhere makes the value NaN, never
// chosen: 0.  Else its value against the threshold (offset.c:59,72: both after exp(), which is monotone; a verdict closer than
// 1e-9 is left to the general kernel: -1), s + plen + 1 when it passes.  pb: this lane's primer base (lane x: base x; N past
// the primer's end)
__device__ __forceinline__ int post_verdict(const double *ltab, const unsigned char *wb, int s, bool reverse, int len, int plen, unsigned pb, double thr, int lane) {
	typedef Lay<true> L;
	const bool on = lane < plen && pb != 15u;
	const int at = reverse ? len - 1 - (s + lane) : s + lane;	// (inside the row for every lane below plen: s + plen <= len - 1)
	unsigned nt = 0, idx = PAMD_LT_ZERO;
	if (on) {
		nt = wb[L::NT + at];
		idx = reinterpret_cast<const uint16_t *>(wb + L::PIDX)[at];
	}
	if (ballot(on && (nt & pb) == 0) != 0)
		return 0;
	const double sum = wave_sum_any_order(ltab[idx]);	// (the spare slot of the error block holds 0.0)
	const double n0 = (double) (s + plen + 1);
	if (sum > (thr + 1e-9) * n0)
		return s + plen + 1;
	if (sum < (thr - 1e-9) * n0)
		return 0;
	return -1;
}

// counts over the 16 overlap positions i0 .. i0+15 of candidate `ov` (ov <= both read lengths, so every position
// below ov exists on both reads): unknowns (an N on either side) in bits 0..9, matching bases in bits 10..
// (algo_simple_bayes.c:40-59).  Forward bases flen-ov+i ascending; reverse bases rlen-1-i descending: the reverse words
// are taken at the mirrored position and bit-reversed.  Reads outside a stream stay inside the wave's LDS and are masked.
// FNO / RNO: where the N maps lie (the read's own, or the copies scan_many_sums makes, in which a base of low quality is an N too)
template <class L, unsigned FNO = L::FN, unsigned RNO = L::RN>
__device__ __forceinline__ unsigned word_counts(const unsigned char *wb, int flen, int rlen, int ov, int i0, bool active) {
	int hi = ov - i0;
	hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
	const unsigned valid = active ? ((unsigned) ((1ull << (2 * hi)) - 1ull) & 0x55555555u) : 0u;
	const int fb = 2 * (flen - ov + i0), rb = 2 * (rlen - 16 - i0);
	const uint32_t *fcw = reinterpret_cast<const uint32_t *>(wb + L::FC) + (fb >> 5), *fnw = reinterpret_cast<const uint32_t *>(wb + FNO) + (fb >> 5);
	const uint32_t *rcw = reinterpret_cast<const uint32_t *>(wb + L::RC) + (rb >> 5), *rnw = reinterpret_cast<const uint32_t *>(wb + RNO) + (rb >> 5);
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
		// (ds_bpermute butterflies here and below, not DPP moves: with those this function's register footprint went from 56 to
		// 128 and the hot path, which has to keep its values across the call, spilled: config 1 -7 %.  It runs for three pairs
		// in a hundred.)
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
			const unsigned cnt = word_counts<Lay<false>>(wb, flen, rlen, o, i0, i0 < o);
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
		const double sc = count_score<ALGO, true>(qual_nn, pmatch, pmismatch, count_table, flen, rlen, ov[k], (int) (acc[k] & 1023u), (int) (acc[k] >> 10), ov[k]);
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

// ---- pear / rdp_mle: one overlap's score, a sum of per-base table values (algo_pear.c:32-59, algo_rdp_mle.c:43-74) ------------
// This lane's share of the terms of candidate `ov`: positions lane, lane + 64, ...; added up in any order by the caller.
template <int ALGO, class L>
__device__ __forceinline__ double score_share(const unsigned char *wb, const double *ltab, int flen, int rlen, int ov, double nrb, double nn, int lane) {
	const unsigned lF = lds_addr(wb + L::F), lR = lds_addr(wb + L::R), lT = lds_addr(ltab);
	double part = 0.0;
#pragma unroll 1	/* (only the many-candidate scan calls this: it is cold, and its registers are what the hot path cannot keep across the call) */
	for (int r = 0; r < L::SR; r++) {
		const int i = 64 * r + lane;
		if (64 * r < ov && i < ov) {
			const int fi = flen - ov + i, ri = rlen - 1 - i;
			const unsigned ef = lds_read_u16(lF + 2u * (unsigned) fi), er = lds_read_u16(lR + 2u * (unsigned) ri);
			const unsigned a = ef & 0xFFu, b = er & 0xFFu, qa = ef >> 8;
			const bool agree = (a & b) != 0;
			if (ALGO == PAMD_ALGO_PEAR) {
				// the second index is the FORWARD read's quality at the reverse index (algo_pear.c:52,54, sic); past the forward
				// read the reference reads beyond the array: defined as quality 0 here
				const unsigned qb = ri < flen ? lds_read_u16(lF + 2u * (unsigned) ri) >> 8 : 0u;
				part += (a == 15u || b == 15u) ? nrb : lds_read_f64(lT + 8u * mat_index(agree, qa, qb));
			} else {
				part += lds_read_f64(lT + 8u * mat_index(agree, qa, er >> 8)) - nn;
			}
		}
	}
	return part;
}

// ---- pear / rdp_mle pairs with MANY candidates: no seeded overlap (every overlap is one) or tandem repeats -------------------
// Walking every candidate costs what the reference pays for such a pair (it counts them as "slow").  Most need not be walked:
// from three counts per candidate -- positions with an N or a base of quality below PAMD_STRONG_Q on either side, agreeing and
// disagreeing bases among the rest, taken sixteen positions per step from the 2-bit code streams like the counting scorers'
// scores -- and three maxima over the scorer's tables (PamdKernelParams::many_*) a candidate's score is bounded from above.
// Candidates are scored best bound first, each by the whole wave (its terms added in any order, |error| < 1e-9), until no
// bound reaches the best score found: what was not scored can neither win nor tie.  A comparison between two scored
// candidates closer than 1e-6 is left to the general kernel, which adds in the reference's order (`ambiguous`).
// Every lane holds up to NC candidates (as scan_many deals them: a short one travels with a long one).
struct ManySums {
	double best;
	int bo, examined, ambiguous;
};

// (At most 56 vector registers: what this function clobbers the caller cannot keep across the call, and the caller is the
// hot path, with 128 registers of its own in use -- given all of them, this function cost config 2's kernel a 16-byte
// spill per lane and pair in the middle of `vmcnt`'s queue, 480 B of HBM writes per pair.  It may spill itself: it is cold.)
template <int ALGO, bool LONG>
__device__ __attribute__((noinline)) ManySums scan_many_sums(unsigned char *wb, const double *ltab, int all, int size, int minov, int flen, int rlen, double nrb, double nn, double many_all, double many_match, double many_strong, double best, int lane) {
	typedef Lay<LONG> L;
	constexpr int NC = LONG ? 5 : 4;	// candidates per lane: 256, or 320
	constexpr unsigned LIST = L::T1, FNQ = L::T1 + 640 + 16, RNQ = FNQ + (LONG ? 96 : 64) + 16;	// (a pad word before either map, as before the reads' own)
	constexpr unsigned BND = RNQ + 96 + 16;	// the candidates' bounds, a float per candidate: [k][lane]
	static_assert(BND % 16 == 0 && BND + 4u * 64u * NC <= L::T1 + 3072, "inside the first two k-mer tables");
	uint16_t *list = reinterpret_cast<uint16_t *>(wb + LIST);	// overlap - minov of every candidate, ascending (where the k-mer tables were)
	int total = size;
	// the candidates, listed
	if (!all) {
		total = 0;
#pragma unroll
		for (int w = 0; w < (LONG ? 2 : 1); w++) {
			const unsigned mine = (w == 0 || lane < 32) ? reinterpret_cast<const unsigned *>(wb + L::FLAGS + 256 * w)[lane] : 0u;	// flag bytes are 0 or 1
			const unsigned c = (mine * 0x01010101u) >> 24;
			unsigned before = 0;
			int here = 0;
#pragma unroll
			for (int b = 0; b < 3; b++) {
				const u64 m = ballot(((c >> b) & 1u) != 0);
				before += (unsigned) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u)) << b;
				here += __popcll(m) << b;
			}
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if ((mine >> (8 * k)) & 1u) {
					list[total + before] = (uint16_t) (256 * w + 4 * lane + k);
					before++;
				}
			}
			total += here;
		}
	}
	// the N maps with the bases of low quality added: the forward read's from its own qualities; the reverse read's from its
	// own (rdp_mle), or from the FORWARD read's at the same index (pear looks its second quality up there, algo_pear.c:52,54;
	// 0 past that read)
	{
		constexpr unsigned QADD = (unsigned) (0x80 - PAMD_STRONG_Q) * 0x01010101u;
		auto low4 = [&](unsigned x, unsigned y) -> unsigned {	// bit 2k: quality of base k (of the four in x, y) below PAMD_STRONG_Q
			const unsigned q4 = __builtin_amdgcn_perm(y, x, 0x07050301u);
			const unsigned lowq = ~(((q4 & 0x7F7F7F7Fu) + QADD) >> 7) & 0x01010101u;
			const unsigned yb = lowq | (lowq >> 6);
			return (yb | (yb >> 12)) & 0x55u;
		};
		if constexpr (LONG) {
			if (lane < 48) {
				const uint4 f = lane < 40 ? reinterpret_cast<const uint4 *>(wb + L::F)[lane] : make_uint4(0u, 0u, 0u, 0u);
				const uint4 r = lane < 40 ? reinterpret_cast<const uint4 *>(wb + L::R)[lane] : make_uint4(0u, 0u, 0u, 0u);
				const unsigned lf = low4(f.x, f.y) | (low4(f.z, f.w) << 8);
				unsigned lr = low4(r.x, r.y) | (low4(r.z, r.w) << 8);
				if (ALGO == PAMD_ALGO_PEAR) {
					int c8 = flen - 8 * lane;
					c8 = c8 < 0 ? 0 : (c8 > 8 ? 8 : c8);
					lr = lf | (~((1u << (2 * c8)) - 1u) & 0x5555u);
				}
				reinterpret_cast<uint16_t *>(wb + FNQ)[lane] = (uint16_t) (reinterpret_cast<const uint16_t *>(wb + L::FN)[lane] | lf);
				reinterpret_cast<uint16_t *>(wb + RNQ)[lane] = (uint16_t) (reinterpret_cast<const uint16_t *>(wb + L::RN)[lane] | (lr << 1));
			}
		} else {
			const uint2 f = reinterpret_cast<const uint2 *>(wb + L::F)[lane], r = reinterpret_cast<const uint2 *>(wb + L::R)[lane];
			const unsigned lf = low4(f.x, f.y);
			unsigned lr = low4(r.x, r.y);
			if (ALGO == PAMD_ALGO_PEAR) {
				int c4 = flen - 4 * lane;
				c4 = c4 < 0 ? 0 : (c4 > 4 ? 4 : c4);
				lr = lf | (~((1u << (2 * c4)) - 1u) & 0x55u);
			}
			wb[FNQ + lane] = (unsigned char) (wb[L::FN + lane] | lf);
			wb[RNQ + lane] = (unsigned char) (wb[L::RN + lane] | (lr << 1));
		}
		if (lane < 4) {	// what lies before a map reads as N, what lies behind it too (the reads' own maps end that way)
			reinterpret_cast<unsigned *>(wb + FNQ - 16)[lane] = 0x55555555u;
			reinterpret_cast<unsigned *>(wb + RNQ - 16)[lane] = 0xAAAAAAAAu;
			reinterpret_cast<unsigned *>(wb + RNQ + (LONG ? 96 : 64))[lane] = 0xAAAAAAAAu;
		}
	}
	wave_sync();
	// This lane's candidates k = 0 .. NC-1 (number lane, 127 - lane, 128 + lane, 255 - lane, 256 + lane of the ascending list) and
	// the bounds of their scores.  Bounds live in LDS as floats rounded UP, candidates are re-read from the list where needed, and
	// no loop over k is unrolled: this function is cold, and every vector register it uses is one the hot path cannot keep
	// across the call (with all of them at its disposal it cost config 2's kernel a 16-byte spill per lane and pair).
	auto cand_of = [&](int k) -> int {	// the overlap of this lane's k-th candidate, 0: none
		const int c = (k & 1) ? 64 * (k + 1) - 1 - lane : 64 * k + lane;
		const int idx = all ? c : (int) list[c < total ? c : 0];
		return c < total ? idx + minov : 0;
	};
	float *const bnd = reinterpret_cast<float *>(wb + BND);
#pragma unroll 1
	for (int k = 0; k < NC; k++) {
		const int o = cand_of(k);
		// (lanes finish together: the k-th candidates of all lanes are walked in lockstep, to the longest of them)
		const int steps = wave_max_i32((o + 15) >> 4);
		unsigned acc = 0;
#pragma unroll 1
		for (int st = 0; st < steps; st++)
			acc += word_counts<L, FNQ, RNQ>(wb, flen, rlen, o, 16 * st, 16 * st < o);
		const int unknowns = (int) (acc & 1023u), matches = (int) (acc >> 10);
		const double b = (double) unknowns * many_all + (double) matches * many_match + (double) (o - unknowns - matches) * many_strong;
		float bf = (float) b;
		if ((double) bf < b)	// towards +inf: a bound may be too large, never too small
			bf = __uint_as_float(__float_as_uint(bf) + (bf > 0.0f ? 1u : (bf < 0.0f ? 0xFFFFFFFFu : 1u)));
		bnd[64 * k + lane] = o != 0 ? bf : -__builtin_inff();
	}
	// best bound first
	ManySums out;
	out.best = best;
	out.bo = -1;
	out.examined = total;
	out.ambiguous = 0;
	for (;;) {
		float b = -__builtin_inff();
		int which = 0;
#pragma unroll 1
		for (int k = 0; k < NC; k++) {
			const float v = bnd[64 * k + lane];
			if (v > b) {
				b = v;
				which = k;
			}
		}
		const int o = cand_of(which);
		// the wave's best: the largest bound (the smaller overlap among equal bounds, so that the order is fixed)
		const double wbnd = wave_max_f64((double) b);
		const int wov = smallest_of(ballot((double) b == wbnd), b > -__builtin_inff() ? o : 0x7FFFFFFF);
		if (wov == 0x7FFFFFFF || wbnd < out.best - 1e-6)
			break;	// nothing left, or nothing left that could reach the best score
		const double pr = wave_sum_any_order(score_share<ALGO, L>(wb, ltab, flen, rlen, wov, nrb, nn, lane));
		if (pr > out.best + 1e-6) {
			out.best = pr;
			out.bo = wov;
		} else if (!(pr < out.best - 1e-6)) {
			out.ambiguous = 1;
		}
		if (o == wov)	// (the winner is its lane's current best: that lane strikes it)
			bnd[64 * which + lane] = -__builtin_inff();
	}
	return out;
}

struct WaveArena {	// this wave's share of the compact arena, in units of 16 bases (rows and chunks are whole multiples of them): two scalar
	unsigned cur;	// registers instead of four -- the kernel has none to spare, and a spilled one costs vector instructions
	unsigned end;
};

// the registers a wave carries from one pair to the next (software prefetch): the lengths and this lane's share of the two
// rows -- four bases (8 bytes), or eight (16 bytes) in the LONG geometry
// v_min_f64 itself (no NaNs where it is used)
__device__ __forceinline__ double min_f64(double a, double b) {
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// ---- min_phred: the row's smallest phred (plugin_min_phred.c:8-22) without a chain of LDS round trips ------------------------
// the wave's smallest value, for values <= 0 (lanes DPP does not reach read as 0.0): DPP moves, no ds_bpermute
__device__ __forceinline__ double wave_min_nonpositive(double v) {
	v = min_f64(v, dpp_f64<0xB1>(v));	// quad_perm:[1,0,3,2]
	v = min_f64(v, dpp_f64<0x4E>(v));	// quad_perm:[2,3,0,1]
	v = min_f64(v, dpp_f64<0x141>(v));	// row_half_mirror
	v = min_f64(v, dpp_f64<0x140>(v));	// row_mirror
	v = min_f64(v, dpp_f64_rows<0x142, 0xA>(v));	// row_bcast:15
	v = min_f64(v, dpp_f64_rows<0x143, 0xC>(v));	// row_bcast:31
	return lane_f64(v, 63);
}
// panda_result_phred (nt.c:126-150) answers from where p lies among the 47 values of qual_score -- on one of them, or
// strictly between two (or beyond the last) -- and from nothing else: its walk only ever compares p with them.  The answers for
// the 47 + 48 places are made once per workgroup BY the walk (fk_phred_places: `on[k]`, `between[k]` = p in (score[k-1],
// score[k]), k = 47: beyond the last); a pair then finds its place with one table read per lane and two ballots instead of six
// dependent reads.  (pamd_io.hip's formatter does the same per base.)
__device__ __forceinline__ void fk_phred_places(const double *score, unsigned char *places, int tid) {
	if (tid < 47)
		places[tid] = (unsigned char) result_phred(score, score[tid]);
	else if (tid < 47 + 48) {
		const int k = tid - 47;
		const double p = k == 0 ? score[0] * 2.0 : (k == 47 ? score[46] * 0.5 : 0.5 * (score[k - 1] + score[k]));
		places[48 + k] = (unsigned char) result_phred(score, p);
	}
}
__device__ __forceinline__ int fk_phred_of(const double *score, const unsigned char *places, double p, int lane) {
	const double s = score[lane < 47 ? lane : 46];
	const u64 on = ballot(lane < 47 && s == p), below = ballot(lane < 47 && s < p);
	return on != 0 ? (int) places[__builtin_ctzll(on)] : (int) places[48 + __popcll(below)];
}


// ---- sums in the reference's order without a second pass over the rows (round 5) -------------------------------------------
// `quality` is three ordered chains over a row's log-probabilities (assembler.c:171,226,236: the forward-only bases, the overlap,
// the reverse-only bases) and (f + r + o) / len (assembler.c:244).  A chain is a dependent sequence of additions: one lane's work,
// and 63 lanes can walk 63 chains at once -- IF each finds its chain's values without a memory round trip per step.  Rounds 2-4
// walked the rows themselves where the merge had put them, in HBM, sixty-four pairs later: 8 bytes per base read back, every lane
// in another row (1.85x the algorithmic traffic, +32 % time at full size).  Now the merge also leaves each base's TABLE ENTRY (u16:
// the log-p is an entry of the staged image; stored as its byte offset there) in a scratch row of this wave's -- FK_XG rows of
// FK_XSTRIDE entries, 90-110 MB for the whole grid --, and every FK_XG pairs sixty-three lanes walk their three chains each from
// there: two bytes per base, sixteen per load, a block of loads under way while the block before it is added up, the values from the
// tables in LDS.  The result goes into the pair's parked record as its final quality (finish64's "q is final").
// Measured (config 1, 50 M pairs): 555 -> 645 M pairs/s (+17 % over any-order sums instead of +32 %), L2 <-> fabric traffic 8 256 ->
// 6 315 B per pair (1.85x -> 1.42x of the algorithmic bytes: the scratch rows, 10 MB per XCD, do not stay in its 4 MB L2; what
// lies behind the fabric -- the 256 MB memory-side cache or HBM -- the counters cannot tell apart).  What the walk costs, by
// ablation on 8 M pairs: the entry stores and the wait for them 1 %, the table reads and additions 11 % (five vector instructions
// per element when first measured, three now), the loads' latency the rest -- 27 % with eight 2-byte loads per lane in flight, the
// compiler waiting for each; 9 % with four 16-byte loads issued a block ahead.  Eight a block ahead cost the callee 120 registers and
// the pair loop spills: slower than four.  Groups of 16 pairs (48 lanes): 630 M; of 21 (63 lanes, finish after 63 pairs): 640 M; the
// reads' rows loaded with the nontemporal hint, so that they do not push the scratch rows out of the L2: 645 M.
// pairs per group: 3 x 21 chains in 63 lanes, three groups per finish (which then takes 63 pairs, not 64); pear / rdp_mle (SCORE):
// a fourth chain per pair, the winning overlap's score (algo_pear.c:32-59, algo_rdp_mle.c:43-74: an ordered sum as well): 4 x 15
// chains in 60 lanes, four groups per finish
template <bool SCORE>
constexpr int fk_xg() {
	return SCORE ? 15 : 21;
}
constexpr int FK_XG_MAX = 21;
template <bool LONG, bool SCORE>
constexpr unsigned fk_xstride() {
	return (unsigned) Lay<LONG>::NR * 64u + (SCORE ? (unsigned) Lay<LONG>::SR * 64u : 0u);	// entries per slot: every round of the merge writes its 64; then the score's terms
}
struct GroupSums {
	unsigned q_lo, q_hi;	// lane base + s: the quality of the group's s-th pair, if `done`
	unsigned b_lo, b_hi;	// SCORE: its score
	bool done;
};
// len_bo / pid / ro_st / off_*: the lane's parked fields (lane = slot of the wave's pending pairs); base: first slot of the group,
// count: pairs in it.  sub (SCORE): what every term of the score has subtracted before it is added (rdp_mle: qual_nn,
// algo_rdp_mle.c:68-70; pear: nothing).  Out of line: it runs once per group and what it clobbers the pair loop does not keep
// across the call.
template <bool LONG, bool SCORE, bool SUBTRACT>
__device__ __attribute__((noinline)) GroupSums exact_sums_group(unsigned lT, const uint16_t *xs, unsigned len_bo, unsigned pid, unsigned ro_st, unsigned off_lo, unsigned off_hi, double sub, int base, int count, int lane) {
	constexpr unsigned XS = fk_xstride<LONG, SCORE>();
	constexpr int XG = fk_xg<SCORE>(), NCH = SCORE ? 4 : 3;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// this wave's index stores have left (a CU's vector L1 is coherent for its own stores)
	const int chain = lane >= 3 * XG ? 3 : (lane >= 2 * XG ? 2 : (lane >= XG ? 1 : 0)), s = lane - XG * chain;	// (lanes past NCH * XG: idle, they read as a further chain of slot 0..)
	const int src = 4 * (base + (s < XG ? s : 0));	// ds_bpermute takes a byte address: lane * 4
	const unsigned g_pid = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) pid), g_len_bo = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) len_bo);
	const unsigned g_ro_st = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) ro_st);
	const unsigned g_off = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) (off_lo & off_hi));	// all ones: the row found no room (the pair ends as NOSPACE)
	const int a0 = (int) (g_pid & 0x3FFu), af = (int) ((g_pid >> 10) & 0x3FFu), seqlen = (int) (g_pid >> 20);	// (the merge's length: `-a` has clipped the record's)
	const int bo = (int) (g_len_bo >> 16);
	const bool valid = s < count && s < XG && lane < NCH * XG && ((g_ro_st >> 16) & 0xFFu) == (unsigned) PAMD_ST_OK && g_off != 0xFFFFFFFFu;
	const int lo = !valid || chain == 0 || chain == 3 ? 0 : (chain == 1 ? a0 : af), hi = chain == 0 ? a0 : (chain == 1 ? af : (chain == 2 ? seqlen : bo));
	const int n = valid && hi > lo ? hi - lo : 0;
	const int steps = wave_max_i32(n);
	// Eight entries per 16-byte load (the chain starts at any entry: the loads are unaligned, which global loads may be), a block of
	// FK_XD loads per lane under way while the block before it is added up: a lane's chain is a few hundred dependent additions
	// behind a few dozen loads from lines the merge wrote some microseconds ago -- what it must not do is wait for each in turn.
	// An entry is the byte offset of its value in the table image; a lane past its chain's end adds the image's spare 0.0 (the
	// score's chain, whose terms have `sub` subtracted: the entry that holds `sub` itself).
	typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(2)));
	typedef __attribute__((address_space(1))) const u32x4u gbl_cu128u;
	gbl_cu128u *const row = (gbl_cu128u *) (xs + (unsigned) (s < XG ? s : 0) * XS + (chain == 3 ? (unsigned) Lay<LONG>::NR * 64u : (unsigned) lo));
	constexpr int FK_XD = 4;
	const bool scoring = SCORE && chain == 3;
	const unsigned pad_off = (SUBTRACT && scoring) ? 8u * (PAMD_LT_SCORE + 47u) : 8u * PAMD_LT_ZERO;	// (slot 47 of the score block holds qual_nn)
	const double my_sub = (SUBTRACT && scoring) ? sub : 0.0;
	const int nv = (n + 7) >> 3, last = nv > 0 ? nv - 1 : 0;	// (a lane past its chain's end re-reads the chain's last vector)
	const int vsteps = (steps + 7) >> 3;
	u32x4u nxt[FK_XD];
#pragma unroll
	for (int d = 0; d < FK_XD; d++)
		nxt[d] = row[d < last ? d : last];
	double acc = 0.0;
	for (int k0 = 0; k0 < vsteps; k0 += FK_XD) {
		u32x4u cur[FK_XD];
#pragma unroll
		for (int d = 0; d < FK_XD; d++)
			cur[d] = nxt[d];
		if (k0 + FK_XD < vsteps) {	// the next block's loads, all of them, before this block's additions
#pragma unroll
			for (int d = 0; d < FK_XD; d++)
				nxt[d] = row[k0 + FK_XD + d < last ? k0 + FK_XD + d : last];
		}
#pragma unroll
		for (int d = 0; d < FK_XD; d++) {
			const int left = n - 8 * (k0 + d);	// entries of this vector that belong to the chain (<= 0: none)
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const unsigned w = u < 2 ? cur[d].x : (u < 4 ? cur[d].y : (u < 6 ? cur[d].z : cur[d].w));
				const unsigned e = (u & 1) ? w >> 16 : w & 0xFFFFu;
				v[u] = lds_read_f64(lT + (u < left ? e : pad_off));
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
				acc += SUBTRACT ? v[u] - my_sub : v[u];	// left to right: the reference's order (x - 0.0 and x + 0.0 are x)
			__builtin_amdgcn_sched_barrier(0);	// (one vector's table reads at a time: hoisted together they take a hundred registers, which the caller then spills)
		}
	}
	// lane s (chain 0) gathers its pair's other sums; then lane base + s takes them
	union {
		double d;
		int i[2];
	} f, o, r, q, sc;
	f.d = acc;
	o.i[0] = __builtin_amdgcn_ds_bpermute(4 * (s + XG), f.i[0]);
	o.i[1] = __builtin_amdgcn_ds_bpermute(4 * (s + XG), f.i[1]);
	r.i[0] = __builtin_amdgcn_ds_bpermute(4 * (s + 2 * XG), f.i[0]);
	r.i[1] = __builtin_amdgcn_ds_bpermute(4 * (s + 2 * XG), f.i[1]);
	sc.d = 0.0;
	if (SCORE) {
		sc.i[0] = __builtin_amdgcn_ds_bpermute(4 * ((s + 3 * XG) & 63), f.i[0]);
		sc.i[1] = __builtin_amdgcn_ds_bpermute(4 * ((s + 3 * XG) & 63), f.i[1]);
	}
	q.d = (f.d + r.d + o.d) / (double) (seqlen + 1);	// assembler.c:244
	// (lanes 0 .. XG-1 hold the group's results; lane base + t of the parked records takes the t-th)
	const int from = 4 * ((lane - base) & 63);
	const int okv = __builtin_amdgcn_ds_bpermute(from, valid ? 1 : 0);	// (chain 0's verdict: the pair has a row)
	GroupSums out;
	out.q_lo = (unsigned) __builtin_amdgcn_ds_bpermute(from, q.i[0]);
	out.q_hi = (unsigned) __builtin_amdgcn_ds_bpermute(from, q.i[1]);
	out.b_lo = (unsigned) __builtin_amdgcn_ds_bpermute(from, sc.i[0]);
	out.b_hi = (unsigned) __builtin_amdgcn_ds_bpermute(from, sc.i[1]);
	out.done = lane >= base && lane < base + count && okv != 0;
	return out;
}

template <bool LONG>
struct Carry;
template <>
struct Carry<false> {
	int flen_v, rlen_v;
	uint2 f, r;
};
template <>
struct Carry<true> {
	int flen_v, rlen_v;
	uint4 f, r;
};

__device__ __forceinline__ void settle(Carry<false> &c) {	// the compiler must treat the carried registers as written here
	asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.r.x), "+v"(c.r.y));
}
__device__ __forceinline__ void settle(Carry<true> &c) {
	asm volatile("" : "+v"(c.flen_v), "+v"(c.rlen_v), "+v"(c.f.x), "+v"(c.f.y), "+v"(c.f.z), "+v"(c.f.w), "+v"(c.r.x), "+v"(c.r.y), "+v"(c.r.z), "+v"(c.r.w));
}

// STREAM (sums in the reference's order): the rows are read once; with the nontemporal hint they leave the L2 to the scratch rows
// of table indices, which are written and read back within a group of pairs (exact_sums_group)
template <bool STREAM>
__device__ __forceinline__ uint2 load_row(const uint8_t *row, unsigned row_bytes, int lane) {
	unsigned off = (unsigned) lane * 8u;
	off = off < row_bytes ? off : row_bytes - 8u;	// lanes past the row's end re-read its last bytes (rows are whole multiples of 16 B)
	if (STREAM) {
		const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(row + off));
		return make_uint2(v.x, v.y);
	}
	return *reinterpret_cast<const uint2 *>(row + off);
}
template <bool STREAM>
__device__ __forceinline__ uint4 load_row16(const uint8_t *row, unsigned row_bytes, int lane) {
	unsigned off = (unsigned) lane * 16u;
	off = off < row_bytes ? off : row_bytes - 16u;
	if (STREAM) {
		const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(row + off));
		return make_uint4(v.x, v.y, v.z, v.w);
	}
	return *reinterpret_cast<const uint4 *>(row + off);
}
// (measured and not kept: the row's address made in the scalar ALU and the load given a scalar base -- the quarter-rate 64-bit
// multiply-add per lane goes, a dozen more scalar registers spill into vector lanes, the rate does not move)
template <bool LONG, bool STREAM = false>
__device__ __forceinline__ void load_pair(Carry<LONG> &c, const PamdKernelParams &P, unsigned q, int lane) {
	c.flen_v = P.flen[q];
	c.rlen_v = P.rlen[q];
	if constexpr (LONG) {
		c.f = load_row16<STREAM>(P.fwd + (u64) q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
		c.r = load_row16<STREAM>(P.rev + (u64) q * P.rev_row_bytes, P.rev_row_bytes, lane);
	} else {
		c.f = load_row<STREAM>(P.fwd + (u64) q * P.fwd_row_bytes, P.fwd_row_bytes, lane);
		c.r = load_row<STREAM>(P.rev + (u64) q * P.rev_row_bytes, P.rev_row_bytes, lane);
	}
}

// one level of the forward insert: the smaller of (slot, carry) stays in the slot, the larger one moves on
// (EMPTY = nothing moves on).  The order of the words is irrelevant; ds_min only makes "who stays" independent of the
// order in which lanes and instructions reach a slot, so no lane ever has to retry.
__device__ __forceinline__ unsigned insert_level(unsigned *table, unsigned slot, unsigned carry, unsigned limit = EMPTY) {
	if (carry < limit) {	// lanes with nothing to insert stay out of the LDS (they would all meet on one slot)
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

// One level of the forward insert for all of a lane's words at once: the ds_min of every word that is not EMPTY is issued --
// a lane with nothing to insert stays out of the LDS (exec) --, then ONE wait; what moves on is max(slot's previous word, word),
// EMPTY staying EMPTY.  Inline assembly because the compiler, given `if (carry != EMPTY) old = atomicMin(..)` four times,
// waits for each atomic inside its own branch: eight exposed LDS round trips per pair instead of two.  (The wait sits inside
// the statement: between issue and wait nothing may touch the destination registers, and the compiler's own counter
// bookkeeping does not see into an asm statement.)
#define FK_INS(W, A, C) "v_cmp_ne_u32_e32 vcc, -1, " C "\n\ts_and_saveexec_b64 %[sv], vcc\n\tds_min_rtn_u32 " W ", " A ", " C " offset:%[off]\n\ts_mov_b64 exec, %[sv]\n\t"
template <int BPL, int OFF = 0>
__device__ __forceinline__ void insert_all(const unsigned (&addr)[BPL], unsigned (&carry)[BPL]) {
	unsigned was[BPL];
#pragma unroll
	for (int j = 0; j < BPL; j++)
		was[j] = 0;
	u64 sv;
	static_assert(BPL >= 3 && BPL <= 5, "three, four or five words per lane");
	if constexpr (BPL == 3)
		asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") "s_waitcnt lgkmcnt(0)"
			     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [sv] "=&s"(sv)
			     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [off] "n"(OFF)
			     : "vcc", "scc", "memory");
	else if constexpr (BPL == 4)
		asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") FK_INS("%[w3]", "%[a3]", "%[c3]") "s_waitcnt lgkmcnt(0)"
			     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [w3] "+v"(was[3]), [sv] "=&s"(sv)
			     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [a3] "v"(addr[3]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [c3] "v"(carry[3]), [off] "n"(OFF)
			     : "vcc", "scc", "memory");
	else
		asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") FK_INS("%[w3]", "%[a3]", "%[c3]") FK_INS("%[w4]", "%[a4]", "%[c4]") "s_waitcnt lgkmcnt(0)"
			     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [w3] "+v"(was[3]), [w4] "+v"(was[4]), [sv] "=&s"(sv)
			     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [a3] "v"(addr[3]), [a4] "v"(addr[4]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [c3] "v"(carry[3]), [c4] "v"(carry[4]), [off] "n"(OFF)
			     : "vcc", "scc", "memory");
#pragma unroll
	for (int j = 0; j < BPL; j++)
		carry[j] = was[j] > carry[j] ? was[j] : carry[j];
}
#undef FK_INS
// (FK_EPOCH: a word is live -- this pair's, not EMPTY, not a stale one just displaced -- when it lies below the epoch's upper limit)
#define FK_INS(W, A, C) "v_cmp_gt_u32_e32 vcc, %[lim], " C "\n\ts_and_saveexec_b64 %[sv], vcc\n\tds_min_rtn_u32 " W ", " A ", " C " offset:%[off]\n\ts_mov_b64 exec, %[sv]\n\t"
template <int BPL, int OFF = 0>
__device__ __forceinline__ void insert_all_live(const unsigned (&addr)[BPL], unsigned (&carry)[BPL], unsigned limit) {
	unsigned was[BPL];
#pragma unroll
	for (int j = 0; j < BPL; j++)
		was[j] = 0;
	u64 sv;
	static_assert(BPL == 3 || BPL == 4, "the short geometry's three or four words per lane");
	if constexpr (BPL == 3)
		asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") "s_waitcnt lgkmcnt(0)"
			     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [sv] "=&s"(sv)
			     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [lim] "s"(limit), [off] "n"(OFF)
			     : "vcc", "scc", "memory");
	else
		asm volatile(FK_INS("%[w0]", "%[a0]", "%[c0]") FK_INS("%[w1]", "%[a1]", "%[c1]") FK_INS("%[w2]", "%[a2]", "%[c2]") FK_INS("%[w3]", "%[a3]", "%[c3]") "s_waitcnt lgkmcnt(0)"
			     : [w0] "+v"(was[0]), [w1] "+v"(was[1]), [w2] "+v"(was[2]), [w3] "+v"(was[3]), [sv] "=&s"(sv)
			     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [a3] "v"(addr[3]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [c3] "v"(carry[3]), [lim] "s"(limit), [off] "n"(OFF)
			     : "vcc", "scc", "memory");
#pragma unroll
	for (int j = 0; j < BPL; j++)
		carry[j] = was[j] > carry[j] ? was[j] : carry[j];
}
#undef FK_INS
// FK_INS_PLAIN (short geometry, four words per lane): the levels at which most words are live -- the bucket's two slots -- without
// the test: a word that is not live (EMPTY, or a stale one just displaced) loses its ds_min against anything live and leaves a slot
// that was not live not live, and max(previous, word) is not live either.  Per word a compare and two scalar instructions (exec
// saved and restored) fewer, and no zeroed result register: 16 vector and 16 scalar instructions a pair, +1.2 % on config 1 (the
// first slot alone: nothing; profiles/r06_experiments_ab.txt).  Not where lanes idle -- three k-mers per lane, or five: their words
// would all meet on one slot.  1: the first slot; 2: the second slot too
#ifndef FK_INS_PLAIN
#define FK_INS_PLAIN 2
#endif
#ifndef FK_INS_PLAIN_ALWAYS	/* measurement: without the test for a read that fills the lanes */
#define FK_INS_PLAIN_ALWAYS 0
#endif
template <int OFF>
__device__ __forceinline__ void insert_all_plain(const unsigned (&addr)[4], unsigned (&carry)[4]) {
	unsigned was[4];
	asm volatile("ds_min_rtn_u32 %[w0], %[a0], %[c0] offset:%[off]\n\tds_min_rtn_u32 %[w1], %[a1], %[c1] offset:%[off]\n\tds_min_rtn_u32 %[w2], %[a2], %[c2] offset:%[off]\n\tds_min_rtn_u32 %[w3], %[a3], %[c3] offset:%[off]\n\ts_waitcnt lgkmcnt(0)"
		     : [w0] "=&v"(was[0]), [w1] "=&v"(was[1]), [w2] "=&v"(was[2]), [w3] "=&v"(was[3])
		     : [a0] "v"(addr[0]), [a1] "v"(addr[1]), [a2] "v"(addr[2]), [a3] "v"(addr[3]), [c0] "v"(carry[0]), [c1] "v"(carry[1]), [c2] "v"(carry[2]), [c3] "v"(carry[3]), [off] "n"(OFF)
		     : "memory");
#pragma unroll
	for (int j = 0; j < 4; j++)
		carry[j] = was[j] > carry[j] ? was[j] : carry[j];
}

// why pairs are deferred, counted in P.phase_cycles[0..5] (read back with pamd_debug_phase_cycles): 0 an nt or quality
// outside the lean domain, 1 a k-mer found no slot, 2 a score comparison too close to call (pear, rdp_mle), 3 more than four
// candidates (pear, rdp_mle, LONG), 4 quality too close to the LOWQ threshold, (debug builds) 5 counts pairs that used the
// last two k-mer tables, 6 the primer scan did not settle a read, 7 an overlap that runs into a clipped primer
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
	for (unsigned i = lane; i < Geo<false, 10>::WAVE_BYTES / 4; i += 64)
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

// FK_LAZYQ (any-order sums, short geometry): a pair's quality sum is not reduced over the wave when the pair is done -- six DPP steps,
// twenty vector instructions -- but merged with the other pairs' of its group of 64 in a binary tree: two vectors of 64 partial sums
// become one that holds 32 of each, two of those one that holds 16 of four pairs each, ... and after the 64th pair one vector holds
// the 64 pairs' sums, one per lane: 63 merges for 64 pairs.  The two levels that merge most often go through v_permlane32_swap /
// v_permlane16_swap (two swaps and an add); the rest through DPP moves with bank masks or quad permutations (seven instructions,
// seven times in 64 pairs).  A level merges along one bit of the lane number: the s-th pair's sum ends in lane bitrev6(s), which
// is where park() puts the pair's other scalars and where finish64<.., LAZYQ> looks for them.
#ifndef FK_LAZYQ
#define FK_LAZYQ 1
#endif
// FK_LAZYP (with FK_LAZYQ, a min_phred module): the row's smallest log-p goes through a merge tree of its own (minimum instead of
// sum) and becomes a phred in finish64, 64 pairs at once: the six DPP steps, the lookup of its place among the score table's values
// and that lookup's two LDS round trips leave the pair loop.  What the pair still says itself: whether one of its values lay at or
// below qual_score[0] (bit 26 of the parked ro_st; panda_result_phred answers 1 there, nt.c:126-150).
#ifndef FK_LAZYP
#define FK_LAZYP 1
#endif
struct QTree {
	double lvl[6];	// level L: the merged partial sums of 2^L pairs, waiting for the next 2^L (held while bit L of the slot count is set)
};
union F64Pair {
	double d;
	unsigned u[2];
};
// one vector from two: lanes whose bit D is clear keep x[i] + x[i ^ D], the others y[i] + y[i ^ D]
// (MIN: the smaller of the two instead of their sum -- the tree of the rows' smallest log-p, FK_LAZYP)
template <int D, bool MIN = false>
__device__ __forceinline__ double qsum_merge(double xd, double yd, int lane) {
	F64Pair x, y, t, w;
	x.d = xd;
	y.d = yd;
	if constexpr (D == 32 || D == 16) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			// (D = 32: the upper half of x trades places with the lower half of y; D = 16: x's odd rows with y's even rows)
			const auto r = D == 32 ? __builtin_amdgcn_permlane32_swap(x.u[k], y.u[k], false, false) : __builtin_amdgcn_permlane16_swap(x.u[k], y.u[k], false, false);
			t.u[k] = r[0];
			w.u[k] = r[1];
		}
	} else if constexpr (D == 8) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			unsigned v = (unsigned) __builtin_amdgcn_update_dpp(0, (int) y.u[k], 0x128, 0xF, 0xF, false);	// row_ror:8, every lane: y's other half-row
			v = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) x.u[k], 0x128, 0xF, 0x3, false);	// ... lanes 0-7 of a row: x's
			w.u[k] = v;
			t.u[k] = (unsigned) __builtin_amdgcn_update_dpp((int) x.u[k], (int) y.u[k], 0xE4, 0xF, 0xC, false);	// x, lanes 8-15 of a row from y
		}
	} else if constexpr (D == 4) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			unsigned v = (unsigned) __builtin_amdgcn_update_dpp(0, (int) x.u[k], 0x104, 0xF, 0x5, false);	// row_shl:4 into banks 0 and 2: x[i + 4]
			v = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) y.u[k], 0x114, 0xF, 0xA, false);	// row_shr:4 into banks 1 and 3: y[i - 4]
			w.u[k] = v;
			t.u[k] = (unsigned) __builtin_amdgcn_update_dpp((int) x.u[k], (int) y.u[k], 0xE4, 0xF, 0xA, false);	// x, banks 1 and 3 from y
		}
	} else {
		const bool up = (lane & D) != 0;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const unsigned moving = up ? x.u[k] : y.u[k];	// x's upper and y's lower lanes of each pair of lanes
			w.u[k] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) moving, D == 2 ? 0x4E : 0xB1, 0xF, 0xF, false);	// quad_perm [2,3,0,1] / [1,0,3,2]
			t.u[k] = up ? y.u[k] : x.u[k];
		}
	}
	if constexpr (MIN) {
		double r;
		asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(t.d), "v"(w.d));	// (the bare instruction: no NaNs here, see min_f64)
		return r;
	}
	return t.d + w.d;
}
// the s-th pair of a group of 64 hands in its 64 partial sums; true when the group's sums are complete (s == 63): they are `q`'s
template <bool MIN = false>
__device__ __forceinline__ bool qsum_insert(QTree &qt, int s, double q, int lane, double &out) {
	if ((s & 1) == 0) {
		qt.lvl[0] = q;
		return false;
	}
	q = qsum_merge<32, MIN>(qt.lvl[0], q, lane);
	if ((s & 2) == 0) {
		qt.lvl[1] = q;
		return false;
	}
	q = qsum_merge<16, MIN>(qt.lvl[1], q, lane);
	if ((s & 4) == 0) {
		qt.lvl[2] = q;
		return false;
	}
	q = qsum_merge<8, MIN>(qt.lvl[2], q, lane);
	if ((s & 8) == 0) {
		qt.lvl[3] = q;
		return false;
	}
	q = qsum_merge<4, MIN>(qt.lvl[3], q, lane);
	if ((s & 16) == 0) {
		qt.lvl[4] = q;
		return false;
	}
	q = qsum_merge<2, MIN>(qt.lvl[4], q, lane);
	if ((s & 32) == 0) {
		qt.lvl[5] = q;
		return false;
	}
	out = qsum_merge<1, MIN>(qt.lvl[5], q, lane);
	return true;
}

// launch parameters every pair needs, read ONCE per wave into scalar registers (FK_KARGS): left as kernel arguments, the compiler
// re-reads them from the argument segment inside the pair loop -- three of the five scalar loads per pair with their wait right
// behind them (the counter is shared with the LDS reads that follow: the wait cannot be put off)
struct HotArgs {
	double qual_nn, pmatch, pmismatch, threshold;
	u64 stride, capacity;
	uint8_t *nt;
	double *p;
};

// LONG: the geometry (Lay<LONG>).  PRIM (LONG only): primers (`-p` / `-q`, found by primer_quick) or trims are configured.
// NARROW (short geometry only): reads of at most 192 bases seed three bases per lane, on windows of the code streams
// POST (LONG only): `-a`, the primers are looked for in the merged sequence and cut off it (reference-order sums since round 5:
// `quality` is the whole merge's, assembler.c:244 precedes the clipping, and comes from the group walk like any other row's)
template <int ALGO, bool EXACT, bool PHRED, bool LONG, bool PRIM, bool NARROW, bool POST, bool LAZYQ = false>
__device__ __forceinline__ void fast_pair(const PamdKernelParams &P, const double *ltab, const unsigned char *places, unsigned char *wb, WaveArena &arena, Pend64 &pd, int slot_idx, bool &deep_dirty, u64 pair, int lane, const int flen, const int rlen, Carry<LONG> &cur, const Carry<LONG> &nxt, unsigned primer_bases, uint16_t *xs, const HotArgs &H, bool &flags_dirty, unsigned &epoch, std::conditional_t<LONG, LenMasks8, LenMasks> &lmf, std::conditional_t<LONG, LenMasks8, LenMasks> &lmr, double *qlane = nullptr, double *plane = nullptr) {
	typedef Lay<LONG> L;
	constexpr bool LAZYP = FK_LAZYP != 0 && LAZYQ && PHRED;
	if constexpr (LAZYP)
		*plane = 0.0;	// (nothing below 0.0: the neutral element of the tree's minimum)
	static_assert(!LAZYQ || (!EXACT && !POST), "the merge tree holds sums in any order, of whole rows");
	if constexpr (LAZYQ)
		*qlane = 0.0;	// (a pair that does not get as far as its row hands in nothing)
	constexpr int BPL = NARROW ? 3 : L::BPL;
	constexpr bool WINDOWS = BPL != 4;	// k-mers from unaligned words of the code streams (else from the neighbouring lanes' registers)
	static_assert(!NARROW || !LONG, "three bases per lane belong to the short geometry");
	static_assert(!PRIM || LONG, "primers and trims are served by the LONG geometry");
	static_assert(!POST || (LONG && !PRIM), "`-a` is served by the LONG geometry");
	const auto fv = cur.f, rv = cur.r;
	bool rotated = false;
	int fo = 0, ro = 0;	// bases clipped from the reads' 5' ends (primers, trims): assembler.c:262-288
	int region_a0 = 0, region_af = 0, region_len = 0;	// the merged row's overlap region and length, for the ordered sums
	bool primers_known = false;
	int status = PAMD_ST_OK;
	u64 off = ~0ull;
	double best = 0.0, qsum = 0.0;
	int seqlen = 0, bo = -1, examined = 0, mism = 0, degen = 0, phmin = 127;
	bool phflag = false;	// (FK_LAZYP) one of the row's log-p lies at or below qual_score[0]
	bool slow = false;
	const int minov = P.minoverlap;
	constexpr int T1B = t1_bits<ALGO, LONG>();
	constexpr bool BUCK = FK_BUCKETS == 2 || (FK_BUCKETS == 1 && !NARROW);
	constexpr bool EP = FK_EPOCH != 0 && !LONG;	// (the LONG geometry lays its primer scan and its merged row over the first table: cleared per pair as ever)
	constexpr unsigned KSH = EP ? 9u : 16u, POS_END = EP ? 0x200u : 0x10000u;	// a table word is key << KSH | position; an entry of the k-mer looked for reads as its bare position, below POS_END
	typedef Geo<LONG, T1B> G;
	unsigned *const t1 = reinterpret_cast<unsigned *>(wb + L::T1), *const t2 = reinterpret_cast<unsigned *>(wb + G::T2);
	unsigned *const t3 = reinterpret_cast<unsigned *>(wb + G::T3), *const t4 = reinterpret_cast<unsigned *>(wb + G::T4);
	uint8_t *const flags = wb + L::FLAGS;
	uint8_t *const nts = wb + L::NT;

	do {
		if (__builtin_expect(flen < 2 || rlen < 2, 0)) {	// assembler.c:255-258
			status = PAMD_ST_BADR;
			break;
		}
		if (!PRIM) {
			if ((flen < rlen ? flen : rlen) < minov) {	// assembler.c:289-292 (no primers, no trims here)
				status = PAMD_ST_BADR;
				break;
			}
			if (minov >= flen || minov >= rlen) {	// assembler.c:73-82
				status = PAMD_ST_NOALGN;
				break;
			}
		}
		wave_sync();	// the previous pair's readers are done with this wave's LDS
		MARK("stage");
		// ---- stage ----
		unsigned fcodes = 0, fnm = 0, rcodes = 0, rnm = 0, weird = 0;
		int uf, ur;
		if constexpr (LONG) {
			const unsigned qadd = (unsigned) (0x80 - P.primer_qthr) * 0x01010101u;
			if (!FK_LEN_MASKS || flen != lmf.len)
				len_masks8(lmf, flen, lane);
			if (!FK_LEN_MASKS || rlen != lmr.len)
				len_masks8(lmr, rlen, lane);
			uf = stage8<true, PRIM>(fv, flen, lane, wb, weird, qadd, lmf);
			ur = stage8<false, PRIM>(rv, rlen, lane, wb, weird, qadd, lmr);
		} else {
#if FK_LEN_MASKS
			if (flen != lmf.len)
				len_masks(lmf, flen, lane);
			if (rlen != lmr.len)
				len_masks(lmr, rlen, lane);
#endif
			uf = stage<true>(fv, flen, lane, wb, fcodes, fnm, weird, lmf);
			ur = stage<false>(rv, rlen, lane, wb, rcodes, rnm, weird, lmr);
		}
		if (__builtin_expect(ballot(weird != 0) != 0, 0)) {	// an IUPAC ambiguity code, nt 0, a quality outside 0..46: the general kernel's pair
			note(P, 0, lane);
			status = PAMD_ST_DEFER;
			break;
		}
		STOP(5);	// (ablation builds: stage alone; STOP(0) then adds the primers)
		if constexpr (PRIM) {
			// ---- primers and trims (assembler.c:262-292); the reverse primer is looked for only in pairs that have the
			// forward one.  One copy of the scan serves both reads (everything that selects the read is wave-uniform).
			MARK("primers");
			wave_sync();
			fo = P.ftrim;
			ro = P.rtrim;
			bool unsettled = false;
			{
				int fpl = P.fprimer_len, rpl = P.rprimer_len;
				asm volatile("" : "+s"(fpl), "+s"(rpl));	// (what follows is two multiplications per pair; hoisted out of the loop it is two spilled doubles)
				// (measured: the primer geometry's kernel re-reads a dozen kernel arguments per pair -- trivially rematerialisable, the
				// allocator loads them again rather than park them in a vector lane; handed over as opaque values from outside the pair
				// loop, nine of 27 scalar loads go: no change in the rate)
				const double thr_f = (double) fpl * P.threshold, thr_r = (double) rpl * P.threshold, inv_err_t = P.primer_inv_err_t;
				// per read: -3 no primer configured, -2 not looked at yet, -1 not settled yet, >= 0 what the scan returns
				// (a primer as long as the read, or longer, is not found: offset.c:55-57, :70)
				int rf = fpl <= 0 ? -3 : (flen - fpl <= 0 ? 0 : -2), rr = rpl <= 0 ? -3 : (rlen - rpl <= 0 ? 0 : -2);
				unsigned pkf = 0, pkr = 0;
				int s0f = -1, s0r = -1;
				if (rf == -2)
					pkf = primer_tier1(lds_addr(wb + L::FC), lds_addr(wb + L::CAREF), flen - fpl, lane, P.primer_codes[0].code[0], P.primer_codes[0].counts[0], s0f);
				if (rr == -2)
					pkr = primer_tier1(lds_addr(wb + L::RC), lds_addr(wb + L::CARER), rlen - rpl, lane, P.primer_codes[1].code[0], P.primer_codes[1].counts[0], s0r);
				double sumf = 0.0, sumr = 0.0;
				if (s0f >= 0 || s0r >= 0)
					primer_sums_any_order(ltab, wb, s0f, s0r, fpl, rpl, (primer_bases >> 16) & 0xFFu, lane, sumf, sumr);
				if (rf == -2)
					rf = s0f >= 0 ? primer_shortcut(ltab, wb, L::F, primer_bases & 0xFFu, pkf, s0f, sumf, flen, fpl, thr_f, inv_err_t, lane) : -1;
				if (rr == -2)
					rr = s0r >= 0 ? primer_shortcut(ltab, wb, L::R, (primer_bases >> 8) & 0xFFu, pkr, s0r, sumr, rlen, rpl, thr_r, inv_err_t, lane) : -1;
				// the reads that did not settle (one copy of the code: everything that selects the read is wave-uniform);
				// the reverse primer is looked for only in pairs that have the forward one
#pragma unroll 1
				for (int which = 0; which < 2; which++) {
					if ((which == 0 ? rf : rr) != -1 || (which == 1 && rf <= 0 && rf != -3))
						continue;
					MARK("primers-settle");
#ifdef PAMD_FK_DEBUG
					note(P, (which == 0 ? s0f : s0r) < 0 ? 8 : 9, lane);
#endif
					const int r = primer_settle(ltab, wb, which == 0 ? L::F : L::R, which == 0 ? flen : rlen, (which == 0 ? primer_bases : primer_bases >> 8) & 0xFFu, which == 0 ? pkf : pkr, which == 0 ? fpl : rpl, which == 0 ? thr_f : thr_r, inv_err_t, lane, which == 0 ? s0f : s0r, which == 0 ? sumf : sumr);
					if (which == 0)
						rf = r;
					else
						rr = r;
				}
				if (rf == -1) {
					unsettled = true;
				} else if (rf == 0) {
					status = PAMD_ST_NOFP;
				} else {
					if (rf > 0)
						fo = rf - 1;
					if (rr == -1)
						unsettled = true;
					else if (rr == 0)
						status = PAMD_ST_NORP;
					else if (rr > 0)
						ro = rr - 1;
				}
			}
			if (unsettled) {	// the primer kernel and the general kernel take it from here
				note(P, 6, lane);
				status = PAMD_ST_DEFER;
				fo = ro = 0;
				break;
			}
			if (status != PAMD_ST_OK) {
				if (status == PAMD_ST_NOFP)
					fo = 0;
				ro = 0;
				break;
			}
			if ((flen < rlen ? flen : rlen) < minov) {	// assembler.c:289-292
				status = PAMD_ST_BADR;
				break;
			}
			if (minov + fo >= flen || minov + ro >= rlen) {	// assembler.c:73-82
				status = PAMD_ST_NOALGN;
				break;
			}
			primers_known = true;
			wave_sync();	// the scan's streams and scratch lie where the k-mer tables are cleared now
		}
		{	// clear the k-mer tables and the candidate flags
			// (the first two tables are whole kilobytes: straight-line stores, no loop counter; the last two, 384 bytes, only
			// when the previous pair used them)
			static_assert((G::T3 - L::T1) % 1024u == 0 && (G::WAVE_BYTES - G::T3) / 16u <= 64u, "whole rounds of 64 lanes x 16 bytes");
			const bool clear_now = !EP || epoch == 0u || deep_dirty;	// (EP: deep_dirty = something was laid over the tables)
			if (clear_now) {
				uint4 *h = reinterpret_cast<uint4 *>(wb + L::T1) + lane;
				unsigned e = EMPTY;
				asm volatile("" : "+v"(e));	// (made here: hoisted out of the pair loop the four registers of the constant are spilled and reloaded per pair)
				const uint4 e4 = make_uint4(e, e, e, e);
#pragma unroll
				for (unsigned i = 0; i < (G::T3 - L::T1) / 1024u; i++)
					h[64u * i] = e4;
				if ((EP || deep_dirty) && lane < (int) ((G::WAVE_BYTES - G::T3) / 16u))
					reinterpret_cast<uint4 *>(wb + G::T3)[lane] = e4;
			}
			if (EP)
				epoch = clear_now ? 126u : epoch - 1u;
			if ((!FK_LAZY_FLAGS || flags_dirty) && lane < (LONG ? 24 : 16)) {	// (FK_LAZY_FLAGS: only a pair whose candidates went through the flag bytes leaves any set)
				unsigned z = 0u;
				asm volatile("" : "+v"(z));	// (likewise)
				reinterpret_cast<uint4 *>(flags)[lane] = make_uint4(z, z, z, z);
			}
			flags_dirty = false;
			deep_dirty = false;
		}
		wave_sync();
		STOP(0);
		MARK("seed-fwd");

		// ---- seed: forward k-mers (assembler.c:92-101) ----
		// This lane's four positions p = 4*lane + j.  The k-mer ending at p covers bases p-7 .. p and is emitted iff
		// p >= 8 and none of the nine bases p-8 .. p is N (misc.h:41): both windows lie in the codes of lanes l-2, l-1, l
		// (what lies before the read, like what lies behind it, reads as N).
		// The largest overlap looked at: the shorter read's length (assembler.c:79-80), or a configured maximum (assembler.c:59,
		// 81-82): the smaller of it and flen + rlen - minov - offsets - 1.  Up to the shorter read's length that is this kernel's
		// arithmetic with a smaller `size`; beyond it an overlap runs past a read's end (assembler.c:162,181,231's VEEZ / WEDGEZ),
		// and below minov the bit list degenerates: the general kernel's pairs.  (Scalar selects, and the verdict rides on the
		// test for a k-mer without a slot below: a branch of its own costs the pair loop two scalar registers it does not have.)
		const int shorter = flen < rlen ? flen : rlen, room = flen + rlen - minov - fo - ro - 1;
		const int cap = P.maxoverlap != 0 ? (P.maxoverlap < room ? P.maxoverlap : room) : shorter;
		const bool beyond = cap > shorter || cap < minov;
		const int maxov = beyond ? shorter : cap;	// (a pair that is handed on is seeded with the usual range: nothing of it is kept)
		const int size = maxov - minov + 1;		// minov <= maxov here
		bool deep = false;
		{
			unsigned w, nw;	// base BPL*l - 8 + b at bits 2b..2b+1 of w, its N flag at bit 2b of nw
			if constexpr (WINDOWS) {
				// BPL positions per lane, p = BPL*lane + j: the sixteen bases from BPL*lane - 8 are one unaligned word of the
				// code stream and of the N map (before the read: the pad word, all N)
				const int fb = 2 * BPL * lane - 16;
				const unsigned *cw = reinterpret_cast<const unsigned *>(wb + L::FC) + (fb >> 5), *nwp = reinterpret_cast<const unsigned *>(wb + L::FN) + (fb >> 5);
				const unsigned c0 = cw[0], c1 = cw[1], n0 = nwp[0], n1 = nwp[1];
				w = __builtin_amdgcn_alignbit(c1, c0, (unsigned) fb & 31u);
				nw = __builtin_amdgcn_alignbit(n1, n0, (unsigned) fb & 31u);
			} else {
				const unsigned x0 = fcodes | (fnm << 8);
				const unsigned x1 = from_lane_below<0x5500u>(x0), x2 = from_lane_below<0x5500u>(x1);
				w = (x2 & 0xFFu) | ((x1 & 0xFFu) << 8) | ((x0 & 0xFFu) << 16);
				nw = ((x2 >> 8) & 0xFFu) | (x1 & 0xFF00u) | ((x0 & 0xFF00u) << 8);
			}
			unsigned carry[BPL];
			const unsigned limit = EP ? (epoch + 1u) << 25 : EMPTY;	// a word below it is live: this pair's (EP), not EMPTY
			const unsigned l1 = lds_addr(t1), l2 = lds_addr(t2);
			if constexpr (EP) {
				unsigned key[BPL];
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					// the k-mer's eight codes are bits 2+2j .. 17+2j of w; the key is epoch | k-mer, the word key | position
					key[j] = __builtin_amdgcn_alignbit(epoch, w << (14 - 2 * j), 16);
					const unsigned word = (key[j] << 9) | (unsigned) (BPL * lane + j);
					carry[j] = (nw & (0x15555u << (2 * j))) == 0 ? word : EMPTY;
				}
				unsigned addr[BPL];
#pragma unroll
				for (int j = 0; j < BPL; j++)
					addr[j] = BUCK ? lshl_add<3>(mul24(key[j], HASH_C) >> (33 - T1B), l1) : lshl_add<2>(mul24(key[j], HASH_C) >> (32 - T1B), l1);
				// (FK_INS_PLAIN: only a read that fills the lanes -- the words of idle lanes, all of one k-mer, would meet on one slot:
				// measured on reads of 200 bases in these rows of 256, fourteen idle lanes: 4 % lost instead of 1.5 % won)
				const bool full = FK_INS_PLAIN_ALWAYS || flen > BPL * 60;
				if constexpr (FK_INS_PLAIN >= 1 && BPL == 4) {
					if (full)
						insert_all_plain<0>(addr, carry);
					else
						insert_all_live<BPL>(addr, carry, limit);
				} else {
					insert_all_live<BPL>(addr, carry, limit);
				}
				if constexpr (BUCK) {	// what the bucket's first slot displaced, to its second
					if constexpr (FK_INS_PLAIN >= 2 && BPL == 4) {
						if (full)
							insert_all_plain<4>(addr, carry);
						else
							insert_all_live<BPL, 4>(addr, carry, limit);
					} else {
						insert_all_live<BPL, 4>(addr, carry, limit);
					}
				}
#pragma unroll
				for (int j = 0; j < BPL; j++)
					addr[j] = lshl_add<2>(__builtin_amdgcn_ubfe(mul24(carry[j] >> 9, HASH_C), 14, 8), l2);
				insert_all_live<BPL>(addr, carry, limit);	// (the second table without the test: eight vector instructions fewer, and -1.5 %: its 256 slots take 256 atomics a pair badly)
			} else {
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					// the k-mer's eight codes are bits 2+2j .. 17+2j of w: shifted to the top half of the table word
					const unsigned word = ((w << (14 - 2 * j)) & 0xFFFF0000u) | (unsigned) (BPL * lane + j);
					carry[j] = (nw & (0x15555u << (2 * j))) == 0 ? word : EMPTY;
				}
				unsigned addr[BPL];
#pragma unroll
				for (int j = 0; j < BPL; j++)
					addr[j] = BUCK ? lshl_add<3>(mul24(carry[j] >> 16, HASH_C) >> (33 - T1B), l1) : lshl_add<2>(mul24(carry[j] >> 16, HASH_C) >> (32 - T1B), l1);
				insert_all<BPL>(addr, carry);
				if constexpr (BUCK) {
					// (the second slot's word has another k-mer, usually: its hash is taken below, from what moved on)
					insert_all<BPL, 4>(addr, carry);
				}
#pragma unroll
				for (int j = 0; j < BPL; j++)
					addr[j] = lshl_add<2>(__builtin_amdgcn_ubfe(mul24(carry[j] >> 16, HASH_C), 14, 8), l2);
				insert_all<BPL>(addr, carry);
			}
			bool lost = beyond;
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				// (measured and not kept, round 4: all of a lane's words sent through the last two tables under one test, one
				// round trip per level -- every pair that needs them for one word then pays the issue for four: -2.5 %)
				if (ballot(carry[j] < limit) != 0) {	// rare from here on: one or two words per pair
					deep = true;
					carry[j] = insert_level(t3, slot3(carry[j] >> KSH), carry[j], limit);
					carry[j] = insert_level(t4, slot4(carry[j] >> KSH), carry[j], limit);
					if (ballot(carry[j] < limit) != 0) {
						// A word found no slot.  The words of one k-mer that did are always its smallest positions (a slot
						// keeps its minimum), so a k-mer with two of them in the tables loses nothing the reference keeps
						// (assembler.c:94-99 drops a third position as well): tandem repeats pass here.
						const unsigned k = carry[j] >> KSH, kk = carry[j] & ~(POS_END - 1u);
						const unsigned s1 = BUCK ? (slot1<T1B>(k) & ~1u) : slot1<T1B>(k);
						const int kept = (int) ((t1[s1] ^ kk) < POS_END) + (BUCK ? (int) ((t1[s1 + 1u] ^ kk) < POS_END) : 0) + (int) ((t2[slot2(k)] ^ kk) < POS_END) + (int) ((t3[slot3(k)] ^ kk) < POS_END) + (int) ((t4[slot4(k)] ^ kk) < POS_END);
						lost |= ballot(carry[j] < limit && kept < 2) != 0;
					}
				}
			}
			deep_dirty = EP ? false : deep;
			DEBUG_DUMP(P, wb, pair, lane, 0, uf, ur, (int) deep, (int) lost, (int) carry[0], (int) carry[1]);
#ifdef PAMD_FK_DEBUG
			if (deep)
				note(P, 5, lane);	// (two pairs in three: far too many same-address atomics for a shipped build)
#endif
			if (__builtin_expect(lost, 0)) {	// a k-mer found no slot (tandem repeats do this): the general kernel's unbounded table
				note(P, 1, lane);
				status = PAMD_ST_DEFER;
				break;
			}
		}
		wave_sync();
		STOP(1);
		MARK("seed-rev");
		int ncand = 0, ov0 = 0, ov1 = 0, ov2 = 0, ov3 = 0;
		bool all = false;
		int quick = 0;	// (an int: a wave-uniform bool that is set in nested branches comes out of them as a lane mask made from a 0 / 1 vector register)
		// ---- seed: reverse k-mers flag candidate overlaps (assembler.c:104-110) ----
		// The k-mer at p covers bases p .. p+7 and is emitted iff p <= rlen-9 and none of p .. p+8 is N: the codes of
		// lanes l, l+1, l+2.  Its two smallest forward positions are the reference's first two (kmerseen[2k], [2k+1]).
		{
			unsigned w, nw;	// base BPL*l + b at bits 2b..2b+1 of w, its N flag at bit 2b+1 of nw
			if constexpr (WINDOWS) {
				const int rb = 2 * BPL * lane;
				const unsigned *cw = reinterpret_cast<const unsigned *>(wb + L::RC) + (rb >> 5), *nwp = reinterpret_cast<const unsigned *>(wb + L::RN) + (rb >> 5);
				const unsigned c0 = cw[0], c1 = cw[1], n0 = nwp[0], n1 = nwp[1];
				w = __builtin_amdgcn_alignbit(c1, c0, (unsigned) rb & 31u);
				nw = __builtin_amdgcn_alignbit(n1, n0, (unsigned) rb & 31u);
			} else {
				const unsigned x0 = rcodes | (rnm << 8);
				const unsigned x1 = from_lane_above<0xAA00u>(x0), x2 = from_lane_above<0xAA00u>(x1);
				w = (x0 & 0xFFu) | ((x1 & 0xFFu) << 8) | ((x2 & 0xFFu) << 16);
				nw = ((x0 >> 8) & 0xFFu) | (x1 & 0xFF00u) | ((x2 & 0xFF00u) << 8);
			}
			const unsigned br = __builtin_bitreverse32(w);	// base b now at bits 30-2b..31-2b, codes read as forward codes
			const unsigned top = (unsigned) (flen + rlen - minov - 1 - BPL * lane);	// assembler.c:107: index = top - j - forward position
			unsigned m1[BPL], m2[BPL], xk[BPL];
			bool ok[BPL];
			const unsigned l1 = lds_addr(t1), l2 = lds_addr(t2);
			// (measured and not kept, round 6: the second insert level's atomics left in flight -- bare `ds_min_rtn`, their returns collected
			// behind these lookups, which read the tables' state in issue order whether or not the atomics have returned: one exposed
			// round trip fewer, and -1.7 % on config 1, -1 % on config 4: the deep-table inserts then wait behind the lookups)
			// (the compiler issues these 2 BPL reads k-mer by k-mer, three LDS round trips; all eight under way before the first
			// use -- bare reads, one wait -- measured in round 4: nothing on configs 1, 2 and 4, -1.5 % on config 3; as volatile
			// reads in round 6: +-0.5 % on configs 1 and 4.  The other waves cover it.)
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				xk[j] = br << (2 * j);	// this k-mer in the top half
				ok[j] = (nw & (0x2AAAAu << (2 * j))) == 0;
				if constexpr (EP)
					xk[j] = __builtin_amdgcn_alignbit(epoch, xk[j], 16);	// the key, epoch | k-mer: what this pair's words of the k-mer carry above their position
				const unsigned prod = mul24(EP ? xk[j] : xk[j] >> 16, HASH_C), kk = EP ? xk[j] << 9 : xk[j] & 0xFFFF0000u;
				const unsigned e2 = lds_read_u32(lshl_add<2>(__builtin_amdgcn_ubfe(prod, 14, 8), l2)) ^ kk;
				// an entry of this k-mer reads as its bare position (< 0x10000), anything else as >= 0x10000
				if constexpr (BUCK) {
					const u32x2 b = *reinterpret_cast<__attribute__((address_space(3))) const u32x2 *>(lshl_add<3>(prod >> (33 - T1B), l1));	// the bucket, one 8-byte read
					const unsigned ea = b.x ^ kk, eb = b.y ^ kk;
					m1[j] = umin3(ea, eb, e2);	// the smallest and the second smallest of three
					m2[j] = umed3(ea, eb, e2);
				} else {
					const unsigned e1 = lds_read_u32(lshl_add<2>(prod >> (32 - T1B), l1)) ^ kk;
					m1[j] = e1 < e2 ? e1 : e2;
					m2[j] = e1 < e2 ? e2 : e1;
				}
			}
			if (deep) {	// wave-uniform: the last two tables hold something
#if FK_DEEP_BATCH
				// (all 2 BPL reads under way before the first is used: left to itself the compiler reads k-mer by k-mer, BPL exposed round trips)
				unsigned r3[BPL], r4[BPL];
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					const unsigned kmer = EP ? xk[j] : xk[j] >> 16;
					r3[j] = lds_read_u32_early(lshl_add<2>(slot3(kmer), lds_addr(t3)));
					r4[j] = lds_read_u32_early(lshl_add<2>(slot4(kmer), lds_addr(t4)));
				}
#endif
#pragma unroll
				for (int j = 0; j < BPL; j++) {
					const unsigned kmer = EP ? xk[j] : xk[j] >> 16, kk = EP ? xk[j] << 9 : xk[j] & 0xFFFF0000u;
#if FK_DEEP_BATCH
					const unsigned e3 = r3[j] ^ kk, e4 = r4[j] ^ kk;
					(void) kmer;
#else
					const unsigned e3 = t3[slot3(kmer)] ^ kk, e4 = t4[slot4(kmer)] ^ kk;
#endif
					const unsigned l34 = e3 < e4 ? e3 : e4, h34 = e3 < e4 ? e4 : e3;
					const unsigned hi = m1[j] < l34 ? l34 : m1[j];
					m1[j] = m1[j] < l34 ? m1[j] : l34;
					const unsigned h = m2[j] < h34 ? m2[j] : h34;
					m2[j] = hi < h ? hi : h;
				}
			}
			u64 second = 0;
			unsigned idx1[BPL];	// the overlap (less minov) this k-mer's first forward position votes for; out of range: none
#pragma unroll
			for (int j = 0; j < BPL; j++) {
				const unsigned idx = top - (unsigned) j - m1[j];	// no match: m1 >= 0x10000 wraps far beyond size
				idx1[j] = ok[j] && idx < (unsigned) size ? idx : EMPTY;
				second |= ballot(ok[j]) & ballot(m2[j] < (EP ? POS_END : 0xFFFFu));	// (an empty slot reads as 0xFFFF for the k-mer 0xFFFF.  Two compares and scalar logic: the ballot of a conjunction goes through a vector register)
			}
			const bool twice = second != 0;	// a k-mer that occurs twice in the forward read
			// Nearly every pair's k-mers vote for one overlap, or for that and a stray one: settled here in registers, the
			// candidates need neither the flag bytes nor the walk over them.
			if (__builtin_expect(!twice, 1)) {
				unsigned mine = EMPTY;
#pragma unroll
				for (int j = BPL - 1; j >= 0; j--)
					mine = idx1[j] != EMPTY ? idx1[j] : mine;
				const u64 have = ballot(mine != EMPTY);
				if (__builtin_expect(have != 0, 1)) {
					const unsigned c0 = (unsigned) __builtin_amdgcn_readlane((int) mine, __builtin_ctzll(have));
					unsigned other = EMPTY;
#pragma unroll
					for (int j = 0; j < BPL; j++)
						other = (idx1[j] != EMPTY && idx1[j] != c0) ? idx1[j] : other;
					const u64 oth = ballot(other != EMPTY);
					if (__builtin_expect(oth == 0, 1)) {
						ncand = 1;
						ov0 = (int) c0 + minov;
						quick = 1;
					} else {
						const unsigned c1 = (unsigned) __builtin_amdgcn_readlane((int) other, __builtin_ctzll(oth));
						bool third = false;
#pragma unroll
						for (int j = 0; j < BPL; j++)
							third |= idx1[j] != EMPTY && idx1[j] != c0 && idx1[j] != c1;
						if (ballot(third) == 0) {
							ncand = 2;
							ov0 = (int) (c0 < c1 ? c0 : c1) + minov;	// ascending, as the scan walks them (assembler.c:120)
							ov1 = (int) (c0 < c1 ? c1 : c0) + minov;
							quick = 1;
						}
					}
				}
			}
			if (__builtin_expect(!quick, 0)) {
				flags_dirty = true;
#pragma unroll
				for (int j = 0; j < BPL; j++)
					if (idx1[j] != EMPTY)
						flags[idx1[j]] = 1;
				if (twice) {
#pragma unroll
					for (int j = 0; j < BPL; j++) {
						const unsigned idx = top - (unsigned) j - m2[j];
						if (ok[j] && idx < (unsigned) size)
							flags[idx] = 1;
					}
				}
			}
		}
		STOP(2);
		MARK("candidates");

		// ---- the candidates (assembler.c:118-143) ----
		if (__builtin_expect(!quick, 0)) {
			wave_sync();
			const unsigned mine = reinterpret_cast<const unsigned *>(flags)[lane];	// overlaps minov + 4*lane + 0..3
			// (kept as scalar selects: indexed by ncand the four overlaps would live in scratch memory)
#define FK_TAKE(NZ, MINE, BASE)                                                                  \
	{                                                                                        \
		u64 nz_ = (NZ);                                                                  \
		while (nz_ != 0 && ncand <= 4) {                                                 \
			const int w = __builtin_ctzll(nz_);                                      \
			nz_ &= nz_ - 1;                                                          \
			unsigned m = (unsigned) __builtin_amdgcn_readlane((int) (MINE), w);      \
			while (m != 0 && ncand <= 4) {                                           \
				const int k = __builtin_ctz(m) >> 3;                             \
				m &= ~(0xFFu << (8 * k));                                        \
				const int ov = (BASE) + 4 * w + k + minov;                       \
				if (ncand == 0)                                                  \
					ov0 = ov;                                                \
				else if (ncand == 1)                                             \
					ov1 = ov;                                                \
				else if (ncand == 2)                                             \
					ov2 = ov;                                                \
				else                                                             \
					ov3 = ov;                                                \
				ncand++;                                                         \
			}                                                                        \
		}                                                                                \
	}
			const u64 nz = ballot(mine != 0);
			all = nz == 0;	// ALL_BITS_IF_NONE: every overlap is a candidate
			FK_TAKE(nz, mine, 0);
			if constexpr (LONG) {	// overlaps beyond 256: lanes 0..15 once more
				const unsigned mine2 = reinterpret_cast<const unsigned *>(flags + 256)[lane & 31];
				const u64 nz2 = ballot(mine2 != 0) & 0xFFFFull;
				all = all && nz2 == 0;
				if (nz2 != 0)
					FK_TAKE(nz2, mine2, 256);
			}
#undef FK_TAKE
		}
		MARK("scan");
#if FK_KARGS
		best = H.qual_nn * (double) (flen + rlen);	// assembler.c:60
#else
		best = P.qual_nn * (double) (flen + rlen);	// assembler.c:60
#endif
		if (!counting_algo<ALGO>()) {
			// ---- pear / rdp_mle: an overlap's score is a sum of per-base table values (algo_pear.c:32-59,
			// algo_rdp_mle.c:43-74).  Lanes produce the terms.  Exact sums: lane 0 adds them in the reference's order.
			// Any-order sums: every lane adds its own, the wave adds the lanes (|error| < 1e-9), and a comparison closer
			// than 1e-6 -- or a min_overlapbits verdict that close -- sends the pair to the general kernel, which sums in
			// reference order.
			const double nrb = -P.pear_random_base, nn = P.qual_nn;
			bool ambiguous = false;
			int nloop = ncand;	// candidates the loop below scores
			if (all || ncand > 4) {
				// ---- many candidates (no seeded overlap, tandem repeats): scored best bound first (scan_many_sums) ----
				const ManySums ms = scan_many_sums<ALGO, LONG>(wb, ltab, all ? 1 : 0, size, minov, flen, rlen, nrb, nn, P.many_all, P.many_match, P.many_strong, best, lane);
				ncand = uni(ms.examined);
				if (uni(ms.ambiguous) != 0) {
					note(P, 2, lane);
					status = PAMD_ST_DEFER;
					break;
				}
				nloop = 0;
				if (uni(ms.bo) >= 0) {
					best = uni_f64(ms.best);
					bo = uni(ms.bo);
				}
				wave_sync();
				deep_dirty = true;	// (the candidate list lay over the k-mer tables)
			}
			const unsigned lF = lds_addr(wb + L::F), lR = lds_addr(wb + L::R), lT = lds_addr(ltab);
			for (int c = 0; c < nloop; c++) {
				const int ov = c == 0 ? ov0 : (c == 1 ? ov1 : (c == 2 ? ov2 : ov3));
				double part = 0.0;
#pragma unroll
				for (int r = 0; r < L::SR; r++) {
					const int i = 64 * r + lane;
					if (64 * r < ov && i < ov) {
						const int fi = flen - ov + i, ri = rlen - 1 - i;
						const unsigned ef = lds_read_u16(lF + 2u * (unsigned) fi), er = lds_read_u16(lR + 2u * (unsigned) ri);
						const unsigned a = ef & 0xFFu, b = er & 0xFFu, qa = ef >> 8;
						const bool agree = (a & b) != 0;
						double term;
						if (ALGO == PAMD_ALGO_PEAR) {
							// the second index is the FORWARD read's quality at the reverse index (algo_pear.c:52,54, sic);
							// past the forward read the reference reads beyond the array: defined as quality 0 here
							const unsigned qb = ri < flen ? lds_read_u16(lF + 2u * (unsigned) ri) >> 8 : 0u;
							term = (a == 15u || b == 15u) ? nrb : lds_read_f64(lT + 8u * mat_index(agree, qa, qb));
						} else {
							term = lds_read_f64(lT + 8u * mat_index(agree, qa, er >> 8)) - nn;
						}
						part += term;
					}
				}
				{
					const double pr = wave_sum_any_order(part);
					if (pr > best + 1e-6) {
						best = pr;
						bo = ov;
					} else if (!(pr < best - 1e-6)) {
						ambiguous = true;
					}
				}
			}
			if (!EXACT) {	// (with sums in the reference's order the module sees the ordered sum itself: exact_sums_group delivers it before finish64)
				for (int mi = 0; mi < P.n_modules; mi++)	// plugin_min_overlapbits.c:17-23 reads this sum
					ambiguous |= P.modules[mi].kind == 2 && fabs(P.modules[mi].nats - best) <= 1e-6;
			}
			if (ambiguous) {
				note(P, 2, lane);
				status = PAMD_ST_DEFER;
				break;
			}
			bo = uni(bo);
			best = uni_f64(best);
			if (EXACT && bo >= 0) {
				// Sums in the reference's order (round 5): the DECISION above is made from any-order sums behind the same 1e-6 guard
				// as in the other mode (what is closer goes to the general kernel, which adds in order); the SCORE the record reports is
				// the winner's terms added in order -- by a lane of the group walk, from their table entries left here beside the row's.
				typedef __attribute__((address_space(1))) uint16_t gbl_u16s;
				gbl_u16s *const sc = (gbl_u16s *) (xs + (unsigned) L::NR * 64u + (unsigned) lane);
#pragma unroll
				for (int r = 0; r < L::SR; r++) {
					const int i = 64 * r + lane;
					if (64 * r < bo && i < bo) {
						const int fi = flen - bo + i, ri = rlen - 1 - i;
						const unsigned ef = lds_read_u16(lF + 2u * (unsigned) fi), er = lds_read_u16(lR + 2u * (unsigned) ri);
						const unsigned a = ef & 0xFFu, b = er & 0xFFu, qa = ef >> 8;
						const bool agree = (a & b) != 0;
						unsigned entry;
						if (ALGO == PAMD_ALGO_PEAR) {
							const unsigned qb = ri < flen ? lds_read_u16(lF + 2u * (unsigned) ri) >> 8 : 0u;
							entry = (a == 15u || b == 15u) ? (unsigned) PAMD_LT_C : mat_index(agree, qa, qb);	// (the slot behind the matrices holds -random_base: the kernel's prologue)
						} else {
							entry = mat_index(agree, qa, er >> 8);
						}
						sc[64 * r] = (uint16_t) (8u * entry);
					}
				}
			}
		} else if (LONG && (all || ncand > 4)) {	// (the candidate-per-lane scan below is laid out for 256 overlaps)
			note(P, 3, lane);
			status = PAMD_ST_DEFER;
			break;
		} else if (!LONG && __builtin_expect(all || ncand > 4, 0)) {
			// ---- pairs without a seeded overlap, tandem repeats: a candidate per lane (three pairs in a hundred) ----
			const ManyResult mr = scan_many<ALGO>(wb, all ? 1 : 0, size, minov, flen, rlen, P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, best, lane);
			best = uni_f64(mr.best);	// a function's results come back in vector registers
			bo = uni(mr.bo);
			ncand = uni(mr.examined);
			deep_dirty = true;	// the candidate list lay over the k-mer tables
		} else {
			// ---- scan + argmax: one candidate per 16-lane row, 16 positions per lane ----
			const int row = lane >> 4;
			// (the row's candidate out of a scalar pair that packs the four: a shift and a mask per lane -- three selects on
			// row masks cost the masks' scalar registers, which the kernel spills into vector lanes and reads back per pair)
			const u64 ovpack = (u64) (unsigned) ov0 | ((u64) (unsigned) ov1 << 16) | ((u64) (unsigned) ov2 << 32) | ((u64) (unsigned) ov3 << 48);
			const int myov = (int) ((unsigned) (ovpack >> (unsigned) (lane & 0x30)) & 0xFFFFu);
			unsigned packed = word_counts<L>(wb, flen, rlen, myov, 16 * (lane & 15), row < ncand);
			if (LONG && (ov0 > 256 || ov1 > 256 || ov2 > 256 || ov3 > 256))	// (a row of lanes covers 256 positions)
				packed += word_counts<L>(wb, flen, rlen, myov, 256 + 16 * (lane & 15), row < ncand);
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x128, 0xF, 0xF, false);	// row_ror:8
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x124, 0xF, 0xF, false);	// row_ror:4
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x122, 0xF, 0xF, false);	// row_ror:2
			packed += (unsigned) __builtin_amdgcn_mov_dpp((int) packed, 0x121, 0xF, 0xF, false);	// row_ror:1
			const int unknowns = (int) (packed & 1023u), matches = (int) (packed >> 10);
#if FK_KARGS
			const double sc = count_score<ALGO, true>(H.qual_nn, H.pmatch, H.pmismatch, P.count_table, flen, rlen, myov, unknowns, matches, myov);
#else
			const double sc = count_score<ALGO, true>(P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.count_table, flen, rlen, myov, unknowns, matches, myov);
#endif
			if (__builtin_expect(ncand == 1, 1)) {	// (nine pairs in ten: one compare, no walk over the rows)
				const double pr = lane_f64(sc, 0);
				if (pr > best) {
					best = pr;
					bo = ov0;
				}
			} else {
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
			}
			// wave-uniform by construction; tell the compiler, so that everything derived from them is scalar
			bo = uni(bo);
			best = uni_f64(best);
		}
		examined = ncand;
		slow = examined == size;	// assembler.c:135
		DEBUG_DUMP(P, wb, pair, lane, 2, uf, ur, ncand, ov0, ov1, bo);
		if (__builtin_expect(bo < 0, 0)) {
			status = PAMD_ST_NOALGN;
			best = 0.0;	// (the record of a pair that did not align carries no score, as in the general kernel)
			break;
		}

		// ---- merge (assembler.c:146-247): df = flen - fo - bo forward-only bases, dr = rlen - ro - bo reverse-only ones ----
		// (with clipped reads either may be negative -- the overlap runs into what was clipped, VEEZ / WEDGEZ of
		// assembler.c:162,181,231 --: the overlap region then starts at 0 and is that much shorter, the index formulas stay)
		const int A = flen - fo - bo, Bn = rlen - ro - bo, novl = bo;
		if (PRIM && A + novl + Bn < 0) {	// assembler.c:146-150: nothing is left of the pair
			status = PAMD_ST_NOALGN;
			best = 0.0;
			bo = -1;	// (the record carries no overlap, as the general kernel's)
			break;
		}
		STOP(3);
		// (the launch parameters the allocation below needs, read together and FIRST: scalar loads and LDS reads share a counter,
		// so a wait for these after the reads below would wait for those as well; and the compiler otherwise re-reads each
		// from the kernel argument segment, one exposed scalar-memory round trip per use, in whichever branch it is used)
#if FK_KARGS
		const u64 k_stride = H.stride, k_capacity = H.capacity;
		uint8_t *const k_nt = H.nt;
		double *const k_p = H.p;
#else
		u64 k_stride = P.seq_stride, k_capacity = P.seq_capacity;
		uint8_t *k_nt = P.seq_nt;
		double *k_p = P.seq_p;
		asm volatile("" : "+s"(k_stride), "+s"(k_capacity), "+s"(k_nt), "+s"(k_p));
#endif
		// The staged bases of all rounds of the merge (below), requested before the register rotation and the row's allocation:
		// sixty scalar instructions that need nothing from the LDS run while the reads are under way.
		// (NARROW: reads of at most 192 bases: a row spans six rounds, a forward base lies in the first three)
		constexpr int NR = NARROW ? 6 : L::NR;
		constexpr int NF = NARROW ? 3 : L::MAXLEN / 64;	// rounds that can hold a forward base (a read is at most MAXLEN bases long)
		unsigned ef[NF], er[NR];
		{
			const unsigned aF = lds_addr(wb + L::F) + 2u * (unsigned) (lane + fo);					// round r: F[fo + 64r + lane] ...
			const unsigned aR = lds_addr(wb + L::R) + 2u * (unsigned) (rlen - 1 + A - lane) - 128u * (NR - 1);	// ... and R[rlen - 1 - (64r + lane - A)]
			auto read_round = [&](auto rc) __attribute__((always_inline)) {
				constexpr int r = decltype(rc)::value;
				if constexpr (r < NF)
					ef[r] = lds_read_u16_bare<128 * r>(aF);
				er[r] = lds_read_u16_bare<128 * (NR - 1 - r)>(aR);
			};
			for_each_int(std::make_integer_sequence<int, NR>(), read_round);
		}
		MARK("rotate+alloc");
		// Rotate the prefetched registers in BEFORE this pair's stores are issued (vmcnt is one in-order queue for
		// loads and stores: waiting for the loads at the top of the next pair would also wait for the stores).
		cur = nxt;
		settle(cur);
		rotated = true;

		seqlen = A + novl + Bn;	// len - 1; len = seqlen + 1 is in 3 .. 511 (639)
		const unsigned need16 = ((unsigned) seqlen + 15u) >> 4;
		if (k_stride != 0) {
			off = pair * k_stride;
		} else {
			if (__builtin_expect(arena.cur + need16 > arena.end, 0)) {
				u64 got = 0;
				if (lane == 0) {
					// (the instruction itself: given atomicAdd() the compiler's atomic optimizer wraps it in a wave-wide prefix sum whose
					// 64-bit multiply-add reads a register pair with an undefined upper half -- harmless, but that half can be a
					// register with one of the bare LDS reads above pending, which tools/isa_hazards.py rightly refuses to tell apart)
					u64 chunk = P.arena_chunk;
					unsigned long long *cursor = P.seq_cursor;
					asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(cursor), "v"(chunk) : "memory");
				}
				const unsigned chunk16 = (unsigned) (P.arena_chunk >> 4);
				arena.cur = (unsigned) uni((int) (unsigned) (got >> 4));	// (the cursor only ever moves by whole chunks: a multiple of 16)
				arena.end = arena.cur + chunk16;
			}
			off = (u64) arena.cur << 4;
			arena.cur += need16;
			if (off + 16ull * need16 > k_capacity)
				off = ~0ull;
		}
		const bool store = off != ~0ull;
		uint8_t *const row_nt = k_nt + (store ? off : 0);
		double *const row_p = k_p + (store ? off : 0);
		double qacc = 0.0;	// this lane's share of the quality sum, any order
		// PHRED: panda_result_phred (nt.c:126-150) answers 1 at or below qual_score[0], 0 between qual_score[0] and [1] (sic) and
		// does not decrease from there on: the row's smallest phred follows from "any log-p at or below qual_score[0]" and the
		// smallest log-p ABOVE it -- not from the smallest log-p alone (a row holding both -3.7 and -1.8 has a phred 0)
		double pmin = 0.0;	// this lane's smallest log-p above qual_score[0] (every log-p is <= 0)
		double pall = 0.0;	// ... and its smallest log-p of all: at or below qual_score[0] if it saw such a one
		const double qs0 = PHRED ? ltab[PAMD_LT_SCORE] : 0.0;
		// One pass over the output row, 64 bases per round: lane l of round r makes base j = 64r + l, from forward base j
		// (j < flen: the forward-only bases [0, A), assembler.c:162-173, and the overlap [A, A + novl), assembler.c:181-228) and
		// from reverse base rlen - 1 - (j - A) (j >= A: the overlap and the reverse-only bases [A + novl, seqlen),
		// assembler.c:231-243): the same two index formulas in all three regions, so a 380-base row takes six rounds and every
		// round writes 512 aligned bytes of log-p.  The staged bases of all rounds are read first (one wait); then, two rounds
		// per trip, each round by its kind -- wave-uniform: all forward-only, all reverse-only (a shift and an add), or mixed
		// (the overlap's table index and base, with per-lane selects at the region borders) -- table reads, then the writes.
		// A lane past the end of the row reads whatever lies there, inside the LDS, and its base is forced to 0.
		const unsigned lT = lds_addr(ltab), lN = lds_addr(nts);
		// a pair whose row does not fit stores nothing (it ends as NOSPACE whatever its sums are)
#ifdef FK_NO_PSTORE	/* experiment: what do the log-p stores cost? (results are wrong) */
		const int lim = 0;
		qacc = -1.0;
#else
		const int lim = store ? seqlen : 0;
#endif
		// the B-cliff cases (assembler.c:192-207) only when a read ends in a run of quality 2
		const bool cliff = uf < flen || ur < rlen;
		const int A0 = PRIM && A < 0 ? 0 : A;	// the overlap region starts here ...
		const int AF = PRIM ? A0 + novl + (A < 0 ? A : 0) + (Bn < 0 ? Bn : 0) : A + novl;	// ... and ends here (assembler.c:181)
		region_a0 = A0;
		region_af = AF;
		region_len = A + novl + Bn;	// (= seqlen; `-a` clips seqlen later, the chains and the divisor are the whole merge's)
		const unsigned nN = lN + (unsigned) lane;
		// (explicitly global: through a generic pointer these would be flat_store, which every later LDS wait waits for as well)
		gbl_f64 *const gP = (gbl_f64 *) (row_p + lane);
		typedef __attribute__((address_space(1))) uint16_t gbl_u16;
		gbl_u16 *const xsP = (gbl_u16 *) (xs + (unsigned) lane);	// (xs: this pair's scratch row)
		lds_wait_all(ef);
		lds_wait_all(er);
		MARK("merge");
		// one round: table index and base of output base 64 R + lane
		auto round = [&](auto rc, unsigned &qi_out, unsigned &nt_out) __attribute__((always_inline)) {
			constexpr int R = decltype(rc)::value;
			const unsigned e_r = er[R];
			const bool ok = lane < lim - 64 * R;
			if (R >= NF || 64 * R >= AF) {	// all reverse-only (assembler.c:231-243); past MAXLEN bases nothing else can be
				qi_out = PAMD_LT_SCORE + (e_r >> 8);
				nt_out = ok ? e_r : 0u;
				return;
			}
			const unsigned e_f = ef[R < NF ? R : 0];
			if (64 * (R + 1) <= A0) {	// all forward-only (assembler.c:162-173)
				qi_out = PAMD_LT_SCORE + (e_f >> 8);
				nt_out = e_f;
				return;
			}
			// (the round kinds of the first trip decided once per row -- fifteen classes by A0 / 64 and ceil(AF / 64), a switch and four
			// straight-line rounds each, round 3's verdict 2b -- measured: a compare tree of four branches instead of eight compares,
			// 400 instructions more, ten more scalar registers spilled: -2 % on configs 1, 3 and 4)
			// (a fourth kind -- every base of the round in the overlap, nine pairs in ten have one: no border selects, eight vector
			// instructions fewer -- measured: -0.8 % on config 1, -1.7 % on config 4; the branch costs more than the selects)
			const int j = 64 * R + lane;
			const bool inF = j < AF, inR = j >= A0;
			const unsigned ab = e_f & e_r;	// low byte: the bases the two reads agree on
			const unsigned fq = e_f >> 8, rq = e_r >> 8;
			const bool match = (ab & 0xFFu) != 0;
			const u64 match_lanes = ballot(match);	// (taken where the comparison is made: past the B-cliff branch below the predicate has become a 0 / 1 register, and its ballot two more vector instructions)
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
				qi = mat_index(match, fq, rq);
			}
			if (cliff) {
				asm volatile("; B-cliff");	// (keeps this a branch: as selects it would cost every pair six instructions a round)
				const bool fmask = j + fo >= uf, rmask = rlen - 1 + A - j >= ur;
				qi = rmask ? PAMD_LT_SCORE + fq : qi;
				qi = fmask ? PAMD_LT_SCORE + (rmask ? 47u : rq) : qi;	// slot 47 of the score block holds qual_nn
			}
			qi = inR ? qi : PAMD_LT_SCORE + fq;	// forward-only
			qi = inF ? qi : PAMD_LT_SCORE + rq;	// reverse-only
			nt = inR ? nt : e_f;
			nt = inF ? nt : e_r;
			qi_out = qi;
			nt_out = ok ? nt : 0u;
#ifndef FK_NO_COUNTS	/* experiment: what do the two counts of the merge cost? (results are wrong) */
			mism += __popcll(ballot(inF) & ballot(inR) & ~match_lanes);
#endif
		};
		// N rounds per trip, straight-line: their table reads are in flight together; trips nest so that nothing but the
		// accumulators lives across them.  Four rounds per trip in the short geometry (a round past the row's end costs three
		// instructions, an LDS round trip a couple of hundred cycles: 13 % on rdp_mle with its modules, nothing either way on
		// simple_bayesian), two in the LONG one, which has no registers to spare.
		auto trip = [&](auto r0c, auto nc) __attribute__((always_inline)) {
			constexpr int R0 = decltype(r0c)::value, N = decltype(nc)::value;
			unsigned qi[4], ntv[4];
			round(std::integral_constant<int, R0>(), qi[0], ntv[0]);
			round(std::integral_constant<int, R0 + 1>(), qi[1], ntv[1]);
			if constexpr (N > 2) {
				round(std::integral_constant<int, R0 + 2>(), qi[2], ntv[2]);
				round(std::integral_constant<int, R0 + 3>(), qi[3], ntv[3]);
			}
			double pv[4];
#pragma unroll
			for (int k = 0; k < N; k++)
				pv[k] = lds_read_f64(lT + 8u * qi[k]);
#pragma unroll
			for (int k = 0; k < N; k++) {
				if (k > 1 && 64 * (R0 + k) >= seqlen)
					break;
				lds_write_u8(nN + 64u * (R0 + k), ntv[k]);	// (measured and not kept, round 6: the base stored to HBM from here, a byte per lane, no staging and no round trip before the row leaves: -0.8 % config 1, -1 % config 4)
				if constexpr (POST) {
					// (the row leaves once the primers are off it: until then it is its table indices, beside the nt bytes)
					*reinterpret_cast<__attribute__((address_space(3))) uint16_t *>(lN + (L::PIDX - L::NT) + 2u * (unsigned) (64 * (R0 + k) + lane)) = (uint16_t) qi[k];
					if (EXACT)
						xsP[64 * (R0 + k)] = (uint16_t) (8u * qi[k]);
					if (lane < seqlen - 64 * (R0 + k))
						qacc += pv[k];
				} else if (lane < lim - 64 * (R0 + k)) {
					// (measured and not kept, round 5: the lanes between the row's end and the end of its last 16-base unit storing 0.0,
					// so that the row's last cache line leaves whole: reads unchanged, 47 more bytes written per pair, -2 % -- the extra
					// select per round costs more than partial lines do)
					// (measured, round 5: the same launch writing a u16 table entry per base instead of the f64 -- a quarter of the row's
					// bytes -- is no faster at 8 M pairs and 8 % SLOWER at 50 M: the kernel is bound by vector issue, not by what it writes)
					store_stream(&gP[64 * (R0 + k)], pv[k]);
					if (EXACT)	// the base's table index for the ordered sums (exact_sums_group): a cached store, read back within FK_XG pairs
						xsP[64 * (R0 + k)] = (uint16_t) (8u * qi[k]);	// (as the entry's byte offset in the image: 8 x 6 723 fits sixteen bits)
					qacc += pv[k];
					if (PHRED) {
						// (a log-p at or below qual_score[0] loses its high word: what is left is a tiny positive number, which
						// no minimum of values <= 0 takes -- one select instead of two; the minimum as the bare instruction:
						// the builtin canonicalises both operands first, two more instructions per base)
						pall = min_f64(pall, pv[k]);
						const u64 bits = (u64) __double_as_longlong(pv[k]);
						const double above = __hiloint2double(pv[k] <= qs0 ? 0 : (int) (bits >> 32), (int) (unsigned) bits);
						pmin = min_f64(pmin, above);
					}
				}
#ifndef FK_NO_COUNTS
				// (measured and not kept, round 6: counted after the merge, from the staged row, and only for pairs one of whose reads holds
				// an N -- nine pairs in ten of config 1 do, two or three bases of 500: -6 %)
				degen += __popcll(ballot((ntv[k] & 0xFFu) == 15u));	// PANDA_NT_IS_DEGN: only N here
#endif
			}
		};
		// (measured and not kept, round 6: the second trip's rounds and table reads issued before the first trip's stores -- no spill,
		// one exposed round trip fewer on rows of more than 256 bases: +0.4 % on config 1, inside the noise)
		typedef std::integral_constant<int, 2> two;
		typedef std::integral_constant<int, 4> four;
		if constexpr (LONG) {
			trip(std::integral_constant<int, 0>(), four());
			if (seqlen > 256) {
				trip(std::integral_constant<int, 4>(), four());
				if (seqlen > 512)
					trip(std::integral_constant<int, 8>(), two());
			}
		} else {
			trip(std::integral_constant<int, 0>(), four());
			if (seqlen > 256) {
				if constexpr (NARROW)
					trip(std::integral_constant<int, 4>(), two());
				else
					trip(std::integral_constant<int, 4>(), four());
			}
		}
		wave_sync();
		STOP(4);
		DEBUG_DUMP(P, wb, pair, lane, 1, A, novl, Bn, mism, degen, seqlen);
		if constexpr (POST) {
			MARK("post-primers");
			// ---- `-a` (assembler.c:300-333).  `quality` is the whole merge's (assembler.c:244 precedes the clipping): finished
			// here, the LOWQ verdict guarded as everywhere
			const double wsum = wave_sum_any_order(qacc), flen1 = (double) (seqlen + 1);
			if (fabs(wsum - P.threshold * flen1) <= 1e-9 * flen1) {
				note(P, 4, lane);
				status = PAMD_ST_DEFER;
				break;
			}
			qsum = wsum / flen1;
			post_streams(wb, seqlen, lane);
			wave_sync();
			int fpl = P.fprimer_len, rpl = P.rprimer_len;
			asm volatile("" : "+s"(fpl), "+s"(rpl));
			int pfo = P.ftrim, pro = P.rtrim;
			bool unsettled = false;
			if (fpl > 0) {	// the forward primer from the sequence's first base on (offset.c:124-133)
				int r = 0;
				if (fpl <= seqlen) {	// (offset.c:55-57; alignments 0 .. seqlen - fpl - 1 are judged, offset.c:70-77)
					int s1;
					const int c = post_candidates(wb, seqlen - fpl, P.primer_codes[0].code[0], P.primer_codes[0].counts[0], false, lane, s1);
					r = c == 0 ? 0 : (c == 1 ? post_verdict(ltab, wb, s1, false, seqlen, fpl, primer_bases & 0xFFu, (double) fpl * P.threshold, lane) : -1);
				}
				if (r < 0) {
					unsettled = true;
				} else if (r == 0) {
					status = PAMD_ST_NOFP;
					mism = degen = 0;	// (the record of such a pair carries neither, as the general kernel's)
					break;
				} else {
					pfo = r - 1;
				}
			}
			fo = pfo;
			if (rpl > 0 && !unsettled) {	// the reverse primer from its last base backwards: alignment s from the end is alignment seqlen - rpl - s of the primer read backwards
				int r = 0;
				if (rpl <= seqlen) {
					int t1;
					const int c = post_candidates(wb, seqlen - rpl + 1, P.primer_codes[2].code[0], P.primer_codes[2].counts[0], true, lane, t1);
					r = c == 0 ? 0 : (c == 1 ? post_verdict(ltab, wb, seqlen - rpl - t1, true, seqlen, rpl, (primer_bases >> 8) & 0xFFu, (double) rpl * P.threshold, lane) : -1);
				}
				if (r < 0) {
					unsettled = true;
				} else if (r == 0) {
					status = PAMD_ST_NORP;
					mism = degen = 0;
					break;
				} else {
					pro = r - 1;
				}
			}
			if (unsettled) {	// two alignments without a disagreement, or a verdict too close to the threshold: the general kernel's pair
				note(P, 6, lane);
				status = PAMD_ST_DEFER;
				fo = 0;
				break;
			}
			ro = pro;
			if (seqlen <= pfo + pro) {	// assembler.c:323-327: counted as a missing forward primer
				status = PAMD_ST_NOFP;
				mism = degen = 0;
				break;
			}
			const int out_len = seqlen - pfo - pro;
			MARK("post-row-out");
			// the kept bases leave: log-p from their table indices, 64 per round; nt sixteen per lane from wherever the forward
			// primer ended (whole words of the staged row, shifted)
			{
				const unsigned lP = lN + (L::PIDX - L::NT) + 2u * (unsigned) (pfo + lane);
				const int rounds = (out_len + 63) >> 6;
				for (int r0 = 0; r0 < rounds; r0 += 4) {
					unsigned ix[4];
					double pv[4];
#pragma unroll
					for (int k = 0; k < 4; k++)
						ix[k] = lds_read_u16(lP + 128u * (unsigned) (r0 + k)) & 0x1FFFu;	// (a round past the row's end reads what lies there: kept inside the LDS)
#pragma unroll
					for (int k = 0; k < 4; k++)
						pv[k] = lds_read_f64(lT + 8u * ix[k]);
#pragma unroll
					for (int k = 0; k < 4; k++) {
						if (lane < out_len - 64 * (r0 + k)) {
							if (store)
								store_stream(&gP[64 * (r0 + k)], pv[k]);
							if (PHRED) {
								pall = min_f64(pall, pv[k]);
								const u64 bits = (u64) __double_as_longlong(pv[k]);
								const double above = __hiloint2double(pv[k] <= qs0 ? 0 : (int) (bits >> 32), (int) (unsigned) bits);
								pmin = min_f64(pmin, above);
							}
						}
					}
				}
				if (store && lane * 16 < out_len) {
					const unsigned a = lN + (unsigned) pfo + 16u * (unsigned) lane, base = a & ~3u, sh = 8u * (a & 3u);
					const unsigned w0 = lds_read_u32(base), w1 = lds_read_u32(base + 4u), w2 = lds_read_u32(base + 8u), w3 = lds_read_u32(base + 12u), w4 = lds_read_u32(base + 16u);
					u32x4 o;
					o.x = __builtin_amdgcn_alignbit(w1, w0, sh);
					o.y = __builtin_amdgcn_alignbit(w2, w1, sh);
					o.z = __builtin_amdgcn_alignbit(w3, w2, sh);
					o.w = __builtin_amdgcn_alignbit(w4, w3, sh);
					store_stream(&((gbl_u128 *) row_nt)[lane], o);
				}
			}
			if (PHRED) {	// the smallest phred of the KEPT bases (plugin_min_phred.c:8-22 sees the clipped sequence)
				phmin = fk_phred_of(ltab + PAMD_LT_SCORE, places, wave_min_nonpositive(pmin), lane);
				if (ballot(pall <= qs0) != 0 && phmin > 1)
					phmin = 1;
				if (out_len <= 0)
					phmin = 127;
			}
			seqlen = out_len;	// what the record reports (assembler.c:328)
			break;
		}
		MARK("row-out+sum");
		// the nt row leaves LDS in 16-byte stores (rows are 16-base aligned)
		if (store && lane * 16 < seqlen)
			store_stream(&((gbl_u128 *) row_nt)[lane], reinterpret_cast<const u32x4 *>(nts)[lane]);
		if constexpr (LAZYP) {
			*plane = pmin;	// this lane's smallest log-p above qual_score[0]; the row's, and its phred, are finish64's
			phflag = ballot(pall <= qs0) != 0;
		} else if (PHRED) {
			// one lookup per pair instead of one walk per base (see above)
			phmin = fk_phred_of(ltab + PAMD_LT_SCORE, places, wave_min_nonpositive(pmin), lane);
			if (ballot(pall <= qs0) != 0 && phmin > 1)
				phmin = 1;
			if (seqlen <= 0)
				phmin = 127;	// an empty row passes any threshold
		}
		if constexpr (LAZYQ) {
			*qlane = qacc;	// this lane's share; the wave's sum comes out of the group's merge tree, the LOWQ guard is finish64's
		} else if (!EXACT) {
			qsum = wave_sum_any_order(qacc);	// divided by len in finish64() ...
			// ... unless the quotient could land within 1e-9 of the LOWQ threshold (tested without dividing): the general
			// kernel then makes the three sums in reference order (assembler.c:171,226,236,244,334)
#if FK_KARGS
			const double k_thr = H.threshold;
#else
			const double k_thr = P.threshold;
#endif
			if (__builtin_expect(fabs(qsum - k_thr * (double) (seqlen + 1)) <= 1e-9 * (double) (seqlen + 1), 0)) {
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
	if constexpr (PRIM) {
		// a deferred pair takes its primer offsets along (what the scan returned, forward | reverse << 16), or the mark that
		// the primer kernel has to find them
		if (status == PAMD_ST_DEFER && P.primer_offsets != nullptr && lane == 0)
			P.primer_offsets[pair] = primers_known ? ((P.fprimer_len > 0 ? (unsigned) fo + 1u : 0u) | ((P.rprimer_len > 0 ? (unsigned) ro + 1u : 0u) << 16)) : 0xFFFFFFFFu;
	}
	// park the pair's scalars in lane `slot_idx` of the pending registers (finish64): every value is wave-uniform and the lane
	// is a scalar, so each field is one v_writelane (a compare-and-select per field costs a move from the scalar file as well)
	union {
		double d;
		unsigned u[2];
	} qq, bb;
	qq.d = qsum;
	bb.d = best;
	// (v_writelane_b32 takes one scalar register besides M0: the lane goes into M0, once per group of fields.  M0 cannot be
	// declared clobbered -- it is a reserved register --; nothing the compiler emits for these kernels reads it: no LDS-DMA, no
	// s_movrel, no GDS; `make asm` + grep m0 is the check after a compiler change)
	const unsigned f_len_bo = (unsigned) seqlen | ((unsigned) (bo < 0 ? 0 : bo) << 16), f_ex_mm = (unsigned) examined | ((unsigned) mism << 16);
	const unsigned f_dg_fo = (unsigned) degen | ((unsigned) fo << 16);
	// (POST: bit 25, "the quality is final": finish64 does not divide it by the clipped length)
	const unsigned f_ro_st = (unsigned) ro | ((unsigned) status << 16) | ((unsigned) slow << 24) | (POST ? 1u << 25 : 0u) | (LAZYP ? (phflag ? 1u << 26 : 0u) : (unsigned) (phmin > 63 ? 63 : phmin) << 26);
	if constexpr (LAZYQ) {	// (the quality sums come out of the merge tree, in lane bitrev6(slot): the other scalars go there too)
		slot_idx = (int) (__builtin_bitreverse32((unsigned) slot_idx) >> 26);
		asm volatile("s_mov_b32 m0, %4\n\t"
			     "v_writelane_b32 %0, %5, m0\n\t"
			     "v_writelane_b32 %1, %6, m0\n\t"
			     "v_writelane_b32 %2, %7, m0\n\t"
			     "v_writelane_b32 %3, %8, m0"
			     : "+v"(pd.off_lo), "+v"(pd.off_hi), "+v"(pd.b_lo), "+v"(pd.b_hi)
			     : "s"(slot_idx), "s"(uni((int) (unsigned) off)), "s"(uni((int) (unsigned) (off >> 32))), "s"(uni((int) bb.u[0])), "s"(uni((int) bb.u[1])));
	} else
	asm volatile("s_mov_b32 m0, %6\n\t"
		     "v_writelane_b32 %0, %7, m0\n\t"
		     "v_writelane_b32 %1, %8, m0\n\t"
		     "v_writelane_b32 %2, %9, m0\n\t"
		     "v_writelane_b32 %3, %10, m0\n\t"
		     "v_writelane_b32 %4, %11, m0\n\t"
		     "v_writelane_b32 %5, %12, m0"
		     : "+v"(pd.off_lo), "+v"(pd.off_hi), "+v"(pd.q_lo), "+v"(pd.q_hi), "+v"(pd.b_lo), "+v"(pd.b_hi)
		     : "s"(slot_idx), "s"(uni((int) (unsigned) off)), "s"(uni((int) (unsigned) (off >> 32))), "s"(uni((int) qq.u[0])), "s"(uni((int) qq.u[1])), "s"(uni((int) bb.u[0])), "s"(uni((int) bb.u[1])));
	asm volatile("s_mov_b32 m0, %4\n\t"
		     "v_writelane_b32 %0, %5, m0\n\t"
		     "v_writelane_b32 %1, %6, m0\n\t"
		     "v_writelane_b32 %2, %7, m0\n\t"
		     "v_writelane_b32 %3, %8, m0"
		     : "+v"(pd.len_bo), "+v"(pd.ex_mm), "+v"(pd.dg_fo), "+v"(pd.ro_st)
		     : "s"(slot_idx), "s"(uni((int) f_len_bo)), "s"(uni((int) f_ex_mm)), "s"(uni((int) f_dg_fo)), "s"(uni((int) f_ro_st)));
	if (EXACT) {	// exact sums: where the row's three chains begin and end (assembler.c:171,226,236): the overlap is [A0, AF)
		const unsigned f_pid = (unsigned) region_a0 | ((unsigned) region_af << 10) | ((unsigned) region_len << 20);	// (each at most 640)
		asm volatile("s_mov_b32 m0, %1\n\tv_writelane_b32 %0, %2, m0" : "+v"(pd.pid) : "s"(slot_idx), "s"(uni((int) f_pid)));
	}
}

#ifndef FK_NO_DUAL	/* (tests that only look at the single-pair kernels' ISA compile without the dual geometry) */
#include "pamd_fast_dual.h"
#endif

// PHRED: a min_phred module is loaded, the merge tracks the row's smallest log-p (plugin_min_phred.c:8-22)
// (one 16-wave workgroup fills a CU's LDS: four waves per SIMD whatever the registers allow, so the compiler is told to
// spend all 128 vector registers on scheduling instead of saving them for waves that cannot come -- 1 % on a 16 M-pair
// launch, nothing measurable at 50 M.)
template <int ALGO, bool EXACT, bool PHRED, bool LONG, bool PRIM, bool NARROW, bool POST = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pamd_fast_kernel(const PamdKernelParams P) {
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
	unsigned char *wb = lds + P.lds_primer_off + (size_t) wave * Geo<LONG, t1_bits<ALGO, LONG>()>::WAVE_BYTES;	// (the per-wave regions start where the general kernel keeps its primers)
	unsigned char *places = lds + P.lds_primer_off + (size_t) nwaves * Geo<LONG, t1_bits<ALGO, LONG>()>::WAVE_BYTES;	// PHRED: 96 bytes behind the last wave's region (fk_phred_places)
	// EXACT: this wave's scratch rows for the table indices of FK_XG pairs (exact_sums_group)
	uint16_t *const xs = EXACT ? reinterpret_cast<uint16_t *>(P.score_scratch) + ((size_t) blockIdx.x * (size_t) nwaves + (size_t) wave) * (size_t) (fk_xg<EXACT && !counting_algo<ALGO>()>() * fk_xstride<LONG, EXACT && !counting_algo<ALGO>()>()) : nullptr;
	unsigned primer_bases = 0;	// PRIM: this lane's base of the forward primer | of the reverse primer << 8 (N past their ends) | << 16: lanes 0..31 the forward, lanes 32..63 the reverse primer's bases 0..31
	// what lies before a read reads as N: the word before the forward N map (seed windows start eight bases early)
	if (lane < 4)
		reinterpret_cast<unsigned *>(wb + Lay<LONG>::PAD0)[lane] = 0x55555555u;
	if constexpr (LONG) {
		if (lane < 4)
			reinterpret_cast<unsigned *>(wb + Lay<true>::PAD1)[lane] = 0xAAAAAAAAu;
		if constexpr (PRIM || POST)
			primer_bases = (lane < P.fprimer_len ? (unsigned) P.primers[lane] : 15u) | ((lane < P.rprimer_len ? (unsigned) P.primers[456 + lane] : 15u) << 8)
				       | ((lane < 32 ? (lane < P.fprimer_len ? (unsigned) P.primers[lane] : 15u) : (lane - 32 < P.rprimer_len ? (unsigned) P.primers[456 + lane - 32] : 15u)) << 16);
	}
	__syncthreads();
	if (EXACT && ALGO == PAMD_ALGO_PEAR) {	// the slot behind the two matrices: what an N adds to pear's score (algo_pear.c:47-49), for the score's chain
		if (tid == 0)
			ltab[PAMD_LT_C] = -P.pear_random_base;
		__syncthreads();
	}
	if (PHRED) {	// (after the tables are staged: the places are made from the staged score table)
		fk_phred_places(ltab + PAMD_LT_SCORE, places, tid);
		__syncthreads();
	}

	// (measured, round 4, and left out: the four waves of a SIMD at four static priorities (s_setprio by (wave >> 2) & 3): +-0.3 %;
	// the four started a quarter or half of a pair's time apart (s_sleep): +-0.3 % -- they drift apart by themselves;
	// the compiler's max-ilp and max-memory-clause scheduling strategies: +0.6 % / -0.3 % on config 1, inside the noise on config 4)
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
	// Which pairs a wave takes.  Strided (pair w, w + W, w + 2W, .. for wave w of W): even to the last pair, but a read's row is
	// not a whole number of cache lines and its neighbours belong to other waves, which drift apart by far more than a line lives
	// in the L2 -- every row fetches its first and last line whole, and every 2-byte length a sector (config 1: 1 230 B read per
	// pair for 1 000, config 4: 983 for 600).  In blocks (P.pair_blocks, large launches): F consecutive pairs, F = the pairs of one
	// finish, then the block W * F further on -- consecutive rows come one pair's time apart, and a finish writes F consecutive records.
	constexpr unsigned F = EXACT ? (unsigned) ((EXACT && !counting_algo<ALGO>() ? 4 : 3) * fk_xg<EXACT && !counting_algo<ALGO>()>()) : 64u;
	const unsigned step_in = P.pair_blocks ? 1u : total_waves;
	unsigned pair = (blockIdx.x * (unsigned) nwaves + (unsigned) wave) * (P.pair_blocks ? F : 1u);
	const unsigned last = n_pairs - 1u;
	unsigned q = pair < n_pairs ? pair : last;
	Carry<LONG> cur;
	load_pair<LONG, EXACT>(cur, P, q, lane);
	settle(cur);
	int npend = 0, xg = 0;	// pairs parked for finish64; of those, pairs of the current group of ordered sums (EXACT)
	constexpr bool XSCORE = EXACT && !counting_algo<ALGO>();
	constexpr int XG = fk_xg<XSCORE>();
	bool deep_dirty = true;	// the first pair clears all four tables
	bool flags_dirty = true;	// ... and the candidate flags
	unsigned epoch = 0;	// FK_EPOCH: the epoch of the words the last pair wrote (0: the next pair clears the tables and starts over at 126)
	std::conditional_t<LONG, LenMasks8, LenMasks> lmf = { -1 }, lmr = { -1 };	// FK_LEN_MASKS
	constexpr bool LQ = FK_LAZYQ != 0 && !EXACT && !POST && !LONG;	// (the LONG geometry has no registers for the tree's six levels)
	static_assert(!LQ || F == 64u, "the merge tree closes after 64 pairs");
	constexpr bool LP = FK_LAZYP != 0 && LQ && PHRED;
	QTree qt, pt;
	HotArgs H = { P.qual_nn, P.sb_pmatch, P.sb_pmismatch, P.threshold, P.seq_stride, P.seq_capacity, P.seq_nt, P.seq_p };
#if FK_KARGS
	asm volatile("" : "+s"(H.qual_nn), "+s"(H.pmatch), "+s"(H.pmismatch), "+s"(H.threshold), "+s"(H.stride), "+s"(H.capacity), "+s"(H.nt), "+s"(H.p));
#endif
	Pend64 pd = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	auto take_group = [&](int base, int count) __attribute__((always_inline)) {
		const GroupSums g = exact_sums_group<LONG, XSCORE, ALGO == PAMD_ALGO_RDP_MLE>(lds_addr(ltab), xs, pd.len_bo, pd.pid, pd.ro_st, pd.off_lo, pd.off_hi, P.qual_nn, base, count, lane);
		pd.q_lo = g.done ? g.q_lo : pd.q_lo;
		pd.q_hi = g.done ? g.q_hi : pd.q_hi;
		if (XSCORE) {
			pd.b_lo = g.done ? g.b_lo : pd.b_lo;
			pd.b_hi = g.done ? g.b_hi : pd.b_hi;
		}
		pd.ro_st |= g.done ? 1u << 25 : 0u;	// "the quality is final"
	};
	unsigned group_first = pair;
	while (pair < n_pairs) {
		const unsigned next = npend + 1 == (int) F ? group_first + total_waves * F : pair + step_in;	// (strided: the same pair either way)
		q = next < n_pairs ? next : last;
		Carry<LONG> nxt;
		load_pair<LONG, EXACT>(nxt, P, q, lane);
		const int flen = uni(cur.flen_v), rlen = uni(cur.rlen_v);
		int lane_now = lane;
		if constexpr (LONG || ALGO == PAMD_ALGO_EA_UTIL || ALGO == PAMD_ALGO_STITCH || !counting_algo<ALGO>()) {	// (the two table-scored counting scorers spill in the short geometry too; so do pear and rdp_mle, whose many-candidate scan is a call the hot path's registers have to survive)
			// The compiler hoists everything that depends on the lane number alone out of this loop -- multiples, shifts,
			// LDS addresses -- and, out of registers, spills it: a reload from scratch memory then waits, vmcnt being one
			// in-order queue, for whatever went before it, the rows just requested for the next pair among them.  Made opaque
			// here, the lane number's derivatives are recomputed where they are used (an instruction or two each).
			asm volatile("" : "+v"(lane_now));
		}
		double qlane, plane;
		fast_pair<ALGO, EXACT, PHRED, LONG, PRIM, NARROW, POST, LQ>(P, ltab, places, wb, arena, pd, npend, deep_dirty, pair, lane_now, flen, rlen, cur, nxt, primer_bases, EXACT ? xs + (unsigned) xg * fk_xstride<LONG, XSCORE>() : nullptr, H, flags_dirty, epoch, lmf, lmr, &qlane, &plane);
		MARK("loop");
		if constexpr (LQ) {
			F64Pair sums;
			if (qsum_insert(qt, npend, qlane, lane, sums.d)) {
				pd.q_lo = sums.u[0];
				pd.q_hi = sums.u[1];
			}
			if constexpr (LP) {
				if (qsum_insert<true>(pt, npend, plane, lane, sums.d)) {
					pd.p_lo = sums.u[0];
					pd.p_hi = sums.u[1];
				}
			}
		}
		npend++;
		if (EXACT && ++xg == XG) {	// sums in the reference's order: a group of pairs is complete
			take_group(npend - XG, XG);
			xg = 0;
		}
		if (npend == (int) F) {
			finish64<false, true, false, false, LQ, LP>(P, lctr, pd, npend, group_first, step_in, lane, ltab + PAMD_LT_SCORE);
			npend = 0;
			group_first = next;
		}
		pair = next;
	}
	if (EXACT && xg != 0)
		take_group(npend - xg, xg);
	if (npend > 0) {
		if constexpr (LQ) {	// the last group is not full: the tree is closed with pairs that hand in nothing
			F64Pair sums;
			sums.d = 0.0;
			for (int s = npend; s < 64; s++)
				qsum_insert(qt, s, 0.0, lane, sums.d);
			pd.q_lo = sums.u[0];
			pd.q_hi = sums.u[1];
			if constexpr (LP) {
				sums.d = 0.0;
				for (int s = npend; s < 64; s++)
					qsum_insert<true>(pt, s, 0.0, lane, sums.d);
				pd.p_lo = sums.u[0];
				pd.p_hi = sums.u[1];
			}
		}
		finish64<false, true, false, false, LQ, LP>(P, lctr, pd, npend, group_first, step_in, lane, ltab + PAMD_LT_SCORE);
	}
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
// PAMD_FAST_POST: `-a`, reads of at most 320 bases (the LONG geometry), either summation mode.
// 0: not served (the general kernel takes the launch whole); PAMD_FAST_SHORT: reads of at most 256 bases, no primers or
// trims; PAMD_FAST_LONG: reads of at most 320 bases; PAMD_FAST_PRIM: the same with `-p` / `-q` primers (at most 24 bases, no
// penalty) and / or trims.
int pamd_fast_supports(const PamdKernelParams &p) {
	if (p.maxoverlap != 0 && (p.maxoverlap < p.minoverlap || p.maxoverlap > 0x7FFF))
		return 0;	// (a maximum the reads can reach is served pair by pair: fast_pair)
	if (p.n_pairs > 0x7FFFFFFFull)
		return 0;	// 32-bit pair indices in the kernel and in the deferred list
	if (p.post_primers) {
		// `-a` (assembler.c:300-333): the lean scan sums only alignments without a disagreeing base, which is exact when a
		// disagreement makes an alignment's value NaN or -inf (notp_never_finite); no penalty; primers of at most 24 bases
		if (!p.notp_never_finite || p.primer_penalty != 0.0 || p.smax > (uint32_t) Lay<true>::MAXLEN)
			return 0;
		if (p.fprimer_len > 24 || p.rprimer_len > 24 || p.ftrim < 0 || p.rtrim < 0 || p.ftrim > 2 * Lay<true>::MAXLEN || p.rtrim > 2 * Lay<true>::MAXLEN)
			return 0;
		if ((double) p.fprimer_len * p.threshold < -700.0 || (double) p.rprimer_len * p.threshold < -700.0)
			return 0;
		return PAMD_FAST_POST;
	}
	const bool clipped = p.fprimer_len > 0 || p.rprimer_len > 0 || p.ftrim != 0 || p.rtrim != 0;
	if (clipped) {
		if (p.smax > (uint32_t) Lay<true>::MAXLEN || p.primer_penalty != 0.0 || p.fprimer_len > 24 || p.rprimer_len > 24)
			return 0;
		if (p.ftrim < 0 || p.rtrim < 0 || p.ftrim > Lay<true>::MAXLEN || p.rtrim > Lay<true>::MAXLEN)
			return 0;
		// The lean scan compares alignments before exp() (offset.c:59,72-74 compare after it).  With plen * threshold near
		// exp()'s underflow both sides of the reference's comparison are 0 or denormal and nothing is found; the careful primer
		// kernel, which compares after exp(), takes such a launch.
		if ((double) p.fprimer_len * p.threshold < -700.0 || (double) p.rprimer_len * p.threshold < -700.0)
			return 0;
		return PAMD_FAST_PRIM;
	}
	if (p.smax <= (uint32_t) Lay<false>::MAXLEN)
		return PAMD_FAST_SHORT;
	if (p.smax <= (uint32_t) Lay<true>::MAXLEN)
		return PAMD_FAST_LONG;
	return 0;
}

// Two pairs per wave (pamd_fast_dual.h): reads of at most 160 bases, a counting scorer, any-order sums, no min_phred module, no
// configured maximum overlap (and the short geometry to begin with: no primers, no trims)
bool pamd_fast_dual(const PamdKernelParams &p) {
#ifdef FK_NO_DUAL
	return false;
#else
	if (!p.two_pairs || pamd_fast_supports(p) != PAMD_FAST_SHORT || p.smax > (uint32_t) dual::D_MAXLEN)
		return false;
	if (p.algo == PAMD_ALGO_PEAR || p.algo == PAMD_ALGO_RDP_MLE || p.exact_sums || p.has_min_phred || p.maxoverlap != 0)
		return false;
	return true;
#endif
}

// The launch geometry of the lean kernel for this configuration on a device of `cu_count` CUs: workgroups, waves per workgroup,
// LDS bytes per workgroup -- and the LDS offsets of `p` set to what its kernel expects.
void pamd_fast_geometry(PamdKernelParams &p, int cu_count, size_t lds_limit, int max_waves, int *workgroups, int *waves_per_wg, size_t *lds_bytes) {
	const bool rdp = p.algo == PAMD_ALGO_RDP_MLE;
	const int cus = cu_count > 0 ? cu_count : 256;
	if (p.algo == PAMD_ALGO_PEAR && p.exact_sums && pamd_fast_supports(p) != 0) {	// (the slot behind the matrices: -random_base, for the score's chain)
		p.lds_table_doubles = PAMD_LT_C + 1;
		p.lds_ctr_off = (p.lds_table_doubles * 8u + 15u) & ~15u;
		p.lds_primer_off = (p.lds_ctr_off + PAMD_NUM_COUNTERS * 4u + 15u) & ~15u;
	}
	const size_t wave = pamd_fast_supports(p) == PAMD_FAST_SHORT ? (rdp ? Geo<false, 9>::WAVE_BYTES : Geo<false, 10>::WAVE_BYTES) : (rdp ? Geo<true, 9>::WAVE_BYTES : Geo<true, 10>::WAVE_BYTES);
	int waves = max_waves;
	while (waves > 1 && (size_t) p.lds_primer_off + (size_t) waves * wave + 96u > lds_limit)	// (+ the 96 bytes of fk_phred_places)
		waves--;
	if (waves > max_waves)
		waves = max_waves;
	size_t wgs = (size_t) cus;
	const size_t want = ((size_t) p.n_pairs + (size_t) waves - 1) / (size_t) waves;
	if (wgs > want)
		wgs = want;
	*workgroups = (int) wgs;
	*waves_per_wg = waves;
	*lds_bytes = (size_t) p.lds_primer_off + (size_t) waves * wave + 96u;
}

// bytes of PamdKernelParams::score_scratch the lean kernel wants for this launch (sums in the reference's order: FK_XG rows of
// table indices per wave, exact_sums_group); 0 when it wants none
size_t pamd_fast_scratch_bytes(const PamdKernelParams &p, int workgroups, int waves_per_wg) {
	if (!p.exact_sums || pamd_fast_supports(p) == 0)
		return 0;
	return (size_t) workgroups * (size_t) waves_per_wg * (size_t) FK_XG_MAX * (size_t) fk_xstride<true, true>() * sizeof(uint16_t);	// (an upper bound over the instantiations)
}

#ifndef FK_BLOCKS_MIN
#define FK_BLOCKS_MIN 32ull
#endif
int pamd_launch_fast(const PamdKernelParams &p, int workgroups, int waves_per_wg, size_t lds_bytes, void *stream) {
	const dim3 grid((unsigned) workgroups), block((unsigned) waves_per_wg * 64u);
	hipStream_t st = static_cast<hipStream_t>(stream);
	void (*kern)(const PamdKernelParams) = nullptr;
	const int variant = pamd_fast_supports(p);
#ifdef FK_NO_NARROW	/* measurement: reads of at most 192 bases in the geometry of four bases per lane */
	const bool narrow = false;
#else
	const bool narrow = p.smax <= 192;	// three bases per lane cover the reads
#endif
#define FK_PICK3(A, PH, E) (narrow ? pamd_fast_kernel<A, E, PH, false, false, true> : pamd_fast_kernel<A, E, PH, false, false, false>)
#define FK_PICK4(A, PH, E) (variant == PAMD_FAST_LONG ? pamd_fast_kernel<A, E, PH, true, false, false> : pamd_fast_kernel<A, E, PH, true, true, false>)
#define FK_PICK2(A, PH) (variant == PAMD_FAST_POST ? (p.exact_sums ? pamd_fast_kernel<A, true, PH, true, false, false, true> : pamd_fast_kernel<A, false, PH, true, false, false, true>) \
						   : variant == PAMD_FAST_SHORT ? (p.exact_sums ? FK_PICK3(A, PH, true) : FK_PICK3(A, PH, false)) \
										: (p.exact_sums ? FK_PICK4(A, PH, true) : FK_PICK4(A, PH, false)))
#define FK_PICK(A) (p.has_min_phred ? FK_PICK2(A, true) : FK_PICK2(A, false))
#ifndef FK_NO_DUAL
	if (pamd_fast_dual(p)) {
		switch (p.algo) {
		case PAMD_ALGO_SIMPLE_BAYES:
			kern = pamd_fast2_kernel<PAMD_ALGO_SIMPLE_BAYES>;
			break;
#ifndef FK_DEV_ONE
		case PAMD_ALGO_UPARSE:
			kern = pamd_fast2_kernel<PAMD_ALGO_UPARSE>;
			break;
		case PAMD_ALGO_FLASH:
			kern = pamd_fast2_kernel<PAMD_ALGO_FLASH>;
			break;
		case PAMD_ALGO_EA_UTIL:
			kern = pamd_fast2_kernel<PAMD_ALGO_EA_UTIL>;
			break;
		case PAMD_ALGO_STITCH:
			kern = pamd_fast2_kernel<PAMD_ALGO_STITCH>;
			break;
#endif
		default:
			break;
		}
	}
#endif
	if (kern == nullptr) {
#ifdef FK_DEV_ONE	/* development: one instantiation family, for quick compiles */
#ifndef FK_DEV_EXACT
#define FK_DEV_EXACT false
#endif
#ifdef FK_DEV_PRIM	/* ... config 2's: pear, the primers' geometry */
	kern = pamd_fast_kernel<PAMD_ALGO_PEAR, FK_DEV_EXACT, false, true, true, false>;
#elif defined(FK_DEV_PHRED)	/* ... config 3's: rdp_mle with a min_phred module */
	kern = pamd_fast_kernel<PAMD_ALGO_RDP_MLE, FK_DEV_EXACT, true, false, false, false>;
#else
	kern = narrow ? pamd_fast_kernel<PAMD_ALGO_SIMPLE_BAYES, FK_DEV_EXACT, false, false, false, true> : pamd_fast_kernel<PAMD_ALGO_SIMPLE_BAYES, FK_DEV_EXACT, false, false, false, false>;
#endif
#else
	switch (variant == 0 ? -1 : p.algo) {
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
		pamd_set_error("the lean kernel does not serve algorithm %d in this configuration", p.algo);
		return PAMD_EINVAL;
	}
#endif	// FK_DEV_ONE
	}
#undef FK_PICK
#undef FK_PICK2
#undef FK_PICK3
#undef FK_PICK4
	if (pamd_fast_scratch_bytes(p, workgroups, waves_per_wg) > 0 && p.score_scratch == nullptr) {
		pamd_set_error("the lean kernel's ordered sums need their scratch rows");
		return PAMD_EINVAL;
	}
	// pairs in blocks (see the kernel): when every wave has FK_BLOCKS_MIN blocks or more, so that the last, uneven round of blocks
	// is a small share of the launch
	PamdKernelParams pk = p;
	{
		const unsigned long long per_finish = !p.exact_sums ? 64ull : (p.algo == PAMD_ALGO_PEAR || p.algo == PAMD_ALGO_RDP_MLE ? 60ull : 63ull);
		pk.pair_blocks = p.n_pairs >= (unsigned long long) workgroups * (unsigned long long) waves_per_wg * per_finish * FK_BLOCKS_MIN ? 1 : 0;
		if (const char *e = getenv("PAMD_FK_BLOCKS"))	// measurements only
			pk.pair_blocks = atoi(e) != 0;
	}
	hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
	if (e0 != hipSuccess) {
		pamd_set_error("hipFuncSetAttribute(%zu bytes LDS, lean kernel): %s", lds_bytes, hipGetErrorString(e0));
		return PAMD_ENODEVICE;
	}
	hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, pk);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("lean kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
