// pandaseq_amd -- the primer kernel: where the forward and the reverse primer end in every read pair.
//
// `-p` / `-q` (assembler.c:262-288 calling panda_compute_offset_qual, offset.c:47-112): before a pair is aligned, its
// reads are scanned from their 5' ends for the primer; what the scan returns is how much of the read is cut off, or that
// the pair has no primer (NOFP / NORP).  The scan needs a read, one primer and two 47-entry tables -- none of what the
// assembly kernel carries -- so it runs as a kernel of its own over the whole batch before the assembly kernel, at twice
// the assembly kernel's occupancy and without its register pressure, and leaves one word per pair
// (PamdKernelParams::primer_offsets: forward result | reverse result << 16, each what computeoffset() returns) for the
// assembly kernel to pick up.  It costs a second read of the input rows (HBM is not what bounds either kernel).
//
// One wave per pair: both rows are loaded with one 16-byte load per lane, staged in LDS as the u16-per-base rows they are,
// and the rows of the wave's next pair are in flight while this one is scanned.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "pamd_internal.h"
#include "pamd_device.h"

namespace {

using namespace pamd_dev;

// what the kernel needs of PamdKernelParams (a short argument block keeps it in scalar registers without spilling)
// -DPK_DEBUG: event counts per launch, printed by the launcher (development aid; never in a shipped build)
#ifdef PK_DEBUG
__device__ unsigned long long pk_events[16];
#define PK_COUNT(i) do { if (lane == 0) atomicAdd(&pk_events[i], 1ull); } while (0)
#else
#define PK_COUNT(i)
#endif

struct PrimerParams {
	const uint8_t *fwd;
	const uint8_t *rev;
	const uint16_t *flen;
	const uint16_t *rlen;
	const double *tables;
	const uint8_t *primers;
	uint32_t *primer_offsets;
	const uint32_t *in_list;	// listed launches: the pairs to scan ...
	const unsigned *in_count;	// ... and how many (device word)
	uint32_t *redo_list;	// the pairs the fast kernel leaves to the careful one ...
	unsigned *redo_count;	// ... and how many (device word)
	unsigned *cursor;	// work counter (device word, zeroed per kernel): the batch goes out sixteen pairs at a time
	uint64_t n_pairs;
	uint32_t fwd_row_bytes;
	uint32_t rev_row_bytes;
	int32_t fprimer_len;
	int32_t rprimer_len;
	double threshold;
	double primer_penalty;
};

constexpr unsigned NT_N = 15;

__device__ __forceinline__ int nt_of(unsigned v) {
	return (int) (v & 0xFFu);
}

__device__ __forceinline__ int qual_of(unsigned v) {
	return (int) (int8_t) (v >> 8);
}

// One alignment by the whole wave: lane x looks up the term of primer base x, and the terms are added in primer order --
// the same additions in the same order as a lane of the scan below makes for its own alignment (a primer N adds 0.0, which
// changes nothing: the sum is never -0.0).  plen <= 64.
__device__ __forceinline__ double primer_alignment_sum(const double *ltab, const uint16_t *S, int s, const uint8_t *primer, int plen, int lane) {
	double t = 0.0;
	if (lane < plen) {
		const unsigned pn = primer[lane];
		const unsigned e = S[s + lane];
		const int q = clampq(qual_of(e));
		const double v = ltab[(((nt_of(e) & pn) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + q];
		t = pn != NT_N ? v : 0.0;
	}
	// lanes past the primer hold 0.0 as well: eight terms at a time, lane numbers as constants
	double acc = 0.0;
#pragma unroll
	for (int c = 0; c < 8; c++) {
		if (c * 8 >= plen)
			break;
#pragma unroll
		for (int u = 0; u < 8; u++)
			acc += lane_f64(t, c * 8 + u);
	}
	return acc;
}

// Lanes as read positions (nt, and whether the position's term is below the part of the floor when it agrees -- low_s --
// and when it does not -- low_e): the alignments with at least K of their first eight terms below that part, bit s for the
// alignment that starts at the block's position s.  HASN: one of the primer's first eight bases is N (no term).
template <int K, bool HASN>
__device__ __forceinline__ u64 low_terms(const unsigned nt, const unsigned (&pn8)[8], const u64 low_s, const u64 low_e, const u64 valid) {
	const u64 either = low_s ^ low_e;
	u64 ge1 = 0, ge2 = 0, ge3 = 0;
#pragma unroll
	for (int x = 0; x < 8; x++) {
		if (x == 4 && (valid & ~(K == 1 ? ge1 : (K == 2 ? ge2 : ge3))) == 0)
			break;	// four bases were enough: nothing of the block is left (four blocks in five)
		if (HASN && pn8[x] == NT_N)
			continue;
		const u64 low = (low_e ^ (ballot((nt & pn8[x]) != 0) & either)) >> x;
		if (K >= 3)
			ge3 |= ge2 & low;
		if (K >= 2)
			ge2 |= ge1 & low;
		ge1 |= low;
	}
	return K == 1 ? ge1 : (K == 2 ? ge2 : ge3);
}

// ---- primer offset (offset.c:47-112) ---------------------------------------------------------
// Lane s owns the alignment of the primer starting at read position s and accumulates its
// log-probability in primer order x = 0..plen-1, which is the order in which offset.c:79-87 adds
// into bucket s.  That bucket is judged at index = s + plen (offset.c:72-77).
// Returns what computeoffset() returns: 0 = not found, else 1 + number of bases to clip.
// logdomain: compare the values before exp() (zero penalty only, where exp is the last, monotone step); `near` then
// reports whether a comparison was so close that exp() might have decided it otherwise.
template <bool LOGDOMAIN>
__device__ __forceinline__ int primer_offset_scan(const double threshold, const double penalty, const double *ltab, const uint16_t *S, int len, const uint8_t *primer, int plen, int lane, bool &near) {
	constexpr bool logdomain = LOGDOMAIN;
	bool close = false;
	// bestpr = exp(primerlen * threshold); with a zero penalty every candidate is exp(x) of some x
	// and exp is monotone, so we compare x (docs/NOTEBOOK.md 3.3).
	double best = (double) plen * threshold;
	if (!logdomain)
		best = exp(best);
	int besti = 0;
	// Pruning (zero penalty only).  Every term is a log-probability <= 0, and `bound` is a value some alignment is known
	// to reach (at first the threshold itself): an alignment whose sum is clearly below bound * (index + 1) cannot reach it,
	// and k of its terms each below a k-th of that floor put it there.  Short primers (<= 64 bases), block by block:
	//  * lanes as read POSITIONS: a ballot per primer base 0..7 says which positions agree with it, two more which
	//    positions' terms -- qual_score[q] if they agree, qual_score_err[q] if not -- are below the block's floor / k;
	//    shifted against each other and counted in bit-sliced counters they give the alignments with fewer than k such
	//    terms among their first eight bases (a block is 57 alignments then, so that all eight positions of each lie in
	//    the 64 the lanes hold).  k = 1, 2 or 3, the looser the bound the larger (any k is exact; it is a guess at what
	//    removes most);
	//  * while the bound is loose, the first alignment that disagrees with the primer in at most one of its first eight
	//    bases is summed by the whole wave (primer_alignment_sum) and its value becomes the bound -- it is the real primer
	//    nearly always, and after it a mismatching base or two put an alignment out;
	//  * no survivor: next block;  one or two: the wave sums them, one after the other;  more (repeats of the primer's
	//    first bases, a bound too loose, no primer in this read): the block is walked with lanes as ALIGNMENTS, four bases
	//    at a time.
	// Whatever is skipped is provably below the bound; whatever is summed is summed in the reference's order.
	const bool shortcuts = logdomain && plen <= 64;
	const int step = shortcuts ? 57 : 64;	// all eight positions of each alignment lie in the 64 the lanes hold
	unsigned pn8[8];	// the primer's first eight bases, N past its end (no term), in scalar registers
	{
		const unsigned pv = (shortcuts && lane < plen) ? primer[lane] : NT_N;
#pragma unroll
		for (int x = 0; x < 8; x++)
			pn8[x] = (unsigned) __builtin_amdgcn_readlane((int) pv, x);
	}
	const bool has_n = pn8[0] == NT_N || pn8[1] == NT_N || pn8[2] == NT_N || pn8[3] == NT_N || pn8[4] == NT_N || pn8[5] == NT_N || pn8[6] == NT_N || pn8[7] == NT_N;
	double bound = best;
	for (int base = 0; base + plen <= len - 1; base += step) {
		int nvalid = len - plen - base;	// alignments base .. base + nvalid - 1 exist (index <= len - 1)
		nvalid = nvalid > step ? step : nvalid;
		const int s = base + lane;
		const int index = s + plen;
		const bool have = lane < nvalid;
		const int ss = have ? s : 0;	// lanes past the end walk the first alignment and are ignored
		double acc = 0.0;
		bool mine = have;	// this lane's alignment is still in the race (and `acc` is, or will be, its full sum)
		bool summed = false;
		PK_COUNT(0);
		if (shortcuts) {
			const int p = base + lane;
			const unsigned e = S[p < len ? p : len - 1];
			const int q = clampq(qual_of(e));
			const unsigned nt = (unsigned) nt_of(e);
			const u64 valid = (1ull << nvalid) - 1ull;	// nvalid <= 57
			int known = -1;	// the alignment whose full sum the wave has made already ...
			double known_sum = 0.0;	// ... and that sum
			double floor_blk = bound * (double) (base + nvalid + plen) - 1e-6;	// the lowest floor of the block
			int k = 1 + uni((int) (floor_blk * -0.25));
			if (k >= 3) {
				// a loose bound (at first the threshold itself): the first alignment that disagrees with the primer in at
				// most one of its first eight bases is summed, and its value becomes the bound if it is better
				u64 d1 = 0, d2 = 0;	// alignments with at least 1, 2 disagreements (an N agrees with everything)
#pragma unroll
				for (int x = 0; x < 8; x++) {
					const u64 differ = ballot((nt & pn8[x]) == 0) >> x;
					d2 |= d1 & differ;
					d1 |= differ;
				}
				const u64 good = valid & ~d1, fair = valid & ~d2;
				PK_COUNT(1);
				if (fair != 0) {
					PK_COUNT(2);
					known = __builtin_ctzll(good != 0 ? good : fair);
					known_sum = primer_alignment_sum(ltab, S, base + known, primer, plen, lane);
					// its value, a shade too low (a bound may be anything an alignment is known to reach; this saves a division)
					const double kv = known_sum * (double) (1.0f / (float) (base + known + plen + 1)) * (1.0 + 1e-5);
					if (kv > bound) {
						bound = kv;
						floor_blk = bound * (double) (base + nvalid + plen) - 1e-6;
						k = 1 + uni((int) (floor_blk * -0.25));
					}
				}
				k = k > 3 ? 3 : k;
			}
			// any part works: k terms below it are together below the floor, up to a rounding the floor's margin dwarfs
			const double part = floor_blk * (k == 1 ? 1.0 : (k == 2 ? 0.5 : 1.0 / 3.0));
			const u64 low_s = ballot(ltab[PAMD_LT_SCORE + q] < part), low_e = ballot(ltab[PAMD_LT_ERR + q] < part);
			u64 killed;
			if (!has_n)
				killed = k == 1 ? low_terms<1, false>(nt, pn8, low_s, low_e, valid) : (k == 2 ? low_terms<2, false>(nt, pn8, low_s, low_e, valid) : low_terms<3, false>(nt, pn8, low_s, low_e, valid));
			else
				killed = k == 1 ? low_terms<1, true>(nt, pn8, low_s, low_e, valid) : (k == 2 ? low_terms<2, true>(nt, pn8, low_s, low_e, valid) : low_terms<3, true>(nt, pn8, low_s, low_e, valid));
			const u64 alive = valid & ~killed;
			PK_COUNT(2 + k);
			if (alive == 0)
				continue;	// no alignment of this block can beat what has been found
			PK_COUNT(6);
			mine = ((alive >> lane) & 1ull) != 0;
			if (__popcll(alive) <= 2) {
				u64 left = alive;
				while (left != 0) {
					const int l = __builtin_ctzll(left);
					left &= left - 1;
					if (l != known)
						PK_COUNT(7);
					const double total = l == known ? known_sum : primer_alignment_sum(ltab, S, base + l, primer, plen, lane);
					if (lane == l)
						acc = total;
				}
				summed = true;
			}
		}
		if (!summed) {
			PK_COUNT(8);
			// same order as offset.c:79-87 (x ascending); the loads of four steps are issued together
			const double floor_acc = bound * (double) (index + 1) - 1e-6;
			int x = 0;
			bool live = true;
			for (; x + 4 <= plen && live; x += 4) {
				unsigned e[4], pn[4];
				double t[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					e[u] = S[ss + x + u];
					pn[u] = primer[x + u];
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int q = clampq(qual_of(e[u]));
					t[u] = ltab[(((nt_of(e[u]) & pn[u]) != 0) ? PAMD_LT_SCORE : PAMD_LT_ERR) + q];
				}
#pragma unroll
				for (int u = 0; u < 4; u++)
					if (pn[u] != NT_N)
						acc += t[u];
				live = !logdomain || ballot(have && !(acc < floor_acc)) != 0;
				PK_COUNT(9);
			}
			if (!live)
				continue;	// no alignment of this block can beat what has been found
			for (; x < plen; x++) {
				const unsigned pn = primer[x];
				if (pn != NT_N) {
					const unsigned e = S[ss + x];
					const int q = clampq(qual_of(e));
					acc += ((nt_of(e) & pn) != 0) ? ltab[PAMD_LT_SCORE + q] : ltab[PAMD_LT_ERR + q];
				}
			}
		}
		double v = acc / (double) (index + 1);
		if (!logdomain)
			v = exp(v) - (double) index * penalty;
		const bool better = mine && v > best;	// strict: the earliest index wins ties (offset.c:74)
		// Compared before exp(), two values may differ and their exponentials not: exp'(x) <= 1 here, so a difference
		// below an ulp of 1 can vanish in rounding, and the reference then keeps the earlier index.  Such a scan is
		// repeated with the comparisons made after exp(), as the reference makes them (primer_offset below).
		close |= logdomain && better && v - best <= 4.5e-16;
		if (better) {
			best = v;
			besti = index + 1;
		}
		const u64 improved = ballot(better);
		if (logdomain && improved != 0) {
			double m;
			if ((improved & (improved - 1)) == 0) {	// one lane: its value, or the bound a summed guess has set already
				m = lane_f64(best, __builtin_ctzll(improved));
			} else {
				m = besti != 0 ? best : bound;
				for (int off = 32; off > 0; off >>= 1) {
					const double o = __shfl_xor(m, off);
					m = o > m ? o : m;
				}
				m = uni_f64(m);
			}
			bound = m > bound ? m : bound;
		}
	}
	// first-maximum across lanes: larger value wins, equal values -> smaller index (values whose exponentials may
	// coincide are ordered by their exponentials, as above)
	const u64 got = ballot(besti != 0);
	if ((got & (got - 1)) == 0) {	// nobody, or one lane only
		near = ballot(close) != 0;
		return got != 0 ? __builtin_amdgcn_readlane(besti, __builtin_ctzll(got)) : 0;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const double ob = __shfl_xor(best, off);
		const int oi = __shfl_xor(besti, off);
		const bool greater = ob > best, equal = ob == best;
		close |= logdomain && oi != 0 && besti != 0 && ob != best && fabs(ob - best) <= 4.5e-16;
		const bool take = (oi != 0) && (besti == 0 || greater || (equal && oi < besti));
		if (take) {
			best = ob;
			besti = oi;
		}
	}
	near = ballot(close) != 0;
	return uni(besti);
}

constexpr int PK_WAVES = 4;	// per workgroup
constexpr int PK_ROW = 928;	// bytes of one staged row: 456 bases + one 16-byte store of slack

// LOGDOMAIN: the fast kernel, for a zero penalty -- exp is then the last, monotone step of every value and the comparisons
// are made before it; it takes all pairs and lists those in which two alignments came so close that exp() may not tell
// them apart (once in 1e13 pairs).  !LOGDOMAIN: the careful kernel, every comparison made after exp() as offset.c:59,72-74
// makes them; it takes all pairs when there is a penalty (`-P`), else the listed ones.  Two kernels because the careful
// scan, compiled into the fast kernel, costs it three of its eight waves per SIMD in registers.
template <bool LOGDOMAIN>
__global__ __launch_bounds__(PK_WAVES * 64) void pamd_primer_kernel(const PrimerParams P, const bool listed) {
	__shared__ double ltab[96];	// qual_score at PAMD_LT_SCORE, qual_score_err at PAMD_LT_ERR
	__shared__ uint8_t lprimers[912];
	__shared__ __attribute__((aligned(16))) uint8_t rows[PK_WAVES][2][PK_ROW];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = uni(tid >> 6);
	for (int i = tid; i < 96; i += PK_WAVES * 64)
		ltab[i] = P.tables[i];
	for (int i = tid; i < 912; i += PK_WAVES * 64)
		lprimers[i] = P.primers[i];
	__syncthreads();
	const u64 n_work = listed ? (u64) *P.in_count : P.n_pairs;
	// (work from a counter, not in fixed shares: a read without its primer costs five times one with it)
	u64 took = 0;
	unsigned took_left = 0;
	auto take = [&]() -> u64 {
		if (took_left == 0) {
			unsigned got = 0;
			if (lane == 0)
				got = atomicAdd(P.cursor, 16u);
			took = (u64) (unsigned) uni((int) got);
			took_left = 16;
		}
		took_left--;
		return took++;
	};
	u64 at = take();
	const uint16_t *F = reinterpret_cast<const uint16_t *>(rows[wave][0]), *R = reinterpret_cast<const uint16_t *>(rows[wave][1]);
	const bool in_f = (unsigned) lane * 16u < P.fwd_row_bytes, in_r = (unsigned) lane * 16u < P.rev_row_bytes;
	uint4 fv = make_uint4(0, 0, 0, 0), rv = make_uint4(0, 0, 0, 0);
	int flen = 0, rlen = 0;
	u64 pair = 0;
	if (at < n_work) {
		pair = listed ? (u64) P.in_list[at] : at;
		flen = P.flen[pair];
		rlen = P.rlen[pair];
		if (in_f)
			fv = reinterpret_cast<const uint4 *>(P.fwd + pair * P.fwd_row_bytes)[lane];
		if (in_r)
			rv = reinterpret_cast<const uint4 *>(P.rev + pair * P.rev_row_bytes)[lane];
	}
	while (at < n_work) {
		if (in_f)
			reinterpret_cast<uint4 *>(rows[wave][0])[lane] = fv;
		if (in_r)
			reinterpret_cast<uint4 *>(rows[wave][1])[lane] = rv;
		const int fl = uni(flen), rl = uni(rlen);
		const u64 this_pair = pair;
		at = take();
		if (at < n_work) {	// the next pair's rows travel while this one is scanned
			pair = listed ? (u64) P.in_list[at] : at;
			flen = P.flen[pair];
			rlen = P.rlen[pair];
			if (in_f)
				fv = reinterpret_cast<const uint4 *>(P.fwd + pair * P.fwd_row_bytes)[lane];
			if (in_r)
				rv = reinterpret_cast<const uint4 *>(P.rev + pair * P.rev_row_bytes)[lane];
		}
		// assembler.c:262-288: the reverse primer is looked for only in pairs that have the forward one.  One copy of the
		// scan serves both reads (everything that selects the read is wave-uniform).
		int found[2] = { 0, 0 };
		bool again = false;
		// a listed pair whose word is not the "unknown" mark has its offsets already (the lean kernel found them, pamd_fast.hip)
		if (listed && LOGDOMAIN && P.primer_offsets[this_pair] != 0xFFFFFFFFu)
			continue;
#pragma unroll 1
		for (int which = 0; which < 2; which++) {
			const int plen = which == 0 ? P.fprimer_len : P.rprimer_len;
			const int len = which == 0 ? fl : rl;
			if (plen <= 0 || (which == 1 && found[0] == 0 && P.fprimer_len > 0))
				continue;
			int r = 0;
			if (plen <= len) {
				bool near = false;
				r = primer_offset_scan<LOGDOMAIN>(P.threshold, P.primer_penalty, ltab, which == 0 ? F : R, len, lprimers + (which == 0 ? 0 : 456), plen, lane, near);
				again |= near;
			}
			if (which == 0)
				found[0] = r;
			else
				found[1] = r;
		}
		if (lane == 0) {
			if (LOGDOMAIN && again)
				P.redo_list[atomicAdd(P.redo_count, 1u)] = (uint32_t) this_pair;
			else
				P.primer_offsets[this_pair] = (uint32_t) found[0] | ((uint32_t) found[1] << 16);
		}
	}
}

}	// namespace

// list == nullptr: pairs 0 .. n_pairs-1; else the pairs list[0 .. *count - 1] (device memory: the lean kernel's deferred pairs)
int pamd_launch_primers(const PamdKernelParams &p, int cu_count, void *stream, const uint32_t *list, const unsigned *count) {
	if (p.n_pairs == 0)
		return PAMD_OK;
	if (p.n_pairs > 0xFF000000ull) {	// (the work counter is 32 bits wide)
		pamd_set_error("primer kernel: more than 2^32 - 2^24 pairs in one launch");
		return PAMD_EINVAL;
	}
	// every wave walks its share of the batch: exactly as many workgroups as are resident at once
	static std::atomic<int> per_cu[2];	// (assemblers launch from their own threads: the cached answer is the same for all)
	const int which = p.primer_penalty != 0.0 ? 0 : 1;
	int resident = per_cu[which].load(std::memory_order_relaxed);
	if (resident == 0) {
		int nb = 0;
		const hipError_t oe = which ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pamd_primer_kernel<true>, PK_WAVES * 64, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pamd_primer_kernel<false>, PK_WAVES * 64, 0);
		resident = (oe == hipSuccess && nb > 0) ? nb : 4;
		per_cu[which].store(resident, std::memory_order_relaxed);
	}
	size_t blocks = (size_t) (cu_count > 0 ? cu_count : 256) * (size_t) resident;
	const size_t want = (p.n_pairs + PK_WAVES - 1) / PK_WAVES;
	if (blocks > want)
		blocks = want;
	const bool listed = list != nullptr;
	PrimerParams q;
	q.fwd = p.fwd;
	q.rev = p.rev;
	q.flen = p.flen;
	q.rlen = p.rlen;
	q.tables = p.tables;
	q.primers = p.primers;
	q.primer_offsets = p.primer_offsets;
	q.in_list = list;
	q.in_count = count;
	q.redo_list = p.primer_offsets + p.n_pairs;	// the buffer holds 2 n words + the count (pamd_api.cpp)
	q.redo_count = reinterpret_cast<unsigned *>(p.primer_offsets + 2 * p.n_pairs);
	q.cursor = q.redo_count + 1;
	q.n_pairs = p.n_pairs;
	q.fwd_row_bytes = p.fwd_row_bytes;
	q.rev_row_bytes = p.rev_row_bytes;
	q.fprimer_len = p.fprimer_len;
	q.rprimer_len = p.rprimer_len;
	q.threshold = p.threshold;
	q.primer_penalty = p.primer_penalty;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (p.primer_penalty != 0.0) {
		if (hipMemsetAsync(q.cursor, 0, sizeof(unsigned), st) != hipSuccess) {
			pamd_set_error("primer kernel: hipMemsetAsync failed");
			return PAMD_ENODEVICE;
		}
		hipLaunchKernelGGL(pamd_primer_kernel<false>, dim3((unsigned) blocks), dim3(PK_WAVES * 64), 0, st, q, listed);
	} else {
		if (hipMemsetAsync(q.redo_count, 0, 2 * sizeof(unsigned), st) != hipSuccess) {
			pamd_set_error("primer kernel: hipMemsetAsync failed");
			return PAMD_ENODEVICE;
		}
		hipLaunchKernelGGL(pamd_primer_kernel<true>, dim3((unsigned) blocks), dim3(PK_WAVES * 64), 0, st, q, listed);
		if (hipMemsetAsync(q.cursor, 0, sizeof(unsigned), st) != hipSuccess) {
			pamd_set_error("primer kernel: hipMemsetAsync failed");
			return PAMD_ENODEVICE;
		}
		q.in_list = q.redo_list;
		q.in_count = q.redo_count;
		hipLaunchKernelGGL(pamd_primer_kernel<false>, dim3(8), dim3(PK_WAVES * 64), 0, st, q, true);	// next to nothing to do
	}
#ifdef PK_DEBUG
	{
		unsigned long long h[16];
		(void) hipDeviceSynchronize();
		(void) hipMemcpyFromSymbol(h, HIP_SYMBOL(pk_events), sizeof h);
		fprintf(stderr, "pk_events (%llu pairs): blocks %llu loose %llu guessed %llu k1 %llu k2 %llu k3 %llu alive %llu extra_sums %llu lockstep_blocks %llu lockstep_batches %llu exp_domain %llu\n", (unsigned long long) p.n_pairs, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
		memset(h, 0, sizeof h);
		(void) hipMemcpyToSymbol(HIP_SYMBOL(pk_events), h, sizeof h);
	}
#endif
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		pamd_set_error("primer kernel launch: %s", hipGetErrorString(e));
		return PAMD_ENODEVICE;
	}
	return PAMD_OK;
}
