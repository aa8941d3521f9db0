// pamd_device.h -- device-side helpers shared by the two assemble kernels (pamd_kernels.hip: the general
// kernel, every configuration and every input; pamd_fast.hip: the lean kernel for the common case).
// Everything here is __forceinline__ device code; include it inside a HIP translation unit only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pamd_internal.h"

namespace pamd_dev {

typedef unsigned long long u64;

// per-pair status used between the lean kernel and its finish only: the pair goes to the general kernel
// through the deferred list and gets no record from the lean kernel
constexpr int PAMD_ST_DEFER = 8;

__device__ __forceinline__ void wave_sync() {
	// Lanes of one wavefront exchange data through LDS.  The LDS unit executes one wave's DS
	// operations in issue order, so ordering the compiler is all that is needed.
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// __ballot() takes an int: a bool predicate is first turned into 0/1 in a vector register (v_cndmask) and compared with
// zero again (v_cmp) -- two VALU instructions per ballot.  The builtin takes the predicate as it is when it is ONE
// comparison; a conjunction is still materialised, so hot code writes ballot(a) & ballot(b) (two compares it needed
// anyway and a scalar AND).
__device__ __forceinline__ u64 ballot(bool pred) {
	return __builtin_amdgcn_ballot_w64(pred);
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

// ---- wave-wide f64 sum without the LDS -----------------------------------------------------------------
// __shfl_xor() is a ds_bpermute: an LDS round trip the wave parks on, six in a row for a butterfly.  DPP moves
// stay in the vector ALU: xor 1 and xor 2 inside quads, then the mirrored half-row and row, give every lane
// its row's sum (16 lanes); the four row sums are read back as scalars.  The order differs from the butterfly's,
// which only callers that accept any order use.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
	union {
		double d;
		int i[2];
	} x, y;
	x.d = v;
	y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], CTRL, 0xF, 0xF, true);
	y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], CTRL, 0xF, 0xF, true);
	return y.d;
}

// the same move into the rows ROWS selects only; the other rows receive 0.0
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64_rows(double v) {
	union {
		double d;
		int i[2];
	} x, y;
	x.d = v;
	y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], CTRL, ROWS, 0xF, false);
	y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], CTRL, ROWS, 0xF, false);
	return y.d;
}

__device__ __forceinline__ double wave_sum_any_order(double v) {
	v += dpp_f64<0xB1>(v);	// quad_perm:[1,0,3,2]
	v += dpp_f64<0x4E>(v);	// quad_perm:[2,3,0,1]
	v += dpp_f64<0x141>(v);	// row_half_mirror
	v += dpp_f64<0x140>(v);	// row_mirror: every lane holds its row's sum
	v += dpp_f64_rows<0x142, 0xA>(v);	// row_bcast:15 into rows 1 and 3: row 0's sum joins row 1's, row 2's joins row 3's
	v += dpp_f64_rows<0x143, 0xC>(v);	// row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
	return lane_f64(v, 63);
}

__device__ __forceinline__ int clampq(int q) {	// PHREDCLAMP, prob.h:23
	return q > 46 ? 46 : (q < 0 ? 0 : q);
}

// Five of the seven scorers judge an overlap from three integer counts -- unknown (an N on either side),
// matching and mismatching bases among those present on both reads -- and differ only in the closed form:
// simple_bayesian (algo_simple_bayes.c:61-65), uparse (algo_uparse.c:61-65), flash (algo_flash.c:60),
// ea_util (algo_ea_util.c:56) and stitch (algo_stitch.c:56).  pear and rdp_mle sum per-base table values.
template <int ALGO>
__device__ __forceinline__ constexpr bool counting_algo() {
	return ALGO != PAMD_ALGO_PEAR && ALGO != PAMD_ALGO_RDP_MLE;
}

// walked = bases present on both reads (the reference's real_overlap)
// NONNEG: the caller knows ov <= min(flen, rlen) (no configured maximum overlap: the lean kernel), so the count of bases
// outside the overlap cannot wrap
template <int ALGO, bool NONNEG = false>
__device__ __forceinline__ double count_score(double qual_nn, double pmatch, double pmismatch, const double *count_table, int flen, int rlen, int ov, int unknowns, int matches, int walked) {
	const int mismatches = walked - unknowns - matches;
	if (ALGO == PAMD_ALGO_FLASH) {
		// log(mismatches / real_overlap) with an INTEGER quotient (sic): 1 when nothing matches, else 0;
		// an N counts as a mismatch here
		if (walked == 0)
			return -2.0;
		return walked - matches == walked ? 0.0 : -__builtin_inf();
	}
	if (ALGO == PAMD_ALGO_EA_UTIL) {
		// log((mismatches^2 + 1) / real_overlap), an N counting as a mismatch: looked up (PamdKernelParams::count_table)
		return count_table[walked * PAMD_CT_EA + (walked - matches)];
	}
	if (ALGO == PAMD_ALGO_STITCH) {
		// the score is a size_t: more mismatches than matches wraps to just under 2^64
		// log(score / (flen + rlen)), looked up likewise (the table holds the wrapped scores too)
		return count_table[(matches - mismatches + 450) * PAMD_CT_ST + flen + rlen];
	}
	// simple_bayesian / uparse: size_t arithmetic inside the parenthesis (algo_simple_bayes.c:61-65), wrap
	// included: when flen + rlen - 2*ov + unknowns is negative the reference's unsigned value is
	// 2^64 - |x|, whose conversion to double goes through the 64-bit path; otherwise one 32-bit conversion is exact
	const int fbases = (ov >= flen && ov >= rlen) ? unknowns : flen + rlen - 2 * ov + unknowns;
	const double free_bases = (NONNEG || fbases >= 0) ? (double) (unsigned) fbases : (double) (u64) (long long) fbases;
	return qual_nn * free_bases + (double) matches * pmatch + (double) mismatches * pmismatch;
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

// ---- fast-sums finish, 64 pairs at a time -----------------------------------------------------------
// What is left to do for a pair once its row is written -- the f64 division of the quality sum, LOWQ,
// the check modules, the 48-byte record, the counters -- is wave-uniform work: done per pair it occupies a
// whole wave for a few hundred instructions with one useful lane.  Instead the pair's scalars are dropped
// into lane `slot` of ten registers (one v_cndmask each), and every 64 pairs all lanes finish their
// pair at once.  Rows of pairs that turn out LOWQ / rejected stay in the arena as holes, as in exact mode.
struct Pend64 {
	unsigned off_lo, off_hi;	// row offset in seq_nt / seq_p (~0 = none)
	unsigned q_lo, q_hi;		// the any-order quality SUM (or, flagged, the finished quality)
	unsigned b_lo, b_hi;		// estimated_overlap_probability
	unsigned len_bo;		// seq_len | overlap << 16
	unsigned ex_mm;			// examined | mismatches << 16
	unsigned dg_fo;			// degenerates | forward offset << 16
	unsigned ro_st;			// reverse offset | status << 16 | slow << 24 | "q is final" << 25 | smallest phred (63 = none) << 26
	unsigned pid;			// the pair's index, where a wave takes its pairs from a work queue (general kernel); the lean kernel with
					// exact sums: where the row's overlap region begins | ends << 16
	unsigned p_lo, p_hi;		// (finish64<.., LAZYP>) the row's smallest log-p above qual_score[0]
};

__device__ __forceinline__ unsigned put_lane(unsigned reg, unsigned value, bool mine) {
	return mine ? value : reg;	// one v_cndmask: the compare is shared by the twelve fields
}


// lane s finishes the wave's s-th parked pair: pair index first_pair + s * pair_step, or (LIST: the general kernel working
// through the lean kernel's deferred pairs) entry first_pair + s * pair_step of P.pair_list.
// DEFER (the lean kernel): a pair parked as PAMD_ST_DEFER gets no record and no counter here; its index is appended
// to P.defer_list -- one global atomic per 64 pairs -- and the general kernel assembles it from scratch.
// One ordered chain of a row's log-probabilities: acc += row[j] for j = lo .. hi-1, left to right (assembler.c:171,226,236),
// eight values (four 16-byte loads) in flight per lane.  Every lane walks its OWN pair's row, so a load instruction
// touches 64 different lines: measured on 8 M config-1 pairs, this depth costs the lean kernel 18 % against any-order
// sums; sixteen values in flight cost 27 % (the lines no longer stay in the L1), and walking a row's three chains at once
// with 8-byte loads doubles the kernel's time.
__device__ __forceinline__ double ordered_row_sum(const double *row, int lo, int hi) {
	double acc = 0.0;
	int j = lo;
	if (j < hi && (j & 1)) {	// reach a 16-byte boundary
		acc += row[j];
		j++;
	}
	for (; j + 8 <= hi; j += 8) {
		const double2 v0 = *reinterpret_cast<const double2 *>(row + j), v1 = *reinterpret_cast<const double2 *>(row + j + 2);
		const double2 v2 = *reinterpret_cast<const double2 *>(row + j + 4), v3 = *reinterpret_cast<const double2 *>(row + j + 6);
		acc += v0.x;
		acc += v0.y;
		acc += v1.x;
		acc += v1.y;
		acc += v2.x;
		acc += v2.y;
		acc += v3.x;
		acc += v3.y;
	}
	for (; j < hi; j++)
		acc += row[j];
	return acc;
}

// EXACTQ (the lean kernel with exact sums): the quality sums are made here, in the reference's order, every lane walking
// the three chains of its own pair's row (which this wave wrote; pd.pid says where the overlap region begins and ends).
// QUEUED: the pairs' indices are in pd.pid (the wave took them from a work queue).
// LAZYQ (the lean kernel's any-order sums, pamd_fast.hip qsum_insert()): the s-th parked pair sits in lane bitrev6(s) -- that is
// where the merge tree of the pairs' quality sums leaves its sum --, and the test "is the quality too close to the LOWQ threshold
// to call from a sum in any order" is made here, for 64 pairs at once: such a pair is deferred.
// LAZYP (with LAZYQ and a min_phred module): pd.p_lo / p_hi hold the row's smallest log-p above qual_score[0] (out of a merge tree
// of its own), bit 26 of pd.ro_st says whether one lay at or below it; the phred (nt.c:126-150) is looked up here, in `score`.
template <bool LIST, bool DEFER, bool EXACTQ = false, bool QUEUED = false, bool LAZYQ = false, bool LAZYP = false>
__device__ void finish64(const PamdKernelParams &P, unsigned *lctr, const Pend64 &pd, int npend, u64 first_pair, u64 pair_step, int lane_id, const double *score = nullptr) {
	int status = -1, reject = -1;
	bool slow = false;
	if (EXACTQ)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// this wave's row stores have reached L2
	const int lane = LAZYQ ? (int) (__builtin_bitreverse32((unsigned) lane_id) >> 26) : lane_id;	// which parked pair this lane finishes
	u64 pair = first_pair + (u64) lane * pair_step;
	if (QUEUED)
		pair = (u64) pd.pid;
	else if (LIST && lane < npend)
		pair = P.pair_list[pair];
	bool deferred = false;
	if (DEFER) {
		deferred = lane < npend && ((pd.ro_st >> 16) & 0xFFu) == (unsigned) PAMD_ST_DEFER;
		if (LAZYQ) {
			union {
				double d;
				unsigned u[2];
			} qs;
			qs.u[0] = pd.q_lo;
			qs.u[1] = pd.q_hi;
			const double len1 = (double) ((int) (pd.len_bo & 0xFFFFu) + 1);
			const bool close = lane < npend && ((pd.ro_st >> 16) & 0xFFu) == (unsigned) PAMD_ST_OK && fabs(qs.d - P.threshold * len1) <= 1e-9 * len1;
			const u64 cmask = ballot(close);
			if (cmask != 0 && lane_id == 0 && P.phase_cycles != nullptr)
				atomicAdd(&P.phase_cycles[4], (unsigned long long) __popcll(cmask));
			deferred |= close;	// (its row stays behind in the arena as a hole)
		}
		const u64 dmask = ballot(deferred);
		if (dmask != 0) {
			unsigned base = 0;
			if (lane_id == 0)
				base = atomicAdd(P.defer_count, (unsigned) __popcll(dmask));
			base = (unsigned) uni((int) base);
			if (deferred)
				P.defer_list[base + (unsigned) __popcll(dmask & ((1ull << lane_id) - 1ull))] = (uint32_t) pair;
		}
	}
	if (lane < npend && !deferred) {
		const u64 row = (u64) pd.off_lo | ((u64) pd.off_hi << 32);
		union {
			double d;
			unsigned u[2];
		} q, e;
		q.u[0] = pd.q_lo;
		q.u[1] = pd.q_hi;
		e.u[0] = pd.b_lo;
		e.u[1] = pd.b_hi;
		status = (int) ((pd.ro_st >> 16) & 0xFFu);
		if (status != PAMD_ST_OK)
			e.d = 0.0;	// (assembler.c:247 sets it once the pair has merged; a scorer's running best of a pair that found no overlap is not a result)
		slow = ((pd.ro_st >> 24) & 1u) != 0;
		const bool q_final = ((pd.ro_st >> 25) & 1u) != 0;
		const int seqlen = (int) (pd.len_bo & 0xFFFFu), bo = (int) (pd.len_bo >> 16);
		unsigned dg_fo = pd.dg_fo;
		if (EXACTQ) {
			if (status == PAMD_ST_OK && row != ~0ull) {
				// plain cached loads: this wave's own stores have been waited for (below), and a CU's vector L1 is
				// coherent for its own stores.  pd.pid: where the overlap region of the row begins | ends << 16
				const int A0 = (int) (pd.pid & 0xFFFFu), AF = (int) (pd.pid >> 16);
				const double *rowp = P.seq_p + row;
				const double fq = ordered_row_sum(rowp, 0, A0);
				const double rq = ordered_row_sum(rowp, AF, seqlen);
				const double oq = ordered_row_sum(rowp, A0, AF);
				q.d = (fq + rq + oq) / (double) (seqlen + 1);	// assembler.c:244
			}
		}
		double quality = 0.0;
		u64 off = ~0ull;
		if (status == PAMD_ST_OK && row == ~0ull) {
			// the output arena was too small for this row: the launch reports it (pamd_assemble_status), the pair is
			// counted as seen and as nothing else
			atomicAdd(P.error_flag, 1ull);
			status = PAMD_ST_NOSPACE;
		}
		if (status == PAMD_ST_OK) {
			// assembler.c:244: the divisor is the sequence length + 1.  A sum that lands within 1e-9 of the LOWQ
			// threshold was already redone in reference order when the pair was parked (flagged final).
			quality = (q_final || EXACTQ) ? q.d : q.d / (double) (seqlen + 1);
			if (quality < P.threshold) {	// assembler.c:334-338
				status = PAMD_ST_LOWQ;
			} else {
				int phmin;
				if (LAZYP) {
					union {
						double d;
						unsigned u[2];
					} pm;
					pm.u[0] = pd.p_lo;
					pm.u[1] = pd.p_hi;
					phmin = result_phred(score, pm.d);
					if (((pd.ro_st >> 26) & 1u) != 0 && phmin > 1)
						phmin = 1;
					if (seqlen <= 0)
						phmin = 127;	// an empty row passes any threshold
				} else {
					const int ph6 = (int) (pd.ro_st >> 26);
					phmin = ph6 == 63 ? 127 : ph6;
				}
				for (int mi = 0; mi < P.n_modules && reject < 0; mi++) {	// module.c:124-136
					const PamdModule md = P.modules[mi];
					if (md.kind == 1) {	// plugin_min_phred.c:8-22
						if (phmin < md.phred)
							reject = mi;
					} else if (md.kind == 2) {	// plugin_min_overlapbits.c:17-23
						if (!(md.nats <= e.d))
							reject = mi;
					}
				}
				if (reject >= 0)
					status = PAMD_ST_REJECT;
				else
					off = row;
			}
		}
		q.d = quality;
		uint4 *dst = reinterpret_cast<uint4 *>(&P.results[pair]);
		dst[0] = make_uint4(q.u[0], q.u[1], e.u[0], e.u[1]);
		dst[1] = make_uint4((unsigned) off, (unsigned) (off >> 32), pd.len_bo, pd.ex_mm);
		dst[2] = make_uint4(dg_fo, (pd.ro_st & 0xFFFFu) | ((unsigned) status << 16) | (((unsigned) reject & 0xFFu) << 24), slow ? 1u : 0u, 0u);
		if (status == PAMD_ST_OK)
			atomicAdd(&lctr[PAMD_CTR_OVERLAP0 + bo], 1u);
		else if (status == PAMD_ST_REJECT)
			atomicAdd(&lctr[PAMD_CTR_REJECTED0 + reject], 1u);
	}
	// workgroup-local counters (assembler.h:59-67,75): one LDS atomic per status per group
	const int ctr_of[6] = { PAMD_CTR_OK, PAMD_CTR_BADR, PAMD_CTR_NOFP, PAMD_CTR_NORP, PAMD_CTR_NOALGN, PAMD_CTR_LOWQ };
#pragma unroll
	for (int st = 0; st < 6; st++) {
		const int c = __popcll(ballot(status == st));
		if (c != 0 && lane_id == 0)
			atomicAdd(&lctr[ctr_of[st]], (unsigned) c);
	}
	const int nslow = __popcll(ballot(slow));
	const int ndefer = DEFER ? __popcll(ballot(deferred)) : 0;
	if (lane_id == 0) {
		atomicAdd(&lctr[PAMD_CTR_COUNT], (unsigned) (npend - ndefer));
		if (nslow)
			atomicAdd(&lctr[PAMD_CTR_SLOW], (unsigned) nslow);
	}
}

}	// namespace pamd_dev
