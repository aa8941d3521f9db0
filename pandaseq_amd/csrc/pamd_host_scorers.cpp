// pamd_host_scorers.cpp -- the seven built-in algorithm classes' two public functions, on the host.
//
// struct panda_algorithm_class is public (pandaseq-common.h:594-605): a caller, or a plugin, may score an overlap or a merged
// base through `overlap_probability` / `match_probability` of ANY class object, the library's own included (algo.c:27-32 does
// for every algorithm).  These are those functions for the seven classes -- what algo_simple_bayes.c:33-75, algo_pear.c:32-68,
// algo_rdp_mle.c:29-74, algo_uparse.c:33-75, algo_flash.c:30-81, algo_ea_util.c:30-68 and algo_stitch.c:30-67 compute, written
// against this library's host copy of the tables (pamd_tables.cpp) -- and they are what the per-pair entry point
// panda_assembler_assemble() runs on (pamd_compat.cpp host_assemble(): one pair handed over at a time is scored on the caller's
// thread; arrays and streams of pairs go to the kernels, which implement the same arithmetic in pamd_kernels.hip / pamd_fast.hip
// and are tested equal to these, record for record).  Product code, written for this library: the test-side checker is never involved.
#include <cmath>
#include <cstddef>
#include "pamd_internal.h"
#include "pamd_host_scorers.h"

namespace {

const PamdTables &tabs() {	// (built on first use; afterwards one guard-byte test per call)
	static const PamdTables *const t = [] {
		PamdTables *p = new PamdTables;
		pamd_build_tables(p);
		return p;
	}();
	return *t;
}

inline int clampq(int q) {	// PHREDCLAMP, prob.h:23
	return q > 46 ? 46 : (q < 0 ? 0 : q);
}

// The positions an overlap pairs up (the loop header every scorer shares): position i of `overlap` meets forward base
// flen - overlap + i and reverse base rlen - 1 - i; positions that fall off either read are skipped.
template <typename Visit>
inline void walk(size_t flen, size_t rlen, size_t overlap, Visit visit) {
	const ptrdiff_t f0 = (ptrdiff_t) flen - (ptrdiff_t) overlap;
	// i >= -f0 keeps the forward index >= 0; i <= rlen - 1 keeps the reverse index >= 0; forward index < flen always
	const size_t lo = f0 < 0 ? (size_t) -f0 : 0, hi = overlap < rlen ? overlap : rlen;
	for (size_t i = lo; i < hi; i++)
		visit((size_t) (f0 + (ptrdiff_t) i), rlen - 1 - i);
}

struct Counts {
	size_t matches = 0, mismatches = 0, unknowns = 0, walked = 0;
};

inline Counts count(const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	Counts c;
	walk(flen, rlen, overlap, [&](size_t fi, size_t ri) {
		const panda_nt a = f[fi].nt, b = r[ri].nt;
		c.walked++;
		if (a == (panda_nt) 15 || b == (panda_nt) 15)	// PANDA_NT_IS_N
			c.unknowns++;
		else if ((a & b) != 0)
			c.matches++;
		else
			c.mismatches++;
	});
	return c;
}

// simple_bayesian and uparse: three counts and two constants (data: q, pmatch, pmismatch); the bases outside the overlap are
// counted in size_t, wrap-around included, as the reference's expression does
double counted_score(const double *data, size_t flen, size_t rlen, size_t overlap, const Counts &c) {
	const double nn = tabs().qual_nn;
	const size_t free_bases = (overlap >= flen && overlap >= rlen) ? c.unknowns : flen + rlen - 2 * overlap + c.unknowns;
	return nn * (double) free_bases + (double) c.matches * data[1] + (double) c.mismatches * data[2];
}

double sb_overlap(void *data, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	return counted_score(static_cast<const double *>(data), flen, rlen, overlap, count(f, flen, r, rlen, overlap));
}

double sb_match(void *, bool match, char a, char b) {
	const PamdTables &T = tabs();
	return (match ? T.match_sb : T.mismatch_sb)[clampq(a) * PAMD_NQ + clampq(b)];
}

// pear: a table value per position, indexed by the FORWARD read's quality at the forward and at the REVERSE index (sic,
// algo_pear.c:52,54); where that index lies past the forward read the reference reads beyond its array: quality 0 here, as in
// the kernels (include/pandaseq_amd.h)
double pear_overlap(void *data, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	const PamdTables &T = tabs();
	const double random_base = static_cast<const double *>(data)[0];
	double p = 0;
	walk(flen, rlen, overlap, [&](size_t fi, size_t ri) {
		const panda_nt a = f[fi].nt, b = r[ri].nt;
		if (a == (panda_nt) 15 || b == (panda_nt) 15) {
			p -= random_base;
			return;
		}
		const int at = clampq(f[fi].qual) * PAMD_NQ + (ri < flen ? clampq(f[ri].qual) : 0);
		p += ((a & b) != 0 ? T.match_pear : T.mismatch_pear)[at];
	});
	return p;
}

double pear_match(void *, bool match, char a, char b) {
	const PamdTables &T = tabs();
	return (match ? T.match_pear : T.mismatch_pear)[clampq(a) * PAMD_NQ + clampq(b)];
}

// rdp_mle: simple_bayesian's match table or its own mismatch table, less qual_nn, per position; an N is a base like any other
double rdp_overlap(void *, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	const PamdTables &T = tabs();
	double p = 0;
	walk(flen, rlen, overlap, [&](size_t fi, size_t ri) {
		const int at = clampq(f[fi].qual) * PAMD_NQ + clampq(r[ri].qual);
		p += ((f[fi].nt & r[ri].nt) != 0 ? T.match_sb : T.mismatch_rdp)[at] - T.qual_nn;
	});
	return p;
}

double rdp_match(void *, bool match, char a, char b) {
	const PamdTables &T = tabs();
	if (match)
		return T.score[clampq(a >= b ? a : b)];
	return T.mismatch_rdp_asm[clampq(a) * PAMD_NQ + clampq(b)];
}

double uparse_overlap(void *data, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	return counted_score(static_cast<const double *>(data), flen, rlen, overlap, count(f, flen, r, rlen, overlap));
}

double uparse_match(void *, bool match, char a, char b) {
	const PamdTables &T = tabs();
	return (match ? T.match_uparse : T.mismatch_uparse)[clampq(a) * PAMD_NQ + clampq(b)];
}

// flash: the logarithm of an INTEGER quotient (sic, algo_flash.c:60): 0 when nothing in the overlap matches, -inf otherwise;
// an N is a mismatch
double flash_overlap(void *, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	const Counts c = count(f, flen, r, rlen, overlap);
	if (c.walked == 0)
		return -2;
	return std::log((double) ((c.walked - c.matches) / c.walked));
}

double flash_match(void *, bool match, char a, char b) {
	int score;
	if (match) {
		score = a > b ? clampq(a) : clampq(b);
	} else {
		score = clampq(a) - clampq(b);
		score = score < 0 ? -score : score;
		score = score < 2 ? 2 : score;
	}
	return tabs().score[score];
}

double ea_util_overlap(void *, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	const Counts c = count(f, flen, r, rlen, overlap);
	const size_t mismatches = c.walked - c.matches;	// an N is a mismatch
	return std::log((((double) mismatches) * mismatches + 1) / c.walked);
}

double ea_util_match(void *, bool, char a, char b) {
	return tabs().score[a > b ? clampq(a) : clampq(b)];
}

// stitch: matches less mismatches in size_t -- more mismatches than matches wraps to just under 2^64 (algo_stitch.c:56)
double stitch_overlap(void *, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen, size_t overlap) {
	const Counts c = count(f, flen, r, rlen, overlap);
	const size_t score = c.matches - c.mismatches;
	return std::log(score / (double) (flen + rlen));
}

}	// namespace

const PamdHostScorer pamd_host_scorers[PAMD_NUM_ALGOS] = {
	// indexed by PAMD_ALGO_*
	{ sb_overlap, sb_match },
	{ pear_overlap, pear_match },
	{ rdp_overlap, rdp_match },
	{ uparse_overlap, uparse_match },
	{ flash_overlap, flash_match },
	{ ea_util_overlap, ea_util_match },
	{ stitch_overlap, sb_match },	// (stitch does not reconstruct qualities: simple_bayesian's tables, algo_stitch.c:59-67)
};
