/*
 * panda_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 * See panda_oracle.h for scope and pinning status.  Every function names the reference
 * lines it restates; nothing here is copied from the reference, it is re-derived from the
 * behavioural spec and checked against what can be checked.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (the reference is built without FMA
 * contraction, Makefile.am:62-66; x86-64 gcc does not contract by default but we say so).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "panda_oracle.h"

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif

#define NT_N 15
#define KMER_LEN 8

static inline int clampq(int q) {	/* PHREDCLAMP, prob.h:23 */
	return q > PO_PHREDMAX ? PO_PHREDMAX : (q < 0 ? 0 : q);
}

static inline int is_degenerate(unsigned nt) {	/* PANDA_NT_IS_DEGN == popcount != 1, pandaseq-nt.h:55 */
	return __builtin_popcount(nt & 0xFFu) != 1;
}

/* ------------------------------------------------------------------ tables */

/* tablebuilder.c:86,124,147 print every entry with "%g"; the C compiler then parses that text.
 * Round-tripping through the same text reproduces the 6-significant-digit values exactly. */
static double six_digits(double v) {
	char buf[64];
	snprintf(buf, sizeof buf, "%g", v);
	return strtod(buf, NULL);
}

static double phred_p(int q) {	/* PROBABILITY, prob.h:21 */
	return pow(10.0, (-(double) q) / 10.0);
}

/* mktable.c:23-31 */
static double f_match(double p, double q) {
	return (1 - p) * (1 - q) + p * q / 3;
}

/* mktable.c:33-41 and :84-92 (the rdp_mle mismatch formula is the same expression) */
static double f_mismatch(double p, double q) {
	return (1 - p) * q / 3 + (1 - q) * p / 3 + 2 * p * q / 9;
}

/* mktable.c:43-51 */
static double f_match_pear(double p, double q) {
	return (1 - (1 - q) * p / 3 - (1 - p) * q / 3 - 2 * (1 - p) * (1 - q) / 9);
}

/* mktable.c:53-61 */
static double f_mismatch_pear(double p, double q) {
	return (1 - p) * q / 3 + (1 - q) * p / 3 + p * q / 2;
}

/* mktable.c:94-104 */
static double f_mismatch_rdp_asm(double p, double q) {
	double mn = (p <= q) ? p : q;
	double v = 1 - (mn - p * q / 3.0) / (p + q - 4.0 / 3.0 * p * q);
	return (v == 0) ? DBL_MIN : v;
}

/* mktable.c:106-117 */
static double f_match_uparse(double p, double q) {
	double v = 1 - p * q / (1 - p - q + 4 * p * q / 3);
	return (v <= 0) ? DBL_MIN : v;
}

/* mktable.c:119-130 */
static double f_mismatch_uparse(double p, double q) {
	double v = 1 - (p + q / 3) / (p + q - 4 * p * q / 3);
	return (v <= 0) ? DBL_MIN : v;
}

void po_build_tables(po_tables *t) {
	int i, j;
	t->qual_nn = six_digits(log(0.25));	/* mktable.c:141 + tablebuilder.c:124 -> -1.38629 */
	for (i = 0; i < PO_NQ; i++) {
		double p = phred_p(i);
		/* mktable.c:63-73: Q=0 (p==1) is pinned to -2, not log(0); log_output=false (mktable.c:150) */
		t->score[i] = six_digits(p == 1 ? -2 : log(1.0 - p));
		t->score_err[i] = six_digits(log(p));	/* mktable.c:75-82 */
		for (j = 0; j < PO_NQ; j++) {
			double q = phred_p(j);
			/* tablebuilder.c:161-168: formula(P(x), P(y)) then log() */
			t->match_sb[i][j] = six_digits(log(f_match(p, q)));
			t->mismatch_sb[i][j] = six_digits(log(f_mismatch(p, q)));
			t->match_pear[i][j] = six_digits(log(f_match_pear(p, q)));
			t->mismatch_pear[i][j] = six_digits(log(f_mismatch_pear(p, q)));
			t->mismatch_rdp[i][j] = six_digits(log(f_mismatch(p, q)));
			t->mismatch_rdp_asm[i][j] = six_digits(log(f_mismatch_rdp_asm(p, q)));
			t->match_uparse[i][j] = six_digits(log(f_match_uparse(p, q)));
			t->mismatch_uparse[i][j] = six_digits(log(f_mismatch_uparse(p, q)));
		}
	}
}

/* ------------------------------------------------------------------ config */

void po_config_set_sb_q(po_config *c, double q) {	/* algo_simple_bayes.c:126-135 */
	if (q > 0 && q < 1) {
		c->sb_q = q;
		c->sb_pmatch = log(0.25 * (1 - 2 * q + q * q));
		c->sb_pmismatch = log((3 * q - 2 * q * q) / 18.0);
	}
}

void po_config_set_uparse_q(po_config *c, double q) {	/* algo_uparse.c:126-135 */
	if (q > 0 && q < 1) {
		c->up_q = q;
		c->up_pmatch = log(1 - q * q * (1 - 2 * q + 4 * q * q / 3));
		c->up_pmismatch = log(1 - 4 * q / 3 / (2 * q - 4 * q * q / 3));
	}
}

void po_config_default(po_config *c) {	/* assembler_support.c:61-97 */
	memset(c, 0, sizeof *c);
	c->algo = PO_ALGO_SIMPLE_BAYES;
	po_config_set_sb_q(c, 0.36);	/* algo_simple_bayes.c:113 */
	po_config_set_uparse_q(c, 0.36);	/* algo_uparse.c:103 */
	c->pear_random_base = log(0.25);	/* algo_pear.c:106 */
	c->threshold = log(0.6);
	c->minoverlap = 2;
	c->maxoverlap = 0;
	c->num_kmers = 2;
	c->primer_penalty = 0;
}

/* the two one-line pieces of the input domain, exported so that tests can hold them against the reference's own macros
 * (oracle/ref_macros.c: PHREDCLAMP prob.h:23, PANDA_NT_IS_DEGN pandaseq-nt.h:55) */
int po_phredclamp(int q) {
	return clampq(q);
}

int po_nt_is_degenerate(unsigned nt) {
	return is_degenerate(nt);
}

/* ------------------------------------------------------------------ k-mers */

static inline unsigned kcode(unsigned nt) {	/* misc.h:41: T=3 G=2 C=1, everything else 0 */
	return nt == 8 ? 3u : nt == 4 ? 2u : nt == 2 ? 1u : 0u;
}

/* misc.h:41-42: position p is emitted iff p >= 8 and no N among the 9 positions ending at p */
size_t po_kmers_forward(const po_qual *s, size_t len, uint16_t *kmer, uint16_t *posn) {
	size_t n = 0;
	unsigned k = 0;
	int bad = KMER_LEN;
	for (size_t p = 0; p < len; p++) {
		k = ((k << 2) | kcode(s[p].nt)) & 0xFFFFu;
		if (s[p].nt == NT_N) {
			bad = KMER_LEN;
		} else if (bad > 0) {
			bad--;
		} else {
			kmer[n] = (uint16_t) k;
			posn[n] = (uint16_t) p;
			n++;
		}
	}
	return n;
}

/* misc.h:43: same walk from len-1 down to 0 */
size_t po_kmers_reverse(const po_qual *s, size_t len, uint16_t *kmer, uint16_t *posn) {
	size_t n = 0;
	unsigned k = 0;
	int bad = KMER_LEN;
	for (ptrdiff_t p = (ptrdiff_t) len - 1; p >= 0; p--) {
		k = ((k << 2) | kcode(s[p].nt)) & 0xFFFFu;
		if (s[p].nt == NT_N) {
			bad = KMER_LEN;
		} else if (bad > 0) {
			bad--;
		} else {
			kmer[n] = (uint16_t) k;
			posn[n] = (uint16_t) p;
			n++;
		}
	}
	return n;
}

/* ------------------------------------------------------------------ scoring */

/* algo_simple_bayes.c:33-66 */
static double overlap_sb(const po_config *c, const po_tables *t, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t ov) {
	size_t matches = 0, mismatches = 0, unknowns = 0;
	for (size_t i = 0; i < ov; i++) {
		int fi = (int) (flen + i - ov);
		int ri = (int) (rlen - i - 1);
		if (fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		unsigned a = f[fi].nt, b = r[ri].nt;
		if (a == NT_N || b == NT_N)
			unknowns++;
		else if ((a & b) != 0)
			matches++;
		else
			mismatches++;
	}
	if (ov >= flen && ov >= rlen)
		return t->qual_nn * unknowns + matches * c->sb_pmatch + mismatches * c->sb_pmismatch;
	/* size_t arithmetic inside the parenthesis, then converted to double (algo_simple_bayes.c:64) */
	return t->qual_nn * (flen + rlen - 2 * ov + unknowns) + matches * c->sb_pmatch + mismatches * c->sb_pmismatch;
}

/* algo_pear.c:32-59.  The second table index is the FORWARD read's quality at rindex (sic, :52,:54).
 * When rlen > flen that index can run past the forward read: undefined behaviour in the reference;
 * this restatement (and the device path) define such an out-of-range quality as 0. */
static double overlap_pear(const po_config *c, const po_tables *t, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t ov) {
	double p = 0;
	for (size_t i = 0; i < ov; i++) {
		int fi = (int) (flen + i - ov);
		int ri = (int) (rlen - i - 1);
		if (fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		unsigned a = f[fi].nt, b = r[ri].nt;
		if (a == NT_N || b == NT_N) {
			p -= c->pear_random_base;
		} else {
			int qa = clampq(f[fi].qual);
			int qb = (size_t) ri < flen ? clampq(f[ri].qual) : 0;
			if ((a & b) != 0)
				p += t->match_pear[qa][qb];
			else
				p += t->mismatch_pear[qa][qb];
		}
	}
	return p;
}

/* algo_rdp_mle.c:43-74: no N case; (table - qual_nn) is formed first, then added */
static double overlap_rdp(const po_tables *t, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t ov) {
	double p = 0;
	for (size_t i = 0; i < ov; i++) {
		int fi = (int) (flen + i - ov);
		int ri = (int) (rlen - i - 1);
		/* the reference tests "> length" here (:58); findex < flen and rindex < rlen always hold, so it never matters */
		if (fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		int qa = clampq(f[fi].qual), qb = clampq(r[ri].qual);
		if ((f[fi].nt & r[ri].nt) != 0)
			p += t->match_sb[qa][qb] - t->qual_nn;
		else
			p += t->mismatch_rdp[qa][qb] - t->qual_nn;
	}
	return p;
}

/* The walk the four remaining scorers share (algo_uparse.c:44-59, algo_flash.c:44-58, algo_ea_util.c:43-54,
 * algo_stitch.c:39-54): bases that exist on both reads, split into unknown (an N on either side),
 * matching and mismatching. */
static void overlap_counts(const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t ov, size_t *unknowns, size_t *matches, size_t *mismatches) {
	*unknowns = *matches = *mismatches = 0;
	for (size_t i = 0; i < ov; i++) {
		int fi = (int) (flen + i - ov);
		int ri = (int) (rlen - i - 1);
		if (fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		unsigned a = f[fi].nt, b = r[ri].nt;
		if (a == NT_N || b == NT_N)
			(*unknowns)++;
		else if ((a & b) != 0)
			(*matches)++;
		else
			(*mismatches)++;
	}
}

double po_overlap_probability(const po_config *c, const po_tables *t, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t ov) {
	size_t u, m, mm;
	switch (c->algo) {
	case PO_ALGO_UPARSE:	/* algo_uparse.c:33-66: simple_bayesian's form with its own two constants */
		overlap_counts(f, flen, r, rlen, ov, &u, &m, &mm);
		if (ov >= flen && ov >= rlen)
			return t->qual_nn * u + m * c->up_pmatch + mm * c->up_pmismatch;
		return t->qual_nn * (flen + rlen - 2 * ov + u) + m * c->up_pmatch + mm * c->up_pmismatch;
	case PO_ALGO_FLASH: {	/* algo_flash.c:30-61: an N counts as a mismatch; the quotient is an INTEGER division (sic) */
		overlap_counts(f, flen, r, rlen, ov, &u, &m, &mm);
		size_t bad = u + mm, real = u + m + mm;
		return real == 0 ? -2 : log(bad / real);
	}
	case PO_ALGO_EA_UTIL: {	/* algo_ea_util.c:29-56 */
		overlap_counts(f, flen, r, rlen, ov, &u, &m, &mm);
		size_t bad = u + mm, real = u + m + mm;
		return log((((double) bad) * bad + 1) / real);
	}
	case PO_ALGO_STITCH: {	/* algo_stitch.c:28-57: the score is a size_t, so more mismatches than matches wraps */
		overlap_counts(f, flen, r, rlen, ov, &u, &m, &mm);
		size_t score = m - mm;
		return log(score / (double) (flen + rlen));
	}
	case PO_ALGO_PEAR:
		return overlap_pear(c, t, f, flen, r, rlen, ov);
	case PO_ALGO_RDP_MLE:
		return overlap_rdp(t, f, flen, r, rlen, ov);
	default:
		return overlap_sb(c, t, f, flen, r, rlen, ov);
	}
}

/* algo_simple_bayes.c:68-75, algo_pear.c:61-68, algo_rdp_mle.c:29-41 */
double po_match_probability(const po_config *c, const po_tables *t, int match, int8_t a, int8_t b) {
	switch (c->algo) {
	case PO_ALGO_UPARSE:	/* algo_uparse.c:68-75 */
		return (match ? t->match_uparse : t->mismatch_uparse)[clampq(a)][clampq(b)];
	case PO_ALGO_FLASH: {	/* algo_flash.c:63-80 */
		int score;
		if (match) {
			score = (a > b) ? clampq(a) : clampq(b);
		} else {
			score = clampq(a) - clampq(b);
			if (score < 0)
				score = -score;
			if (score < 2)
				score = 2;
		}
		return t->score[score];
	}
	case PO_ALGO_EA_UTIL:	/* algo_ea_util.c:58-67 */
		return t->score[(a > b) ? clampq(a) : clampq(b)];
	case PO_ALGO_STITCH:	/* algo_stitch.c:59-67: simple_bayesian's tables */
		return (match ? t->match_sb : t->mismatch_sb)[clampq(a)][clampq(b)];
	case PO_ALGO_PEAR:
		return (match ? t->match_pear : t->mismatch_pear)[clampq(a)][clampq(b)];
	case PO_ALGO_RDP_MLE:
		if (match) {
			int8_t mx = (a >= b) ? a : b;	/* raw max, then clamp */
			return t->score[clampq(mx)];
		}
		return t->mismatch_rdp_asm[clampq(a)][clampq(b)];
	default:
		return (match ? t->match_sb : t->mismatch_sb)[clampq(a)][clampq(b)];
	}
}

/* ------------------------------------------------------------------ primer offset (offset.c:47-112) */

size_t po_compute_offset_qual(const po_tables *t, double threshold, double penalty, const po_qual *seq, size_t len, const uint8_t *primer, size_t plen) {
	double pr[PO_MAX_LEN];
	double bestpr = exp(plen * threshold);
	size_t best = 0;
	if (plen > len || plen == 0)
		return 0;
	for (size_t i = 0; i < plen; i++)
		pr[i] = -INFINITY;
	for (size_t index = 0; index < len; index++) {
		double last = exp(pr[index % plen] / (index + 1)) - index * penalty;
		if (last > bestpr) {
			bestpr = last;
			best = index + 1;
		}
		pr[index % plen] = 0;
		for (ptrdiff_t x = (ptrdiff_t) (plen > index ? index : plen - 1); x >= 0; x--) {
			if (primer[x] != NT_N) {
				int q = clampq(seq[index].qual);
				pr[(index - (size_t) x) % plen] += ((seq[index].nt & primer[x]) != 0) ? t->score[q] : t->score_err[q];
			}
		}
	}
	return best;
}

/* panda_log1mexp (offset.c:35-38): log(1 - exp(p)) for a log-probability p.  Restated as written, including
 * its test `p > M_LN2`, which for p <= 0 always takes the second form. */
double po_log1mexp(double p) {
	return (p > 0.69314718055994530942) ? log1p(-exp(-p)) : log(-expm1(-p));
}

/* computeoffset() with result_base_score (offset.c:46-89,114-133): the same circular-bucket scan over the
 * merged bases; a base that agrees with the primer adds its log-probability, one that does not adds
 * log(1 - exp(p)).  reverse walks the sequence from its last base. */
size_t po_compute_offset_result(double threshold, double penalty, int reverse, const uint8_t *nt, const double *p, size_t len, const uint8_t *primer, size_t plen) {
	double pr[PO_MAX_LEN];
	double bestpr = exp(plen * threshold);
	size_t best = 0;
	if (plen > len || plen == 0)
		return 0;
	for (size_t i = 0; i < plen; i++)
		pr[i] = -INFINITY;
	for (size_t index = 0; index < len; index++) {
		double last = exp(pr[index % plen] / (index + 1)) - index * penalty;
		if (last > bestpr) {
			bestpr = last;
			best = index + 1;
		}
		pr[index % plen] = 0;
		const size_t at = reverse ? len - index - 1 : index;
		for (ptrdiff_t x = (ptrdiff_t) (plen > index ? index : plen - 1); x >= 0; x--) {
			if (primer[x] != NT_N)
				pr[(index - (size_t) x) % plen] += ((nt[at] & primer[x]) != 0) ? p[at] : po_log1mexp(p[at]);
		}
	}
	return best;
}

/* ------------------------------------------------------------------ phred of an output base (nt.c:126-150) */

int po_result_phred(const po_tables *t, double p) {
	int lower = 0, upper = PO_PHREDMAX;
	if (p <= t->score[0])
		return 1;
	while (lower < upper) {
		int mid = lower + (upper - lower) / 2;
		if (t->score[mid] == p)
			return mid;
		if (mid == lower)
			return lower;
		else if (t->score[mid] > p)
			upper = mid;
		else if (t->score[mid] < p)
			lower = mid + 1;
	}
	return lower;
}

/* ------------------------------------------------------------------ align (assembler.c:48-250) */

/* The reference's table is KMERSEEN_SIZE(num_kmers) = 65536 * num_kmers entries (assembler_support.c:82, misc.h:34).  The oracle keeps
 * room for PO_MAX_NUM_KMERS of them per k-mer ("no more than 10", pandaseq-assembler.h:58) and bounds every index by the size the
 * reference would have allocated. */
#define PO_MAX_NUM_KMERS 16
struct po_scratch {
	uint16_t *seen;		/* 65536 * PO_MAX_NUM_KMERS, zero when idle (assembler_support.c:82,94) */
};

po_scratch *po_scratch_new(void) {
	po_scratch *s = malloc(sizeof *s);
	s->seen = calloc(65536 * PO_MAX_NUM_KMERS, sizeof(uint16_t));
	return s;
}

void po_scratch_free(po_scratch *s) {
	if (s) {
		free(s->seen);
		free(s);
	}
}

#define VEEZ(x) ((x) < 0 ? 0 : (x))
#define WEDGEZ(x) ((x) > 0 ? 0 : (x))

static int align(const po_config *c, const po_tables *t, po_scratch *sc, const po_qual *F, size_t flen, const po_qual *R, size_t rlen, size_t fo, size_t ro, po_result *res, uint8_t *S_nt, double *S_p) {
	const size_t minov = (size_t) c->minoverlap;
	const double NN = t->qual_nn;
	uint16_t kf[PO_MAX_LEN], pf[PO_MAX_LEN], kr[PO_MAX_LEN], prv[PO_MAX_LEN];
	uint32_t bits[(2 * PO_MAX_LEN) / 32 + 2];

	/* assembler.c:59, unsigned arithmetic as in the reference */
	size_t maxov = flen + rlen - minov - fo - ro - 1;
	double best = NN * (flen + rlen);	/* :60 */
	ptrdiff_t bo = -1;

	res->slow = 0;
	if (minov + fo >= flen || minov + ro >= rlen)	/* :73 */
		return 0;
	if (c->maxoverlap == 0)	/* :78-82 */
		maxov = flen < rlen ? flen : rlen;
	else if (maxov > (size_t) c->maxoverlap)
		maxov = (size_t) c->maxoverlap;
	size_t size = minov <= maxov ? maxov - minov + 1 : 1;	/* :84 */
	size_t words = size / 32 + 1;
	memset(bits, 0, words * sizeof(uint32_t));

	/* :92-101 forward k-mers.  The free slot is LOOKED FOR at kmerseen[(kmer << 1) + j] (:94) and the position WRITTEN to
	 * kmerseen[kmer * num_kmers + j] (:99): the same entry only for num_kmers == 2.  For any other value the test reads entries
	 * that belong to other k-mers (k-mer k's own are [k * n, k * n + n)); the oracle does exactly that on one flat array.  With
	 * num_kmers == 1 the test's index reaches 2 * 65535 in an array of 65536 entries: the reference reads beyond its allocation
	 * there (undefined); defined here as "reads 0" (the slot is free).  num_kmers == 0: every k-mer is lost (j == num_kmers at once). */
	const size_t nk = c->num_kmers < 0 ? 0 : (c->num_kmers > PO_MAX_NUM_KMERS ? PO_MAX_NUM_KMERS : (size_t) c->num_kmers);
	const size_t table = nk << 16;	/* entries the reference allocates */
	size_t nf = po_kmers_forward(F, flen, kf, pf);
	for (size_t a = 0; a < nf; a++) {
		size_t j;
		for (j = 0; j < nk; j++) {
			const size_t probe = ((size_t) kf[a] << 1) + j;
			if (probe >= table || sc->seen[probe] == 0)
				break;
		}
		if (j < nk)
			sc->seen[(size_t) kf[a] * nk + j] = pf[a];
		/* else: lost k-mer (:95-97) */
	}
	/* :104-110 reverse k-mers flag candidate overlaps */
	size_t nr = po_kmers_reverse(R, rlen, kr, prv);
	for (size_t a = 0; a < nr; a++) {
		const uint16_t *slot = &sc->seen[(size_t) kr[a] * nk];
		for (size_t j = 0; j < nk && slot[j] != 0; j++) {
			int index = (int) (flen + rlen - prv[a] - slot[j] - minov - 1);
			if (index >= 0 && (size_t) index < size)
				bits[index / 32] |= 1u << (index % 32);
		}
	}
	/* :113-116 reset */
	for (size_t a = 0; a < nf; a++)
		for (size_t j = 0; j < nk; j++)
			sc->seen[(size_t) kf[a] * nk + j] = 0;
	/* :118 ALL_BITS_IF_NONE */
	{
		uint32_t any = 0;
		for (size_t w = 0; w < words; w++)
			any |= bits[w];
		if (any == 0)
			memset(bits, 0xFF, words * sizeof(uint32_t));
	}

	/* :120-133 scan in ascending overlap, strict > keeps the smallest overlap on ties */
	size_t examined = 0;
	for (size_t counter = 0; counter < size; counter++) {
		if (!(bits[counter / 32] & (1u << (counter % 32))))
			continue;
		size_t ov = counter + minov;
		double pr = po_overlap_probability(c, t, F, flen, R, rlen, ov);
		if (pr > best && ov >= minov) {
			best = pr;
			bo = (ptrdiff_t) ov;
		}
		examined++;
	}
	res->examined = (int32_t) examined;
	if (examined == maxov - minov + 1)	/* :135, unsigned wrap included */
		res->slow = 1;
	if (bo == -1)
		return 0;

	/* :146-155 */
	ptrdiff_t len = (ptrdiff_t) flen - (ptrdiff_t) fo - bo + (ptrdiff_t) rlen - (ptrdiff_t) ro + 1;
	if (len <= 0)
		return 0;
	if ((size_t) len > 2 * PO_MAX_LEN)
		return 0;
	res->seq_len = (int32_t) (len - 1);
	int degenerates = 0, mism = 0;
	double fq = 0, oq = 0, rq = 0;
	ptrdiff_t df = (ptrdiff_t) flen - (ptrdiff_t) fo - bo;
	ptrdiff_t dr = (ptrdiff_t) rlen - (ptrdiff_t) ro - bo;

	/* :162-173 forward-only part */
	for (size_t i = 0; i < (size_t) VEEZ(df); i++) {
		int fi = (int) (i + fo);
		double q = t->score[clampq(F[fi].qual)];
		S_nt[i] = F[fi].nt;
		S_p[i] = q;
		if (is_degenerate(F[fi].nt))
			degenerates++;
		fq += q;
	}
	/* :176-177 B-cliff: trailing run of raw quality 2 */
	size_t uf, ur;
	for (uf = flen; uf > 0 && F[uf - 1].qual == 2; uf--) ;
	for (ur = rlen; ur > 0 && R[ur - 1].qual == 2; ur--) ;

	/* :181-228 overlap part */
	for (size_t i = 0; i < (size_t) (bo + WEDGEZ(df) + WEDGEZ(dr)); i++) {
		int index = (int) (VEEZ(df) + (ptrdiff_t) i);
		int fi = (int) ((ptrdiff_t) fo + VEEZ(df) + (ptrdiff_t) i);
		int ri = (int) ((ptrdiff_t) rlen - (ptrdiff_t) i - 1 + WEDGEZ(df));
		if (index < 0 || fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		int ismatch = (R[ri].nt & F[fi].nt) != 0;
		double q;
		uint8_t nt;
		if (!ismatch)
			mism++;
		if ((size_t) fi >= uf && (size_t) ri >= ur)
			q = NN;
		else if ((size_t) fi >= uf)
			q = t->score[clampq(R[ri].qual)];
		else if ((size_t) ri >= ur)
			q = t->score[clampq(F[fi].qual)];
		else
			q = po_match_probability(c, t, ismatch, F[fi].qual, R[ri].qual);
		if (ismatch)
			nt = R[ri].nt & F[fi].nt;
		else
			nt = (F[fi].qual < R[ri].qual) ? R[ri].nt : F[fi].nt;
		S_nt[index] = nt;
		S_p[index] = q;
		if (is_degenerate(nt))
			degenerates++;
		oq += q;
	}
	/* :231-243 reverse-only part */
	for (size_t i = 0; i < (size_t) VEEZ(dr); i++) {
		int index = (int) (df + bo + (ptrdiff_t) i);
		int ri = (int) ((ptrdiff_t) rlen - bo - (ptrdiff_t) i - 1);
		double q = t->score[clampq(R[ri].qual)];
		rq += q;
		S_nt[index] = R[ri].nt;
		S_p[index] = q;
		if (is_degenerate(R[ri].nt))
			degenerates++;
	}
	res->quality = (fq + rq + oq) / len;	/* :244: divided by len = sequence_length + 1 */
	res->overlap = (int32_t) bo;
	res->est_prob = best;
	res->mismatches = mism;
	res->degenerates = degenerates;
	return 1;
}

/* ------------------------------------------------------------------ assemble_seq (assembler.c:252-348) */

static int run_checks(const po_config *c, const po_tables *t, const po_result *res, const double *S_p) {
	/* module_checkseq, module.c:124-136: in order, first failure wins */
	for (int m = 0; m < c->n_modules; m++) {
		const po_module *mod = &c->modules[m];
		if (mod->kind == PO_MOD_MIN_PHRED) {	/* plugin_min_phred.c:8-22 */
			for (int j = 0; j < res->seq_len; j++)
				if (po_result_phred(t, S_p[j]) < mod->phred)
					return m;
		} else if (mod->kind == PO_MOD_MIN_OVERLAPBITS) {	/* plugin_min_overlapbits.c:17-23 */
			if (!(mod->nats <= res->est_prob))
				return m;
		}
	}
	return -1;
}

int po_assemble(const po_config *c, const po_tables *t, po_scratch *sc, const po_qual *F, size_t flen, const po_qual *R, size_t rlen, po_result *res, uint8_t *S_nt, double *S_p, po_counters *ctr) {
	po_counters dummy;
	if (ctr == NULL) {
		memset(&dummy, 0, sizeof dummy);
		ctr = &dummy;
	}
	memset(res, 0, sizeof *res);
	res->reject_module = -1;
	ctr->count++;
	if (flen < 2 || rlen < 2) {	/* :255-258 */
		ctr->badr++;
		return res->status = PO_BADR;
	}
	/* :259 precheck modules: neither in-scope module has one */
	size_t fo, ro;
	if (c->post_primers) {	/* :285-288 */
		fo = ro = 0;
	} else if (c->fprimer_len > 0) {	/* :263-273 */
		fo = po_compute_offset_qual(t, c->threshold, c->primer_penalty, F, flen, c->fprimer, (size_t) c->fprimer_len);
		if (fo == 0) {
			ctr->nofp++;
			return res->status = PO_NOFP;
		}
		fo--;
	} else {
		fo = (size_t) c->ftrim;
	}
	res->fwd_offset = (int32_t) fo;
	if (c->post_primers) {
	} else if (c->rprimer_len > 0) {	/* :274-284 */
		ro = po_compute_offset_qual(t, c->threshold, c->primer_penalty, R, rlen, c->rprimer, (size_t) c->rprimer_len);
		if (ro == 0) {
			ctr->norp++;
			return res->status = PO_NORP;
		}
		ro--;
	} else {
		ro = (size_t) c->rtrim;
	}
	res->rev_offset = (int32_t) ro;
	if ((flen < rlen ? flen : rlen) < (size_t) c->minoverlap) {	/* :289-292 */
		ctr->badr++;
		return res->status = PO_BADR;
	}
	int ok = align(c, t, sc, F, flen, R, rlen, fo, ro, res, S_nt, S_p);
	if (res->slow)
		ctr->slow++;	/* :135-137 sits inside align, before its failure exits */
	if (!ok) {	/* :293-299 */
		ctr->noalgn++;
		res->seq_len = 0;
		return res->status = PO_NOALGN;
	}
	if (c->post_primers) {	/* :300-333: the primers are looked for in the merged sequence and cut off it */
		const size_t len = (size_t) res->seq_len;
		if (c->fprimer_len > 0) {
			fo = po_compute_offset_result(c->threshold, c->primer_penalty, 0, S_nt, S_p, len, c->fprimer, (size_t) c->fprimer_len);
			if (fo == 0) {
				ctr->nofp++;
				res->seq_len = 0;
				return res->status = PO_NOFP;
			}
			fo--;
		} else {
			fo = (size_t) c->ftrim;
		}
		res->fwd_offset = (int32_t) fo;
		if (c->rprimer_len > 0) {
			ro = po_compute_offset_result(c->threshold, c->primer_penalty, 1, S_nt, S_p, len, c->rprimer, (size_t) c->rprimer_len);
			if (ro == 0) {
				ctr->norp++;
				res->seq_len = 0;
				return res->status = PO_NORP;
			}
			ro--;
		} else {
			ro = (size_t) c->rtrim;
		}
		res->rev_offset = (int32_t) ro;
		if (len <= fo + ro) {	/* :323-327: counted as a missing forward primer */
			ctr->nofp++;
			res->seq_len = 0;
			return res->status = PO_NOFP;
		}
		res->seq_len = (int32_t) (len - fo - ro);
		memmove(S_nt, S_nt + fo, (size_t) res->seq_len);
		memmove(S_p, S_p + fo, (size_t) res->seq_len * sizeof(double));
	}
	if (res->quality < c->threshold) {	/* :334-338 */
		ctr->lowq++;
		return res->status = PO_LOWQ;
	}
	int rej = run_checks(c, t, res, S_p);	/* :339 */
	if (rej >= 0) {
		ctr->rejected[rej]++;
		res->reject_module = rej;
		return res->status = PO_REJECT;
	}
	ctr->ok++;	/* :340-345 */
	ctr->overlapcount[res->overlap]++;
	if (ctr->longest_overlap < res->overlap)
		ctr->longest_overlap = res->overlap;
	return res->status = PO_OK;
}

void po_assemble_batch(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint8_t *seq_nt, double *seq_p, size_t ostride, po_counters *ctr) {
	po_scratch *sc = po_scratch_new();
	uint8_t nt_tmp[2 * PO_MAX_LEN];
	double p_tmp[2 * PO_MAX_LEN];
	for (size_t i = 0; i < n; i++) {
		uint8_t *nt = seq_nt ? seq_nt + i * ostride : nt_tmp;
		double *p = seq_p ? seq_p + i * ostride : p_tmp;
		if (ostride < 2 * PO_MAX_LEN && (seq_nt || seq_p)) {
			/* rows may be narrower than the worst case: assemble into scratch, then copy what fits */
			po_assemble(c, t, sc, fwd + i * fstride, flen[i], rev + i * rstride, rlen[i], &res[i], nt_tmp, p_tmp, ctr);
			size_t m = (size_t) res[i].seq_len < ostride ? (size_t) res[i].seq_len : ostride;
			if (res[i].status == PO_NOALGN)
				m = 0;
			if (seq_nt)
				memcpy(nt, nt_tmp, m);
			if (seq_p)
				memcpy(p, p_tmp, m * sizeof(double));
		} else {
			po_assemble(c, t, sc, fwd + i * fstride, flen[i], rev + i * rstride, rlen[i], &res[i], nt, p, ctr);
		}
	}
	po_scratch_free(sc);
}

struct mt_job {
	const po_config *c;
	const po_tables *t;
	size_t lo, hi;
	const po_qual *fwd, *rev;
	size_t fstride, rstride, ostride;
	const uint16_t *flen, *rlen;
	po_result *res;
	uint8_t *seq_nt;
	double *seq_p;
	po_counters ctr;
};

static void *mt_worker(void *arg) {
	struct mt_job *j = arg;
	size_t n = j->hi - j->lo;
	memset(&j->ctr, 0, sizeof j->ctr);
	po_assemble_batch(j->c, j->t, n, j->fwd + j->lo * j->fstride, j->fstride, j->rev + j->lo * j->rstride, j->rstride, j->flen + j->lo, j->rlen + j->lo, j->res + j->lo, j->seq_nt ? j->seq_nt + j->lo * j->ostride : NULL, j->seq_p ? j->seq_p + j->lo * j->ostride : NULL, j->ostride, &j->ctr);
	return NULL;
}

void po_assemble_batch_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint8_t *seq_nt, double *seq_p, size_t ostride, po_counters *ctr, int threads) {
	if (threads < 1)
		threads = 1;
	struct mt_job *jobs = calloc((size_t) threads, sizeof *jobs);
	pthread_t *tid = calloc((size_t) threads, sizeof *tid);
	for (int k = 0; k < threads; k++) {
		struct mt_job *j = &jobs[k];
		j->c = c;
		j->t = t;
		j->lo = n * (size_t) k / (size_t) threads;
		j->hi = n * (size_t) (k + 1) / (size_t) threads;
		j->fwd = fwd;
		j->rev = rev;
		j->fstride = fstride;
		j->rstride = rstride;
		j->ostride = ostride;
		j->flen = flen;
		j->rlen = rlen;
		j->res = res;
		j->seq_nt = seq_nt;
		j->seq_p = seq_p;
		pthread_create(&tid[k], NULL, mt_worker, &jobs[k]);
	}
	for (int k = 0; k < threads; k++) {
		pthread_join(tid[k], NULL);
		const po_counters *w = &jobs[k].ctr;
		ctr->count += w->count;
		ctr->badr += w->badr;
		ctr->nofp += w->nofp;
		ctr->norp += w->norp;
		ctr->noalgn += w->noalgn;
		ctr->lowq += w->lowq;
		ctr->ok += w->ok;
		ctr->slow += w->slow;
		for (int m = 0; m < PO_MAX_MODULES; m++)
			ctr->rejected[m] += w->rejected[m];
		for (int o = 0; o < 2 * PO_MAX_LEN; o++)
			ctr->overlapcount[o] += w->overlapcount[o];
		if (w->longest_overlap > ctr->longest_overlap)
			ctr->longest_overlap = w->longest_overlap;	/* a max, not a sum (assembler.c:342-344) */
	}
	free(jobs);
	free(tid);
}

/* ---- cpu_baseline of bench.py: the same per-thread loop as pool.c:140-167 runs (one assembler per thread over its own
 * share of the input), timed without the effects a Python caller adds: every thread copies its slice into memory it
 * allocates and first touches itself (so the pages are local to its NUMA node), is pinned to one CPU of the process's
 * affinity mask, and the clock runs from the moment all threads are ready to the moment the last one is done.
 * Returns the elapsed seconds (< 0: could not start); counters are added up as above. */
#include <sched.h>
#include <time.h>

struct bench_job {
	struct mt_job j;
	int cpu;	/* -1: not pinned */
	pthread_barrier_t *ready, *go;
	double seconds;
	int failed;
};

static double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *bench_worker(void *arg) {
	struct bench_job *b = arg;
	struct mt_job *j = &b->j;
	const size_t n = j->hi - j->lo;
	if (b->cpu >= 0) {
		cpu_set_t set;
		CPU_ZERO(&set);
		CPU_SET(b->cpu, &set);
		(void) pthread_setaffinity_np(pthread_self(), sizeof set, &set);
	}
	po_qual *f = malloc((n ? n : 1) * j->fstride * sizeof(po_qual)), *r = malloc((n ? n : 1) * j->rstride * sizeof(po_qual));
	uint16_t *fl = malloc((n ? n : 1) * sizeof(uint16_t)), *rl = malloc((n ? n : 1) * sizeof(uint16_t));
	po_result *res = malloc((n ? n : 1) * sizeof(po_result));
	b->failed = !f || !r || !fl || !rl || !res;
	if (!b->failed) {
		memcpy(f, j->fwd + j->lo * j->fstride, n * j->fstride * sizeof(po_qual));
		memcpy(r, j->rev + j->lo * j->rstride, n * j->rstride * sizeof(po_qual));
		memcpy(fl, j->flen + j->lo, n * sizeof(uint16_t));
		memcpy(rl, j->rlen + j->lo, n * sizeof(uint16_t));
		memset(res, 0, n * sizeof(po_result));
	}
	memset(&j->ctr, 0, sizeof j->ctr);
	pthread_barrier_wait(b->ready);
	pthread_barrier_wait(b->go);
	const double t0 = now_s();
	if (!b->failed)
		po_assemble_batch(j->c, j->t, n, f, j->fstride, r, j->rstride, fl, rl, res, NULL, NULL, 2 * PO_MAX_LEN, &j->ctr);
	b->seconds = now_s() - t0;
	if (!b->failed && j->res)
		memcpy(j->res + j->lo, res, n * sizeof(po_result));
	free(f);
	free(r);
	free(fl);
	free(rl);
	free(res);
	return NULL;
}

/* ---- per-row checksums of a batch (bench.py's at-scale parity check; tests): the batch is assembled in blocks of 256 pairs with
 * their merged rows, and of every accepted pair's row two position-weighted sums are kept, both in wrapping 64-bit integer
 * arithmetic -- so that they are exact whatever order a checker on another machine adds in --:
 *   cs_nt[i] = sum (j+1) * nt_j        cs_p[i] = sum (j+1) * (the 64 bits of p_j)        (0, 0 for a pair without a row) */
struct cs_job {
	struct mt_job j;
	uint64_t *cs_nt, *cs_p;
};

static void *cs_worker(void *arg) {
	struct cs_job *b = arg;
	struct mt_job *j = &b->j;
	enum { BLOCK = 256 };
	const size_t ostride = 2 * PO_MAX_LEN;
	uint8_t *nt = malloc((size_t) BLOCK * ostride);
	double *p = malloc((size_t) BLOCK * ostride * sizeof(double));
	memset(&j->ctr, 0, sizeof j->ctr);
	if (!nt || !p)
		abort();
	for (size_t lo = j->lo; lo < j->hi; lo += BLOCK) {
		const size_t n = j->hi - lo < BLOCK ? j->hi - lo : BLOCK;
		po_counters ctr;
		memset(&ctr, 0, sizeof ctr);
		po_assemble_batch(j->c, j->t, n, j->fwd + lo * j->fstride, j->fstride, j->rev + lo * j->rstride, j->rstride, j->flen + lo, j->rlen + lo, j->res + lo, nt, p, ostride, &ctr);
		for (size_t i = 0; i < n; i++) {
			uint64_t a = 0, q = 0;
			if (j->res[lo + i].status == 0) {
				const size_t m = j->res[lo + i].seq_len;
				for (size_t k = 0; k < m; k++) {
					uint64_t bits;
					memcpy(&bits, &p[i * ostride + k], 8);
					a += (uint64_t) (k + 1) * nt[i * ostride + k];
					q += (uint64_t) (k + 1) * bits;
				}
			}
			b->cs_nt[lo + i] = a;
			b->cs_p[lo + i] = q;
		}
	}
	free(nt);
	free(p);
	return NULL;
}

void po_row_checksums_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint64_t *cs_nt, uint64_t *cs_p, int threads) {
	if (threads < 1)
		threads = 1;
	struct cs_job *jobs = calloc((size_t) threads, sizeof *jobs);
	pthread_t *tid = calloc((size_t) threads, sizeof *tid);
	if (!jobs || !tid)
		abort();
	for (int k = 0; k < threads; k++) {
		struct mt_job *j = &jobs[k].j;
		j->c = c;
		j->t = t;
		j->lo = n * (size_t) k / (size_t) threads;
		j->hi = n * (size_t) (k + 1) / (size_t) threads;
		j->fwd = fwd;
		j->rev = rev;
		j->fstride = fstride;
		j->rstride = rstride;
		j->flen = flen;
		j->rlen = rlen;
		j->res = res;
		jobs[k].cs_nt = cs_nt;
		jobs[k].cs_p = cs_p;
		if (pthread_create(&tid[k], NULL, cs_worker, &jobs[k]) != 0)
			abort();
	}
	for (int k = 0; k < threads; k++)
		pthread_join(tid[k], NULL);
	free(jobs);
	free(tid);
}

double po_bench_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, po_counters *ctr, int threads, int pin) {
	if (threads < 1)
		threads = 1;
	cpu_set_t allowed;
	int cpus[4096], ncpus = 0;
	if (pin && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
		for (int k = 0; k < CPU_SETSIZE && ncpus < 4096; k++)
			if (CPU_ISSET(k, &allowed))
				cpus[ncpus++] = k;
	struct bench_job *jobs = calloc((size_t) threads, sizeof *jobs);
	pthread_t *tid = calloc((size_t) threads, sizeof *tid);
	pthread_barrier_t ready, go;
	if (!jobs || !tid || pthread_barrier_init(&ready, NULL, (unsigned) threads + 1) || pthread_barrier_init(&go, NULL, (unsigned) threads + 1)) {
		free(jobs);
		free(tid);
		return -1.0;
	}
	int started = 0;
	for (int k = 0; k < threads; k++) {
		struct mt_job *j = &jobs[k].j;
		j->c = c;
		j->t = t;
		j->lo = n * (size_t) k / (size_t) threads;
		j->hi = n * (size_t) (k + 1) / (size_t) threads;
		j->fwd = fwd;
		j->rev = rev;
		j->fstride = fstride;
		j->rstride = rstride;
		j->ostride = 2 * PO_MAX_LEN;
		j->flen = flen;
		j->rlen = rlen;
		j->res = res;
		/* spread over the allowed CPUs: with fewer threads than CPUs every thread gets a CPU (and, on most hosts, a core) of its own */
		jobs[k].cpu = ncpus > 0 ? cpus[(size_t) k * (size_t) ncpus / (size_t) threads] : -1;
		jobs[k].ready = &ready;
		jobs[k].go = &go;
		if (pthread_create(&tid[k], NULL, bench_worker, &jobs[k]) != 0)
			break;
		started++;
	}
	if (started != threads) {	/* cannot release a barrier that not everybody reaches: give up on the process' behalf */
		fprintf(stderr, "po_bench_mt: only %d of %d threads started\n", started, threads);
		abort();
	}
	pthread_barrier_wait(&ready);
	const double t0 = now_s();
	pthread_barrier_wait(&go);
	int failed = 0;
	for (int k = 0; k < threads; k++) {
		pthread_join(tid[k], NULL);
		failed |= jobs[k].failed;
		const po_counters *w = &jobs[k].j.ctr;
		ctr->count += w->count;
		ctr->badr += w->badr;
		ctr->nofp += w->nofp;
		ctr->norp += w->norp;
		ctr->noalgn += w->noalgn;
		ctr->lowq += w->lowq;
		ctr->ok += w->ok;
		ctr->slow += w->slow;
	}
	const double elapsed = now_s() - t0;
	pthread_barrier_destroy(&ready);
	pthread_barrier_destroy(&go);
	free(jobs);
	free(tid);
	return failed ? -1.0 : elapsed;
}
