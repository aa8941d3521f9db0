/*
 * panda_oracle.h -- CPU ORACLE for the PANDAseq overlap-assembly hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The HIP product path (pandaseq_amd/csrc) never links,
 * loads or calls it.
 *
 * What it is: a plain-C restatement (written from the behavioural spec in
 * SURVEY.md Appendix A and from reading the reference) of
 *   - the qual->log-probability tables   (mktable.c:23-155, tablebuilder.c:73-183, prob.h:21-23)
 *   - the 8-mer iterator                 (misc.h:36-45)
 *   - align(): seed, scan, argmax, merge (assembler.c:48-250)
 *   - the three scoring algorithms       (algo_simple_bayes.c:33-75, algo_pear.c:32-68, algo_rdp_mle.c:29-74)
 *   - the four remaining scorers         (algo_uparse.c:33-75, algo_flash.c:30-80, algo_ea_util.c:29-67, algo_stitch.c:28-67)
 *   - the primer offset scan             (offset.c:47-112), also over the merged sequence (`-a`, offset.c:35-38,114-133, assembler.c:300-333)
 *   - assemble_seq() control + counters  (assembler.c:252-348)
 *   - panda_result_phred + min_phred / min_overlapbits (nt.c:126-150, plugin_min_phred.c:8-22, plugin_min_overlapbits.c:17-23)
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   tables  : PINNED -- compared value-for-value against table.c emitted by the
 *             reference's own mktable.c+tablebuilder.c compiled here
 *             (oracle/Makefile target `ref`, output in oracle/_ref/).
 *   check modules (min_phred, min_overlapbits) and the result-record layout:
 *             PINNED -- the reference's own plugin_min_phred.c / plugin_min_overlapbits.c
 *             compile from their own files (oracle/Makefile `ref` -> _ref/*.so) and
 *             are run over the device path's results
 *             (tests/test_compat_gpu.py::test_reference_plugin_binaries).
 *   the rest: PARITY UNPINNED against a live reference build.  Every other
 *             reference .c includes the autoconf-generated config.h, and the
 *             assembler needs ltdl.h/bzlib.h, none of which exist in this
 *             image; writing stand-ins for them is not allowed, so the full
 *             reference is unbuildable here.  The only anchors are the
 *             known answers recorded in SURVEY.md from a real reference run
 *             (tests/test_oracle_cpu.py).
 */
#ifndef PANDA_ORACLE_H
#define PANDA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_MAX_LEN 450		/* configure.ac:8 default --with-max-len */
#define PO_PHREDMAX 46		/* prob.h:22 */
#define PO_NQ (PO_PHREDMAX + 1)

/* one base: same 2-byte layout as panda_qual (pandaseq-common.h:210-219) */
typedef struct {
	uint8_t nt;		/* one-hot A1 C2 G4 T8, IUPAC unions, N = 15 (pandaseq-nt.h:35-59) */
	int8_t qual;		/* PHRED score, raw (may lie outside 0..46) */
} po_qual;

enum { PO_ALGO_SIMPLE_BAYES = 0, PO_ALGO_PEAR = 1, PO_ALGO_RDP_MLE = 2,
       PO_ALGO_UPARSE = 3, PO_ALGO_FLASH = 4, PO_ALGO_EA_UTIL = 5, PO_ALGO_STITCH = 6 };	/* SURVEY.md 8f row 4 */

/* per-pair outcome, in the order assemble_seq() tests them (assembler.c:252-348) */
enum {
	PO_OK = 0,
	PO_BADR = 1,		/* badreadcount: a read shorter than 2, or min(len) < minoverlap */
	PO_NOFP = 2,		/* nofpcount */
	PO_NORP = 3,		/* norpcount */
	PO_NOALGN = 4,		/* noalgncount: align() returned false */
	PO_LOWQ = 5,		/* lowqcount */
	PO_REJECT = 6		/* a check module said no; which one is in reject_module */
};

enum { PO_MOD_MIN_PHRED = 1, PO_MOD_MIN_OVERLAPBITS = 2 };

typedef struct {
	int kind;		/* PO_MOD_* */
	int phred;		/* min_phred: the integer v */
	double nats;		/* min_overlapbits: bits * M_LN2 */
} po_module;

#define PO_MAX_MODULES 8

typedef struct {
	double qual_nn;				/* the literal -1.38629 (%g of log 0.25) */
	double score[PO_NQ];			/* qual_score */
	double score_err[PO_NQ];		/* qual_score_err */
	double match_sb[PO_NQ][PO_NQ];		/* qual_match_simple_bayesian */
	double mismatch_sb[PO_NQ][PO_NQ];	/* qual_mismatch_simple_bayesian */
	double match_pear[PO_NQ][PO_NQ];
	double mismatch_pear[PO_NQ][PO_NQ];
	double mismatch_rdp[PO_NQ][PO_NQ];	/* qual_mismatch_rdp_mle */
	double mismatch_rdp_asm[PO_NQ][PO_NQ];	/* qual_mismatch_assembled_rdp_mle */
	double match_uparse[PO_NQ][PO_NQ];	/* qual_match_uparse */
	double mismatch_uparse[PO_NQ][PO_NQ];	/* qual_mismatch_uparse: the one asymmetric table */
} po_tables;

typedef struct {
	int algo;
	double sb_q, sb_pmatch, sb_pmismatch;	/* algo_simple_bayes.c:126-135 */
	double pear_random_base;		/* algo_pear.c:106 */
	double threshold;			/* stored as log, assembler_support.c:76,391-397 */
	int minoverlap;				/* default 2 */
	int maxoverlap;				/* default 0 = min(flen, rlen) */
	int num_kmers;				/* default 2; any other value probes at (kmer << 1) + j and stores at kmer * n + j, as assembler.c:94 vs :99 do */
	uint8_t fprimer[PO_MAX_LEN];
	int fprimer_len;
	uint8_t rprimer[PO_MAX_LEN];
	int rprimer_len;
	int ftrim, rtrim;
	double primer_penalty;
	int n_modules;
	po_module modules[PO_MAX_MODULES];
	double up_q, up_pmatch, up_pmismatch;	/* algo_uparse.c:126-135 */
	int post_primers;			/* `-a`: look for the primers in the merged sequence (assembler.c:300-333) */
} po_config;

typedef struct {
	int32_t status;
	int32_t reject_module;
	int32_t overlap;
	int32_t examined;
	int32_t mismatches;
	int32_t degenerates;
	int32_t fwd_offset;
	int32_t rev_offset;
	int32_t seq_len;
	int32_t slow;		/* this pair bumped slowcount */
	double quality;
	double est_prob;
} po_result;

typedef struct {
	int64_t count, badr, nofp, norp, noalgn, lowq, ok, slow;
	int64_t rejected[PO_MAX_MODULES];
	int64_t overlapcount[2 * PO_MAX_LEN];
	int64_t longest_overlap;
} po_counters;

/* opaque scratch (the 65536 x num_kmers k-mer table of assembler_support.c:82) */
typedef struct po_scratch po_scratch;

void po_build_tables(po_tables *t);
void po_config_default(po_config *c);
void po_config_set_sb_q(po_config *c, double q);
void po_config_set_uparse_q(po_config *c, double q);
po_scratch *po_scratch_new(void);
void po_scratch_free(po_scratch *s);

/* the three callbacks, exposed for unit tests */
double po_overlap_probability(const po_config *c, const po_tables *t, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, size_t overlap);
double po_match_probability(const po_config *c, const po_tables *t, int match, int8_t a, int8_t b);
size_t po_compute_offset_qual(const po_tables *t, double threshold, double penalty, const po_qual *seq, size_t len, const uint8_t *primer, size_t plen);
int po_result_phred(const po_tables *t, double p);
double po_log1mexp(double p);	/* panda_log1mexp, offset.c:35-38 */
/* panda_compute_offset_result (offset.c:114-133): the primer scan over merged bases; reverse = from the 3' end */
size_t po_compute_offset_result(double threshold, double penalty, int reverse, const uint8_t *nt, const double *p, size_t len, const uint8_t *primer, size_t plen);
int po_phredclamp(int q);		/* PHREDCLAMP, prob.h:23 */
int po_nt_is_degenerate(unsigned nt);	/* PANDA_NT_IS_DEGN, pandaseq-nt.h:55 */
size_t po_kmers_forward(const po_qual *s, size_t len, uint16_t *kmer, uint16_t *posn);
size_t po_kmers_reverse(const po_qual *s, size_t len, uint16_t *kmer, uint16_t *posn);

/* one pair; seq_nt / seq_p need room for 2*PO_MAX_LEN entries; ctr may be NULL */
int po_assemble(const po_config *c, const po_tables *t, po_scratch *s, const po_qual *f, size_t flen, const po_qual *r, size_t rlen, po_result *res, uint8_t *seq_nt, double *seq_p, po_counters *ctr);

/* a batch laid out like the device batch: fixed-stride rows of po_qual, per-pair lengths,
 * fixed-stride output rows (ostride entries per pair).  seq_nt/seq_p may be NULL. */
void po_assemble_batch(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint8_t *seq_nt, double *seq_p, size_t ostride, po_counters *ctr);

/* N pthreads over contiguous slices of the batch (the shape of pool.c:140-167), counters summed at the end */
void po_assemble_batch_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint8_t *seq_nt, double *seq_p, size_t ostride, po_counters *ctr, int threads);

#ifdef __cplusplus
}
#endif
/* bench.py's cpu_baseline leg: `threads` pinned workers, each over a private, self-touched copy of its slice; returns the
 * seconds from "all ready" to "all done" (< 0 on failure).  res may be NULL. */
/* per-row checksums in wrapping 64-bit arithmetic: cs_nt[i] = sum (j+1) nt_j, cs_p[i] = sum (j+1) bits(p_j); 0 for a pair without a row */
void po_row_checksums_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, uint64_t *cs_nt, uint64_t *cs_p, int threads);
double po_bench_mt(const po_config *c, const po_tables *t, size_t n, const po_qual *fwd, size_t fstride, const po_qual *rev, size_t rstride, const uint16_t *flen, const uint16_t *rlen, po_result *res, po_counters *ctr, int threads, int pin);

#endif
