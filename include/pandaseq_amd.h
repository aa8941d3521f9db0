/*
 * pandaseq_amd.h -- C ABI of the MI355X (gfx950) overlap-assembly hot path.
 *
 * This is the drop-in boundary for PANDAseq's per-pair assembly path
 * (panda_assembler_assemble(), assembler.c:368-383, and the loop body of
 * panda_assembler_next(), assembler.c:350-366): everything assemble_seq()
 * (assembler.c:252-348) does for one read pair is done here for a whole batch
 * of pairs by one HIP kernel launch, one wavefront per pair.
 *
 * Plain pointers and sizes only: no torch, no C++ types.  `stream` arguments
 * are a hipStream_t passed as void* (NULL = the default stream).
 *
 * Two layers:
 *   1. pamd_*            batch API over device (HBM) or host buffers -- the hot path.
 *   2. panda_assembler_* the reference's own per-assembler API (pandaseq-assembler.h:40-405)
 *                        re-implemented on top of layer 1; declared in pandaseq_amd_compat.h.
 *
 * Every entry point returns 0 on success or a negative PAMD_E* code; the
 * message is available from pamd_last_error().  There is NO CPU fallback: with
 * no usable GPU every compute entry point fails with PAMD_ENODEVICE.
 */
#ifndef PANDASEQ_AMD_H
#define PANDASEQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAMD_MAX_LEN 450	/* PANDA_MAX_LEN / configure.ac:8 */
#define PAMD_MAX_NUM_KMERS 64	/* panda_assembler_new_kmer: "no more than 10" (pandaseq-assembler.h:58); 2 (the default) runs on the device, others on the host path */
#define PAMD_MAX_MODULES 8
#define PAMD_NUM_COUNTERS (8 + PAMD_MAX_MODULES + 2 * PAMD_MAX_LEN)

enum {
	PAMD_OK = 0,
	PAMD_EINVAL = -1,	/* bad argument / value outside the device domain */
	PAMD_ENODEVICE = -2,	/* no HIP device, or HIP runtime error */
	PAMD_ENOMEM = -3,
	PAMD_EUNSUPPORTED = -4,	/* e.g. an algorithm class without a device form */
	PAMD_EIO = -5		/* a file could not be read or written (pamd_pipe_run: pread / pwrite / inflate failed) */
};

/* One base.  Byte-identical to panda_qual (pandaseq-common.h:210-219): nt is the one-hot
 * IUPAC code (A=1 C=2 G=4 T=8, N=15; pandaseq-nt.h:35-59), qual the raw PHRED score.
 * The reverse read is complemented but NOT reversed, exactly as fastq.c:154 produces it. */
typedef struct {
	uint8_t nt;
	int8_t qual;
} pamd_qual;

/* replaces the `.name = "simple_bayesian" / "pear" / "rdp_mle"` classes
 * (algo_simple_bayes.c:100-108, algo_pear.c:93-101, algo_rdp_mle.c:84-92) and the four remaining scorers
 * "uparse" / "flash" / "ea_util" / "stitch" (algo_uparse.c:94-102, algo_flash.c:91-99, algo_ea_util.c:79-87,
 * algo_stitch.c:79-87), all of which score an overlap from three integer counts */
enum { PAMD_ALGO_SIMPLE_BAYES = 0, PAMD_ALGO_PEAR = 1, PAMD_ALGO_RDP_MLE = 2,
       PAMD_ALGO_UPARSE = 3, PAMD_ALGO_FLASH = 4, PAMD_ALGO_EA_UTIL = 5, PAMD_ALGO_STITCH = 6 };

/* per-pair outcome; one value per rejection counter of assembler.h:59-67 */
enum {
	PAMD_ST_OK = 0,
	PAMD_ST_BADR = 1,	/* badreadcount  (assembler.c:255-258, :289-292) */
	PAMD_ST_NOFP = 2,	/* nofpcount     (assembler.c:265-268) */
	PAMD_ST_NORP = 3,	/* norpcount     (assembler.c:276-279) */
	PAMD_ST_NOALGN = 4,	/* noalgncount   (assembler.c:293-299) */
	PAMD_ST_LOWQ = 5,	/* lowqcount     (assembler.c:334-338) */
	PAMD_ST_REJECT = 6,	/* a check module rejected it (module.c:124-136) */
	PAMD_ST_NOSPACE = 7	/* not a reference outcome: the pair assembled but the output arena (seq_capacity) had no room
				 * for its row; it is counted in `count` only, and pamd_assemble_status() reports the launch */
};

/* What panda_result_seq (pandaseq-common.h:277-330) carries per pair, minus the pointers. 48 bytes. */
typedef struct {
	double quality;			/* result.quality */
	double est_overlap_prob;	/* result.estimated_overlap_probability (assembler.c:247; 0 when the pair did not merge: BADR .. NOALGN) */
	uint64_t seq_offset;		/* index of the first merged base in seq_nt / seq_p (valid when status == OK) */
	uint16_t seq_len;		/* result.sequence_length */
	uint16_t overlap;		/* result.overlap */
	uint16_t examined;		/* result.overlaps_examined */
	uint16_t mismatches;		/* result.overlap_mismatches */
	uint16_t degenerates;		/* result.degenerates */
	uint16_t fwd_offset;		/* result.forward_offset */
	uint16_t rev_offset;		/* result.reverse_offset */
	uint8_t status;			/* PAMD_ST_* */
	int8_t reject_module;		/* index of the rejecting module, or -1 */
	uint8_t slow;			/* this pair incremented slowcount (assembler.c:135-137) */
	uint8_t reserved[7];
} pamd_pair_result;

/* counter block, int64 each; the layout the STAT all-reduce sums */
enum {
	PAMD_CTR_COUNT = 0, PAMD_CTR_BADR, PAMD_CTR_NOFP, PAMD_CTR_NORP, PAMD_CTR_NOALGN,
	PAMD_CTR_LOWQ, PAMD_CTR_OK, PAMD_CTR_SLOW,
	PAMD_CTR_REJECTED0 = 8,					/* rejected[module], module.c:131 */
	PAMD_CTR_OVERLAP0 = 8 + PAMD_MAX_MODULES		/* overlapcount[overlap], assembler.c:341 */
};

/* A batch of read pairs and where its results go.  All pointers are DEVICE pointers for
 * pamd_assemble_device() and HOST pointers for pamd_assemble_host().
 *   fwd / rev     rows of pamd_qual, one row per pair, `*_stride` bases apart (multiple of 8,
 *                 >= the longest read in the batch, <= 456); row base 16-byte aligned
 *   flen / rlen   read lengths, each <= PAMD_MAX_LEN
 *   results       n_pairs records
 *   seq_nt/seq_p  merged bases (one-hot nt) and per-base log-probabilities of accepted pairs:
 *                 panda_result.nt / panda_result.p (pandaseq-common.h:221-230), as two arrays
 *   seq_stride    > 0: pair i writes at i * seq_stride (rows must hold the longest merge);
 *                 = 0: compact arena -- rows are allocated from *seq_cursor (device word,
 *                      zero it before the call) in multiples of 16 bases, seq_capacity bases in
 *                      total; a pair that does not fit is reported as PAMD_EINVAL after the run
 *   counters      PAMD_NUM_COUNTERS int64, ADDED to (not overwritten); may be NULL */
typedef struct {
	size_t n_pairs;
	const pamd_qual *fwd;
	size_t fwd_stride;
	const pamd_qual *rev;
	size_t rev_stride;
	const uint16_t *flen;
	const uint16_t *rlen;
	pamd_pair_result *results;
	uint8_t *seq_nt;
	double *seq_p;
	size_t seq_stride;
	size_t seq_capacity;
	uint64_t *seq_cursor;
	int64_t *counters;
} pamd_batch;

/* The configured assembler: the device-side image of struct panda_assembler's
 * configuration fields (assembler.h:44-57,72-78) plus the probability tables. */
typedef struct pamd_assembler pamd_assembler;

const char *pamd_last_error(void);
int pamd_device_count(void);

/* replaces panda_assembler_new (assembler_support.c:27-100): same defaults
 * (threshold log 0.6, minoverlap 2, maxoverlap 0, simple_bayesian q=0.36, no primers) */
pamd_assembler *pamd_assembler_create(int device);
void pamd_assembler_destroy(pamd_assembler *a);

/* replaces panda_assembler_set_algorithm + panda_algorithm_*_new/_set_* (algo_simple_bayes.c:110-135,
 * algo_pear.c:103-116, algo_rdp_mle.c:94-98, algo_uparse.c:104-135).  param: simple_bayes / uparse ->
 * error estimation q in (0,1), pear -> log of the random-base probability; NaN = keep the class default.
 * flash / ea_util / stitch take no parameter.  For ea_util and stitch `est_overlap_prob` is a device log()
 * of a ratio of small integers (within 1 ulp of the host's); every decision is exact because distinct
 * ratios are far apart and equal ratios give equal values. */
int pamd_set_algorithm(pamd_assembler *a, int algo, double param);
/* the setters below mirror assembler_support.c:201-410, including "silently ignore bad values" */
int pamd_set_threshold(pamd_assembler *a, double threshold);		/* (0,1), stored as log */
int pamd_set_log_threshold(pamd_assembler *a, double log_threshold);	/* the stored logarithm itself (< 0): no log(exp()) round trip for a caller that keeps it as struct panda_assembler does */
int pamd_set_minimum_overlap(pamd_assembler *a, int overlap);		/* > 1 */
int pamd_set_maximum_overlap(pamd_assembler *a, int overlap);		/* 0 = min(read lengths) */
int pamd_set_forward_primer(pamd_assembler *a, const uint8_t *nt, size_t length);
int pamd_set_reverse_primer(pamd_assembler *a, const uint8_t *nt, size_t length);	/* already complemented, args_assembler.c:222 */
int pamd_set_forward_trim(pamd_assembler *a, size_t trim);
int pamd_set_reverse_trim(pamd_assembler *a, size_t trim);
int pamd_set_primer_penalty(pamd_assembler *a, double penalty);
/* `-a` (panda_assembler_set_primers_after, assembler_support.c:355-359): look for the primers in the merged
 * sequence and cut them off it (assembler.c:300-333, panda_compute_offset_result offset.c:114-133) instead
 * of in the reads before assembly */
int pamd_set_primers_after(pamd_assembler *a, int after);
/* the two in-scope check modules, fused into the kernel, in the order added
 * (replaces panda_assembler_add_module of plugin_min_phred.c / plugin_min_overlapbits.c) */
int pamd_add_module_min_phred(pamd_assembler *a, int phred);
int pamd_add_module_min_overlapbits(pamd_assembler *a, double bits);
int pamd_clear_modules(pamd_assembler *a);
/* How the ordered f64 sums of the path are produced.
 *   exact = 1 (default): `quality` (assembler.c:171,226,236,244) and the pear / rdp_mle overlap scores
 *           (algo_pear.c:42-56, algo_rdp_mle.c:55-72) are added in the reference's index order: every
 *           output, including these two doubles, is bit-identical to the reference's.
 *   exact = 0: they are added in any order (|error| < 1e-9, far inside north_star's 1e-6); every DECISION
 *           that consumes them (best overlap, LOWQ, min_overlapbits) is re-made in reference order
 *           whenever it is closer than 1e-9 / 1e-6 to its threshold, so statuses, overlaps, merged bases and
 *           per-base log-probabilities stay bit-identical; only `quality` and, for pear / rdp_mle,
 *           `est_overlap_prob` may differ in their last bits.  About 15 % faster, 40 % less HBM traffic. */
int pamd_set_exact_sums(pamd_assembler *a, int exact);
/* Two kernels serve a launch: a lean one for ordinary pairs (reads of at most 256 bases holding only
 * A C G T N with qualities 0..46, a handful of seeded overlaps) and the general one, which then assembles whatever the
 * lean kernel deferred -- and everything, when the configuration is outside the lean kernel's (a maximum overlap below the
 * minimum, reads longer than 320 bases, a primer penalty, primers of more than 24 bases: DESIGN.md section 3).  Results do not
 * depend on which kernel produced them.  enable = 0 sends every pair
 * to the general kernel (tests use it to compare the two); default 1. */
int pamd_set_fast_path(pamd_assembler *a, int enable);
/* how many pairs of the most recent launch the lean kernel handed to the general one (synchronises); < 0 if none ran */
long long pamd_last_deferred(pamd_assembler *a);
/* which instantiation of the lean kernel served the most recent launch (a probe for tests and profiles; DESIGN.md 3.1, docs/NOTEBOOK.md 3.1-3.1e):
 * 0 = none (the general kernel took the launch whole); else PAMD_LEAN_SHORT (reads of at most 256 bases), PAMD_LEAN_LONG (at
 * most 320), PAMD_LEAN_PRIMERS (the same with `-p` / `-q` primers or trims found inside it) or PAMD_LEAN_AFTER (`-a`), plus
 * PAMD_LEAN_NARROW when reads of at most 192 bases were seeded three bases per lane, | PAMD_LEAN_DUAL when reads of at most 160
 * bases were assembled two pairs per wavefront (a counting scorer, any-order sums, no min_phred module, no maximum overlap) */
enum { PAMD_LEAN_SHORT = 1, PAMD_LEAN_LONG = 2, PAMD_LEAN_PRIMERS = 3, PAMD_LEAN_AFTER = 4, PAMD_LEAN_NARROW = 0x100, PAMD_LEAN_DUAL = 0x200 };
int pamd_last_lean_variant(pamd_assembler *a);
/* Two pairs per wavefront for reads of at most 160 bases (pamd_fast_dual.h).  OFF by default: the geometry is parity-green and
 * measured 22 % slower than a pair per wave on BASELINE's 2 x 150 configuration; results do not depend on
 * it.  Tests run both geometries on the same pairs through this switch. */
int pamd_set_two_pairs_per_wave(pamd_assembler *a, int enable);

/* The probability tables this library computes (host side; works without a GPU).
 * Order: qual_nn, qual_score[47], qual_score_err[47], then eight 47x47 row-major matrices:
 * match_simple_bayesian, mismatch_simple_bayesian, match_pear, mismatch_pear, mismatch_rdp_mle,
 * mismatch_assembled_rdp_mle, match_uparse, mismatch_uparse  (table.h as generated by mktable.c:132-155). */
#define PAMD_TABLE_DOUBLES (1 + 47 + 47 + 8 * 47 * 47)
int pamd_tables_get(double *out, size_t n_doubles);

/* Enqueue the assembly of one batch on `stream`; returns once launched (asynchronous). */
int pamd_assemble_device(pamd_assembler *a, const pamd_batch *batch, void *stream);
/* The same for pairs [first, first + count) of a batch that is resident as a whole: the streaming form of the reference's
 * unbounded loop (pool.c:71-108) when a shard's merged output does not fit beside its input.  Records go to
 * results[first ..]; with a compact arena the caller decides where this range's rows go by what seq_nt / seq_p /
 * seq_cursor / seq_capacity it passes (e.g. two arenas used in turn, each drained while the other fills); with strided
 * rows pair i still writes at i * seq_stride. */
int pamd_assemble_device_range(pamd_assembler *a, const pamd_batch *batch, size_t first, size_t count, void *stream);
/* Merged log-probabilities as table indices, for the trip to the host (2 bytes per base instead of 8): every log-p the
 * kernels write (panda_result.p, pandaseq-common.h:221-230) is an entry of the assembler's table image for the
 * current algorithm -- qual_score[], the scorer's match / mismatch matrices, qual_nn -- so seq_idx[j] = its index and
 * pamd_table_image()[seq_idx[j]] is seq_p[j], bit for bit.  n_bases values starting at seq_p (16-byte aligned; rounded up
 * to an even count); what lies between rows of a compact arena encodes as 0xFFFF.  Enqueued on `stream`. */
int pamd_encode_rows_device(pamd_assembler *a, const double *seq_p, size_t n_bases, uint16_t *seq_idx, void *stream);
/* the image itself (host side): *used doubles are written to out (capacity n_doubles) */
int pamd_table_image(pamd_assembler *a, double *out, size_t n_doubles, size_t *used);
/* After the launch has completed (synchronise the stream first): 0, or PAMD_EINVAL when pairs of the most recent
 * pamd_assemble_device() on this assembler did not fit in seq_capacity (their records say PAMD_ST_NOSPACE);
 * *n_overflowed (may be NULL) receives how many.  Resets the flag. */
int pamd_assemble_status(pamd_assembler *a, uint64_t *n_overflowed);
/* Same, from/to host memory: uploads, runs, downloads, synchronises. */
int pamd_assemble_host(pamd_assembler *a, const pamd_batch *batch);
/* HIP-event time in milliseconds of the most recent pamd_assemble_device() kernel on this
 * assembler (synchronises on it); < 0 if none. */
float pamd_last_kernel_ms(pamd_assembler *a);
/* Debug aid: per-phase shader-cycle totals accumulated by a -DPAMD_PHASE_TIMING build of the
 * library (all zero in the shipped build). out10: stage, offsets, seed-fwd, seed-rev, scan, merge,
 * sums, filters+write, (unused), whole-wave. */
int pamd_debug_phase_cycles(pamd_assembler *a, unsigned long long *out10, int reset);
/* persistent-launch geometry actually used by the last launch */
int pamd_last_launch_info(pamd_assembler *a, int *workgroups, int *waves_per_workgroup, int *lds_bytes);

/* Deterministic synthetic pairs [first, first+n) written straight into HBM (pandaseq_amd/synth.py
 * is the host twin).  thresholds: 64 uint32 device-side substitution thresholds. */
typedef struct {
	uint64_t seed;
	uint32_t read_len, tmin, tmax;
	int32_t qhi;
	int32_t novaseq;
	uint32_t fprimer_len, rprimer_len;
	uint8_t fprimer[32];	/* base indices 0..3 (ACGT) */
	uint8_t rprimer[32];
} pamd_synth_spec;
int pamd_synth_generate(int device, const pamd_synth_spec *spec, const uint32_t *thresholds_host, uint64_t first, size_t n, pamd_qual *fwd, pamd_qual *rev, size_t stride, uint16_t *flen, uint16_t *rlen, void *stream);

#ifdef __cplusplus
}
#endif
#endif
