/* oracle/ref_macros.c -- the reference's OWN header-only pieces of the hot path, compiled where they lie.
 *
 * Test infrastructure (see panda_oracle.h): this file holds no logic of its own.  It #includes the reference's misc.h
 * (FOREACH_KMER / FOREACH_KMER_REVERSE, misc.h:36-45), prob.h (PHREDCLAMP, prob.h:23) and, through pandaseq.h,
 * pandaseq-nt.h (PANDA_NT_IS_N / PANDA_NT_IS_DEGN, pandaseq-nt.h:51-55) from $(REF) -- they need nothing generated, no
 * config.h, no library -- and wraps each macro in a function so that the oracle's restatement (po_kmers_forward /
 * po_kmers_reverse, clamps, degenerate count) and, through `overlaps_examined`, the device's candidate sets can be
 * checked against what the reference's preprocessor and compiler make of the reference's text.
 * Built by `make -C oracle ref` into _ref/librefmacros.so; never copied, never shipped in the product.
 */
#include <stddef.h>
#include <stdint.h>
#include "misc.h"
#include "prob.h"

/* FOREACH_KMER(it, seq, .nt): emits (KMER(it), KMER_POSITION(it)) in the order assembler.c:92 walks them */
size_t ref_kmers_forward(const panda_qual *seq, size_t seq_length, uint16_t *kmer, uint16_t *posn) {
	kmer_it it;
	size_t n = 0;
	FOREACH_KMER(it, seq,.nt) {
		kmer[n] = (uint16_t) KMER(it);
		posn[n] = (uint16_t) KMER_POSITION(it);
		n++;
	}
	return n;
}

/* FOREACH_KMER_REVERSE(it, seq, .nt): the order assembler.c:104 walks them */
size_t ref_kmers_reverse(const panda_qual *seq, size_t seq_length, uint16_t *kmer, uint16_t *posn) {
	kmer_it it;
	size_t n = 0;
	FOREACH_KMER_REVERSE(it, seq,.nt) {
		kmer[n] = (uint16_t) KMER(it);
		posn[n] = (uint16_t) KMER_POSITION(it);
		n++;
	}
	return n;
}

int ref_phredclamp(int q) {
	return PHREDCLAMP(q);
}

int ref_nt_is_n(int nt) {
	return PANDA_NT_IS_N((panda_nt) nt) ? 1 : 0;
}

int ref_nt_is_degn(int nt) {
	return PANDA_NT_IS_DEGN((panda_nt) nt) ? 1 : 0;
}

size_t ref_kmer_len(void) {
	return KMER_LEN;
}
