/*
 * bench_next.c -- how fast the reference's own loop runs against libpandaseq_amd.so:
 *
 *     while ((result = panda_assembler_next(a)) != NULL) consume(result);        (pool.c:76)
 *
 * over an in-memory source (the reads of <pairs.bin>, cycled until <total> pairs have been handed out), everything
 * included: the source callback, the copies into pinned memory, PCIe both ways, the kernels, the expansion into
 * panda_result[] and a consumer that reads every merged base.  Plain C, written as an embedder writes it.
 *
 * usage: bench_next <pairs.bin> <total pairs> <algo> <exact sums 0|1> [batch pairs] [consume] [api] [chunk]
 *   api: 0 = the panda_assembler_next loop; 1 = panda_assembler_assemble per pair (diff.c:26-99); 2 = pamd_compat_assemble_many
 *   consume: 0 = the record's scalars only (length, overlap, quality); 1 = + every merged nt; 2 = + every log-p
 *            (default 2: what a FASTQ writer reads, output.c:100-126.  At 16 bytes per base, pandaseq-common.h:221-230,
 *            that is 6 KB of memory traffic per 2x250 pair on the consumer's one thread.)
 *   pairs.bin as for test_compat.c: u32 n, u32 stride, then n * (u16 flen, u16 rlen, stride*2 bytes fwd, stride*2 bytes rev)
 * prints one JSON object.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "pandaseq_amd_compat.h"

struct source {
	uint64_t total, pos;
	uint32_t n, stride;
	unsigned char *blob;
	size_t rec;
};

static bool next_pair(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	struct source *s = user_data;
	if (s->pos >= s->total)
		return false;
	const unsigned char *rec = s->blob + (size_t) (s->pos % s->n) * s->rec;
	uint16_t fl, rl;
	memcpy(&fl, rec, 2);
	memcpy(&rl, rec + 2, 2);
	memcpy(id->instrument, "SIM", 4);	/* a parsed Illumina header would fill the rest (seqid.c:142-285) */
	id->run[0] = id->flowcell[0] = id->tag[0] = '\0';
	id->lane = 1;
	id->tile = (int) (s->pos >> 20);
	id->x = (int) (s->pos & 0xFFFFF);
	id->y = 0;
	*forward = (const panda_qual *) (rec + 4);
	*forward_length = fl;
	*reverse = (const panda_qual *) (rec + 4 + (size_t) s->stride * 2);
	*reverse_length = rl;
	s->pos++;
	return true;
}

static double now(void) {
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

struct tally {
	uint64_t returned, bases, check, pbits;
	double psum;
	int consume;
};

static void tally_visit(size_t index, const panda_result_seq *res, void *user_data) {
	struct tally *t = user_data;
	(void) index;
	if (res == NULL)
		return;
	t->returned++;
	t->bases += res->sequence_length;
	t->psum += res->quality;
	t->check += res->overlap;
	if (t->consume >= 1)
		for (size_t j = 0; j < res->sequence_length; j++)
			t->check += (uint64_t) res->sequence[j].nt;
	if (t->consume >= 2)
		for (size_t j = 0; j < res->sequence_length; j++) {
			uint64_t b;
			memcpy(&b, &res->sequence[j].p, 8);
			t->pbits ^= b;
		}
}

int main(int argc, char **argv) {
	if (argc < 5) {
		fprintf(stderr, "usage: bench_next <pairs.bin> <total pairs> <algo> <exact 0|1> [batch pairs]\n");
		return 2;
	}
	FILE *in = fopen(argv[1], "rb");
	struct source src;
	memset(&src, 0, sizeof src);
	if (!in || fread(&src.n, 4, 1, in) != 1 || fread(&src.stride, 4, 1, in) != 1)
		return 3;
	src.total = strtoull(argv[2], NULL, 10);
	src.rec = 4 + (size_t) src.stride * 4;
	src.blob = malloc((size_t) src.n * src.rec);
	if (src.blob == NULL || fread(src.blob, src.rec, src.n, in) != src.n)
		return 3;
	fclose(in);
	pamd_compat_set_exact_sums(atoi(argv[4]));
	if (argc > 5)
		pamd_compat_set_batch_pairs((size_t) strtoull(argv[5], NULL, 10));
	const int consume = argc > 6 ? atoi(argv[6]) : 2;
	PandaAssembler a = panda_assembler_new(next_pair, &src, NULL, NULL);
	if (a == NULL)
		return 4;
	PandaAlgorithm algo = NULL;
	if (strcmp(argv[3], "pear") == 0)
		algo = panda_algorithm_pear_new();
	else if (strcmp(argv[3], "rdp_mle") == 0)
		algo = panda_algorithm_rdp_mle_new();
	if (algo != NULL) {
		panda_assembler_set_algorithm(a, algo);
		panda_algorithm_unref(algo);
	}
	const int api = argc > 7 ? atoi(argv[7]) : 0;	/* 0: panda_assembler_next; 1: panda_assembler_assemble, pair by pair; 2: pamd_compat_assemble_many */
	const double t0 = now();
	const panda_result_seq *res;
	uint64_t returned = 0, bases = 0, check = 0, pbits = 0;
	double psum = 0.0;
#define CONSUME(res)                                                               \
	do {                                                                       \
		returned++;                                                        \
		bases += (res)->sequence_length;                                   \
		psum += (res)->quality;                                            \
		check += (res)->overlap;                                           \
		if (consume >= 1)                                                  \
			for (size_t j = 0; j < (res)->sequence_length; j++)        \
				check += (uint64_t) (res)->sequence[j].nt;         \
		if (consume >= 2)                                                  \
			for (size_t j = 0; j < (res)->sequence_length; j++) {      \
				uint64_t b;                                        \
				memcpy(&b, &(res)->sequence[j].p, 8);              \
				pbits ^= b;                                        \
			}                                                          \
	} while (0)
	if (api == 0) {
		while ((res = panda_assembler_next(a)) != NULL)
			CONSUME(res);
	} else if (api == 1) {	/* what panda_diff does (diff.c:26-99): the caller holds the pair and asks for its assembly */
		panda_seq_identifier id;
		const panda_qual *f, *r;
		size_t fl, rl;
		while (next_pair(&id, &f, &fl, &r, &rl, &src))
			if ((res = panda_assembler_assemble(a, &id, f, fl, r, rl)) != NULL)
				CONSUME(res);
	} else {	/* the same pairs handed over as arrays, 65536 at a time (or [chunk] at a time) */
		const size_t CHUNK = argc > 8 ? (size_t) atol(argv[8]) : 65536;
		panda_seq_identifier *ids = malloc(CHUNK * sizeof *ids);
		const panda_qual **fs = malloc(CHUNK * sizeof *fs), **rs = malloc(CHUNK * sizeof *rs);
		size_t *fls = malloc(CHUNK * sizeof *fls), *rls = malloc(CHUNK * sizeof *rls);
		struct tally t = { 0, 0, 0, 0, 0.0, consume };
		for (;;) {
			size_t m = 0;
			while (m < CHUNK && next_pair(&ids[m], &fs[m], &fls[m], &rs[m], &rls[m], &src))
				m++;
			if (m == 0)
				break;
			pamd_compat_assemble_many(a, ids, fs, fls, rs, rls, m, tally_visit, &t);
		}
		returned = t.returned;
		bases = t.bases;
		check = t.check;
		pbits = t.pbits;
		psum = t.psum;
	}
	check ^= pbits;
	const double t1 = now();
	printf("{\"pairs\": %llu, \"returned\": %llu, \"merged_bases\": %llu, \"seconds\": %.6f, \"pairs_per_s\": %.1f, \"ok\": %ld, \"noalgn\": %ld, \"lowq\": %ld, \"check\": %llu, \"quality_sum\": %.6f, \"consume\": %d}\n",
	       (unsigned long long) src.total, (unsigned long long) returned, (unsigned long long) bases, t1 - t0, (double) src.total / (t1 - t0),
	       panda_assembler_get_ok_count(a), panda_assembler_get_failed_alignment_count(a), panda_assembler_get_low_quality_count(a), (unsigned long long) check, psum, consume);
	panda_assembler_unref(a);
	free(src.blob);
	return 0;
}
