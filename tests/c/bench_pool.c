/*
 * bench_pool.c -- the reference's own deployment of the assembler: pool.c:140-167 starts T threads, each with an assembler of
 * its own (panda_mux child readers) running
 *
 *     while ((result = panda_assembler_next(assembler)) != NULL) output(result);          (pool.c:76-84)
 *
 * Here: T threads, each a PandaAssembler of libpandaseq_amd.so on the same GPU over its share of <total> pairs from an
 * in-memory source, each consuming every merged base (nt and log-p, what output.c:100-126 reads).  One consumer thread is
 * bound by the 6 KB of panda_result[] it reads per 2x250 pair (bench_next.c); this is the rate with the consumers the
 * reference itself would run.  Plain C.
 *
 * usage: bench_pool <pairs.bin> <total pairs> <algo> <exact sums 0|1> <threads>     prints one JSON object
 */
#define _POSIX_C_SOURCE 199309L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "pandaseq_amd_compat.h"

struct source {
	uint64_t total, pos;
	uint32_t n, stride;
	const unsigned char *blob;
	size_t rec;
	const char *algo;
	uint64_t returned, bases, check;
	long ok;
	int failed;
};

static bool next_pair(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	struct source *s = user_data;
	if (s->pos >= s->total)
		return false;
	const unsigned char *rec = s->blob + (size_t) (s->pos % s->n) * s->rec;
	uint16_t fl, rl;
	memcpy(&fl, rec, 2);
	memcpy(&rl, rec + 2, 2);
	memcpy(id->instrument, "SIM", 4);
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

static void *worker(void *arg) {
	struct source *s = arg;
	PandaAssembler a = panda_assembler_new(next_pair, s, NULL, NULL);
	if (a == NULL) {
		s->failed = 1;
		return NULL;
	}
	PandaAlgorithm algo = NULL;
	if (strcmp(s->algo, "pear") == 0)
		algo = panda_algorithm_pear_new();
	else if (strcmp(s->algo, "rdp_mle") == 0)
		algo = panda_algorithm_rdp_mle_new();
	if (algo != NULL) {
		panda_assembler_set_algorithm(a, algo);
		panda_algorithm_unref(algo);
	}
	const panda_result_seq *res;
	uint64_t pbits = 0;
	while ((res = panda_assembler_next(a)) != NULL) {
		s->returned++;
		s->bases += res->sequence_length;
		for (size_t j = 0; j < res->sequence_length; j++) {
			uint64_t b;
			memcpy(&b, &res->sequence[j].p, 8);
			s->check += (uint64_t) res->sequence[j].nt;
			pbits ^= b;
		}
	}
	s->check ^= pbits;
	s->ok = panda_assembler_get_ok_count(a);
	panda_assembler_unref(a);
	return NULL;
}

int main(int argc, char **argv) {
	if (argc < 6) {
		fprintf(stderr, "usage: bench_pool <pairs.bin> <total pairs> <algo> <exact 0|1> <threads>\n");
		return 2;
	}
	FILE *in = fopen(argv[1], "rb");
	uint32_t n = 0, stride = 0;
	if (!in || fread(&n, 4, 1, in) != 1 || fread(&stride, 4, 1, in) != 1)
		return 3;
	const size_t rec = 4 + (size_t) stride * 4;
	unsigned char *blob = malloc((size_t) n * rec);
	if (blob == NULL || fread(blob, rec, n, in) != n)
		return 3;
	fclose(in);
	const uint64_t total = strtoull(argv[2], NULL, 10);
	int threads = atoi(argv[5]);
	if (threads < 1)
		threads = 1;
	pamd_compat_set_exact_sums(atoi(argv[4]));
	struct source *src = calloc((size_t) threads, sizeof *src);
	pthread_t *tid = calloc((size_t) threads, sizeof *tid);
	const double t0 = now();
	for (int k = 0; k < threads; k++) {
		src[k].total = total * (uint64_t) (k + 1) / (uint64_t) threads - total * (uint64_t) k / (uint64_t) threads;
		src[k].n = n;
		src[k].stride = stride;
		src[k].blob = blob;
		src[k].rec = rec;
		src[k].algo = argv[3];
		pthread_create(&tid[k], NULL, worker, &src[k]);
	}
	uint64_t returned = 0, bases = 0, check = 0;
	long ok = 0;
	int failed = 0;
	for (int k = 0; k < threads; k++) {
		pthread_join(tid[k], NULL);
		returned += src[k].returned;
		bases += src[k].bases;
		check ^= src[k].check;
		ok += src[k].ok;
		failed |= src[k].failed;
	}
	const double t1 = now();
	if (failed)
		return 4;
	printf("{\"pairs\": %llu, \"returned\": %llu, \"merged_bases\": %llu, \"seconds\": %.6f, \"pairs_per_s\": %.1f, \"ok\": %ld, \"threads\": %d, \"check\": %llu}\n",
	       (unsigned long long) total, (unsigned long long) returned, (unsigned long long) bases, t1 - t0, (double) total / (t1 - t0), ok, threads, (unsigned long long) check);
	free(blob);
	return 0;
}
