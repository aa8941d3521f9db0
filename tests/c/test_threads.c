/*
 * test_threads.c -- the embedder that builds one assembler per thread (pool.c:140-160): T threads, each with its own
 * PandaAssembler over its own slice of <pairs.bin>, all on one GPU, running concurrently.
 * usage: test_threads <pairs.bin> <threads>     prints: count ok noalgn lowq badr slow  (summed over the threads)
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pandaseq_amd_compat.h"

struct source {
	uint32_t lo, hi, stride;
	unsigned char *blob;
	size_t rec;
	long sums[6];
	int device;
};

static bool next_pair(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	struct source *s = user_data;
	if (s->lo >= s->hi)
		return false;
	const unsigned char *rec = s->blob + (size_t) s->lo * s->rec;
	uint16_t fl, rl;
	memcpy(&fl, rec, 2);
	memcpy(&rl, rec + 2, 2);
	memset(id, 0, sizeof *id);
	id->x = (int) s->lo;
	*forward = (const panda_qual *) (rec + 4);
	*forward_length = fl;
	*reverse = (const panda_qual *) (rec + 4 + (size_t) s->stride * 2);
	*reverse_length = rl;
	s->lo++;
	return true;
}

static void *worker(void *arg) {
	struct source *s = arg;
	pamd_compat_set_thread_device(s->device);
	PandaAssembler a = panda_assembler_new(next_pair, s, NULL, NULL);
	if (a == NULL)
		return (void *) 1;
	while (panda_assembler_next(a) != NULL)
		;
	s->sums[0] = panda_assembler_get_count(a);
	s->sums[1] = panda_assembler_get_ok_count(a);
	s->sums[2] = panda_assembler_get_failed_alignment_count(a);
	s->sums[3] = panda_assembler_get_low_quality_count(a);
	s->sums[4] = panda_assembler_get_bad_read_count(a);
	s->sums[5] = panda_assembler_get_slow_count(a);
	panda_assembler_unref(a);
	return NULL;
}

int main(int argc, char **argv) {
	if (argc < 3)
		return 2;
	FILE *in = fopen(argv[1], "rb");
	uint32_t n, stride;
	if (!in || fread(&n, 4, 1, in) != 1 || fread(&stride, 4, 1, in) != 1)
		return 3;
	const size_t rec = 4 + (size_t) stride * 4;
	unsigned char *blob = malloc((size_t) n * rec);
	if (blob == NULL || fread(blob, rec, n, in) != n)
		return 3;
	fclose(in);
	const int threads = atoi(argv[2]);
	pamd_compat_set_batch_pairs(3000);
	pthread_t tid[16];
	struct source src[16];
	for (int t = 0; t < threads; t++) {
		memset(&src[t], 0, sizeof src[t]);
		src[t].lo = (uint32_t) ((uint64_t) n * t / threads);
		src[t].hi = (uint32_t) ((uint64_t) n * (t + 1) / threads);
		src[t].stride = stride;
		src[t].blob = blob;
		src[t].rec = rec;
		src[t].device = 0;
		pthread_create(&tid[t], NULL, worker, &src[t]);
	}
	long total[6] = { 0, 0, 0, 0, 0, 0 };
	for (int t = 0; t < threads; t++) {
		void *rv = NULL;
		pthread_join(tid[t], &rv);
		if (rv != NULL)
			return 4;
		for (int k = 0; k < 6; k++)
			total[k] += src[t].sums[k];
	}
	printf("%ld %ld %ld %ld %ld %ld\n", total[0], total[1], total[2], total[3], total[4], total[5]);
	free(blob);
	return 0;
}
