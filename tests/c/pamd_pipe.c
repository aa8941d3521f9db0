/*
 * pamd_pipe.c -- two FASTQ files in, one FASTA / FASTQ file out, through the C ABI only
 * (include/pandaseq_amd.h + include/pandaseq_amd_io.h + the HIP runtime for buffers and copies).
 *
 * This is the shape of the loop a PANDAseq maintainer writes around the batched reader and writers
 * (INTEGRATION.md section 5); it stands where main.c's reader -> pool.c:76 -> output.c chain stands, minus
 * option parsing, decompression and threads.  It is a test client (tests/test_pipe_gpu.py compares its
 * output file byte for byte with the oracle chain), not a command-line tool.
 *
 * usage: pamd_pipe <forward.fastq> <reverse.fastq> <out> <algo 0..6> <fasta|fastq> <pairs per chunk> [primers_after fwd_primer rev_primer]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pandaseq_amd.h"
#include "pandaseq_amd_io.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define CHECK_PAMD(x) do { int r_ = (x); if (r_ != PAMD_OK) { fprintf(stderr, "%s: %s\n", #x, pamd_last_error()); return 3; } } while (0)

static char *slurp(const char *path, size_t *n) {
	FILE *f = fopen(path, "rb");
	if (!f)
		return NULL;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = (char *) malloc((size_t) sz + 16);
	if (buf && fread(buf, 1, (size_t) sz, f) != (size_t) sz) {
		free(buf);
		buf = NULL;
	}
	fclose(f);
	*n = (size_t) sz;
	return buf;
}

/* panda_nt_from_ascii / _complement (nt.c:152-160) for plain ACGTN primers */
static size_t primer_codes(const char *s, int complement, uint8_t *out) {
	size_t n = 0;
	for (; *s; s++) {
		uint8_t v = *s == 'A' ? 1 : *s == 'C' ? 2 : *s == 'G' ? 4 : *s == 'T' ? 8 : 15;
		if (complement && v != 15)
			v = (uint8_t) (((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3));
		out[n++] = v;
	}
	return n;
}

int main(int argc, char **argv) {
	if (argc < 7) {
		fprintf(stderr, "usage: %s fwd.fastq rev.fastq out algo fasta|fastq pairs_per_chunk [primers_after fprimer rprimer]\n", argv[0]);
		return 1;
	}
	size_t fbytes, rbytes;
	char *ftext = slurp(argv[1], &fbytes), *rtext = slurp(argv[2], &rbytes);
	FILE *out = fopen(argv[3], "wb");
	if (!ftext || !rtext || !out) {
		fprintf(stderr, "cannot open the files\n");
		return 1;
	}
	const int algo = atoi(argv[4]);
	const int format = strcmp(argv[5], "fastq") == 0 ? PAMD_OUT_FASTQ : PAMD_OUT_FASTA;
	const size_t chunk_pairs = (size_t) atol(argv[6]);
	const size_t chunk_bytes = chunk_pairs * 1400 + 65536;	/* room for chunk_pairs records of up to ~1.4 KB */
	const size_t stride = 456;

	pamd_assembler *a = pamd_assembler_create(0);
	pamd_fastq_parser *fq = pamd_fastq_parser_create(0, 33, PAMD_TAG_OPTIONAL);
	pamd_formatter *fmt = pamd_formatter_create(0);
	if (!a || !fq || !fmt) {
		fprintf(stderr, "%s\n", pamd_last_error());
		return 3;
	}
	CHECK_PAMD(pamd_set_algorithm(a, algo, NAN));
	if (argc >= 10) {
		uint8_t codes[64];
		CHECK_PAMD(pamd_set_primers_after(a, atoi(argv[7])));
		CHECK_PAMD(pamd_set_forward_primer(a, codes, primer_codes(argv[8], 0, codes)));
		CHECK_PAMD(pamd_set_reverse_primer(a, codes, primer_codes(argv[9], 1, codes)));	/* complemented, args_assembler.c:222 */
	}

	char *d_f, *d_r, *d_text;
	pamd_qual *d_fwd, *d_rev;
	uint16_t *d_flen, *d_rlen;
	pamd_seqid *d_ids;
	pamd_pair_result *d_res;
	uint8_t *d_nt;
	double *d_p;
	uint64_t *d_cursor;
	int64_t *d_ctr;
	const size_t seq_cap = chunk_pairs * 912 + 16384 * 4096, text_cap = chunk_pairs * 2400 + 4096;
	CHECK_HIP(hipMalloc((void **) &d_f, chunk_bytes + 16));
	CHECK_HIP(hipMalloc((void **) &d_r, chunk_bytes + 16));
	CHECK_HIP(hipMalloc((void **) &d_text, text_cap));
	CHECK_HIP(hipMalloc((void **) &d_fwd, chunk_pairs * stride * 2));
	CHECK_HIP(hipMalloc((void **) &d_rev, chunk_pairs * stride * 2));
	CHECK_HIP(hipMalloc((void **) &d_flen, chunk_pairs * 2));
	CHECK_HIP(hipMalloc((void **) &d_rlen, chunk_pairs * 2));
	CHECK_HIP(hipMalloc((void **) &d_ids, chunk_pairs * sizeof(pamd_seqid)));
	CHECK_HIP(hipMalloc((void **) &d_res, chunk_pairs * sizeof(pamd_pair_result)));
	CHECK_HIP(hipMalloc((void **) &d_nt, seq_cap));
	CHECK_HIP(hipMalloc((void **) &d_p, seq_cap * sizeof(double)));
	CHECK_HIP(hipMalloc((void **) &d_cursor, 8));
	CHECK_HIP(hipMalloc((void **) &d_ctr, PAMD_NUM_COUNTERS * 8));
	CHECK_HIP(hipMemset(d_ctr, 0, PAMD_NUM_COUNTERS * 8));
	char *h_text = (char *) malloc(text_cap);
	double *h_quality = (double *) malloc(chunk_pairs * sizeof(double));
	pamd_pair_result *h_res = (pamd_pair_result *) malloc(chunk_pairs * sizeof(pamd_pair_result));

	size_t fpos = 0, rpos = 0, dropped = 0;
	int stop = PAMD_FQ_END;
	unsigned long long stop_record = 0, records = 0;
	while (fpos < fbytes && rpos < rbytes) {
		const size_t fn = fbytes - fpos < chunk_bytes ? fbytes - fpos : chunk_bytes;
		const size_t rn = rbytes - rpos < chunk_bytes ? rbytes - rpos : chunk_bytes;
		CHECK_HIP(hipMemcpy(d_f, ftext + fpos, fn, hipMemcpyHostToDevice));
		CHECK_HIP(hipMemcpy(d_r, rtext + rpos, rn, hipMemcpyHostToDevice));
		pamd_fastq_status st;
		CHECK_PAMD(pamd_fastq_parse_device(fq, d_f, fn, d_r, rn, chunk_pairs, d_fwd, stride, d_rev, stride, d_flen, d_rlen, d_ids, &st, NULL));
		if (st.n_good > 0) {
			pamd_batch b;
			memset(&b, 0, sizeof b);
			b.n_pairs = st.n_good;
			b.fwd = d_fwd;
			b.fwd_stride = stride;
			b.rev = d_rev;
			b.rev_stride = stride;
			b.flen = d_flen;
			b.rlen = d_rlen;
			b.results = d_res;
			b.seq_nt = d_nt;
			b.seq_p = d_p;
			b.seq_stride = 0;
			b.seq_capacity = seq_cap;
			b.seq_cursor = d_cursor;
			b.counters = d_ctr;
			CHECK_HIP(hipMemset(d_cursor, 0, 8));
			CHECK_PAMD(pamd_assemble_device(a, &b, NULL));
			size_t text_bytes = 0, n_fix = 0;
			uint64_t fix[2 * 64];
			CHECK_PAMD(pamd_format_device(fmt, format, st.n_good, d_res, d_nt, d_p, d_ids, d_f, d_text, text_cap, &text_bytes, fix, 64, &n_fix, NULL));
			CHECK_HIP(hipMemcpy(h_text, d_text, text_bytes, hipMemcpyDeviceToHost));
			if (n_fix > 0) {	/* the "%f" fields the device would not decide: re-printed as output.c:93 prints them */
				CHECK_HIP(hipMemcpy(h_res, d_res, st.n_good * sizeof(pamd_pair_result), hipMemcpyDeviceToHost));
				for (size_t k = 0; k < n_fix; k++) {
					char tmp[32];
					snprintf(tmp, sizeof tmp, "%f", exp(h_res[fix[2 * k + 1]].quality));
					memcpy(h_text + fix[2 * k], tmp, 8);
				}
			}
			fwrite(h_text, 1, text_bytes, out);
		}
		records += st.n_good;
		dropped += st.n_dropped;
		if (st.stop_code != PAMD_FQ_END) {	/* the reader gives up at this record, as stream_next_seq() does */
			stop = st.stop_code;
			stop_record = records;
			break;
		}
		if (st.n_records == 0)
			break;	/* only a partial record is left: PANDA_CODE_PREMATURE_EOF in the reference when any bytes remain */
		fpos += st.fwd_consumed;
		rpos += st.rev_consumed;
	}
	fclose(out);
	int64_t ctr[PAMD_NUM_COUNTERS];
	CHECK_HIP(hipMemcpy(ctr, d_ctr, sizeof ctr, hipMemcpyDeviceToHost));
	/* records the reader drops never reach the assembler in the reference (fastq.c:192) */
	printf("STAT\tREADS\t%lld\nSTAT\tNOALGN\t%lld\nSTAT\tLOWQ\t%lld\nSTAT\tBADR\t%lld\nSTAT\tOK\t%lld\nSTAT\tNOFP\t%lld\nSTAT\tNORP\t%lld\nSTOP\t%d\t%llu\n",
	       (long long) ctr[PAMD_CTR_COUNT] - (long long) dropped, (long long) ctr[PAMD_CTR_NOALGN], (long long) ctr[PAMD_CTR_LOWQ],
	       (long long) ctr[PAMD_CTR_BADR] - (long long) dropped, (long long) ctr[PAMD_CTR_OK], (long long) ctr[PAMD_CTR_NOFP], (long long) ctr[PAMD_CTR_NORP], stop, stop_record);
	pamd_formatter_destroy(fmt);
	pamd_fastq_parser_destroy(fq);
	pamd_assembler_destroy(a);
	return 0;
}
