/*
 * pandaseq_amd_io.h -- C ABI of the two data formats either side of the assembly hot path
 * (SURVEY.md 8f rows 1 and 2), both done on the device so that text goes in and text comes out
 * and the host only moves bytes:
 *
 *   FASTQ ingest      paired FASTQ text -> the rows of pamd_qual pamd_assemble_device() consumes.
 *                     Replaces the reader PANDAseq builds with panda_create_fastq_reader()
 *                     (fastq.c:200-237): its per-pair callback stream_next_seq() (fastq.c:108-198),
 *                     read_seq() (fastq.c:45-107), the line reader panda_linebuf_next()
 *                     (linebuf.c:57-89), the ASCII->nt tables (nt.c:46-114) and the header parser
 *                     panda_seqid_parse_fail() (seqid.c:142-285) with panda_seqid_equal() (:96-101).
 *   result formatting accepted pairs -> FASTA / FASTQ text.  Replaces panda_output_fasta() /
 *                     panda_output_fastq() (output.c:85-126) with panda_writer_append_id()
 *                     (writer.c:234-238 -> seqid.c:109-116), panda_nt_to_ascii() (nt.c:167-173) and
 *                     panda_result_phred() (nt.c:126-150).
 *
 * Plain pointers and sizes; every text / row pointer here is a DEVICE pointer.  Decompression
 * (gzip / bzip2, fileio.c) and file handling stay with the caller: these entry points start from
 * decompressed text already in HBM.  Errors as in pandaseq_amd.h (0 or a negative PAMD_E* code,
 * message from pamd_last_error()); there is no CPU fallback.
 */
#ifndef PANDASEQ_AMD_IO_H
#define PANDASEQ_AMD_IO_H

#include "pandaseq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PAMD_TAG_LEN 50			/* PANDA_TAG_LEN, pandaseq-common.h:45 */
#define PAMD_LINE_MAX (10 * PAMD_MAX_LEN)	/* the line reader's buffer, linebuf.c:25 */

enum { PAMD_TAG_PRESENT = 0, PAMD_TAG_ABSENT = 1, PAMD_TAG_OPTIONAL = 2 };	/* PandaTagging, pandaseq-common.h:165-178 */
/* PandaIdFmt, pandaseq-common.h:188-195 */
enum { PAMD_IDFMT_UNKNOWN = 0, PAMD_IDFMT_SRA, PAMD_IDFMT_CASAVA_1_4, PAMD_IDFMT_CASAVA_1_7, PAMD_IDFMT_EBI_SRA, PAMD_IDFMT_CASAVA_CONVERTED };

/* Why the paired reader stops: the PandaCode stream_next_seq() / read_seq() log before returning false. */
enum {
	PAMD_FQ_END = 0,		/* ran out of whole records (fastq.c:123,130) */
	PAMD_FQ_BADID_FWD = 1,		/* PANDA_CODE_ID_PARSE_FAILURE, forward header (fastq.c:126-129) */
	PAMD_FQ_BADID_REV = 2,		/* PANDA_CODE_ID_PARSE_FAILURE, reverse header (fastq.c:133-136) */
	PAMD_FQ_NOTPAIRED = 3,		/* PANDA_CODE_NOT_PAIRED (fastq.c:137-140) */
	PAMD_FQ_BADNT = 4,		/* PANDA_CODE_BAD_NT (fastq.c:63-66) */
	PAMD_FQ_READLEN = 5,		/* PANDA_CODE_READ_TOO_LONG (fastq.c:75-77) */
	PAMD_FQ_BADSEQ = 6,		/* PANDA_CODE_PARSE_FAILURE (fastq.c:78-81) */
	PAMD_FQ_NOQUAL = 7,		/* PANDA_CODE_NO_QUALITY_INFO (fastq.c:96-99) */
	PAMD_FQ_EOF = 8,		/* PANDA_CODE_PREMATURE_EOF: the line reader gave up inside a record */
	PAMD_FQ_LONG_HEADER = 9,	/* a header line the line reader cannot hold: the reference ends the stream silently */
	PAMD_FQ_ROW_OVERFLOW = 10	/* not a reference condition: a read longer than the caller's row stride */
};

/* One parsed sequence header: what panda_seq_identifier (pandaseq-common.h:235-244) holds, with the
 * four strings kept as spans of the forward header text instead of copies.  48 bytes.
 * instrument = text[hdr_off, hdr_off + instr_len); run / flowcell / tag likewise at hdr_off + *_off.
 * For the two Short Read Archive formats the reference prints "<E|S>RR<number>" into instrument
 * (seqid.c:163): `sra` holds that number and instr_len is 3 (the letters); otherwise sra = -1. */
typedef struct {
	uint64_t hdr_off;	/* offset in the forward text of the character after the leading '@' */
	int32_t lane, tile, x, y;
	int32_t sra;
	uint16_t instr_len;
	uint16_t run_off, run_len;
	uint16_t fc_off, fc_len;
	uint16_t tag_off, tag_len;
	uint8_t format;		/* PAMD_IDFMT_* of the forward header */
	uint8_t dir;		/* mate number the forward header carries */
	uint8_t flag;		/* 0 = a pair; 1 = dropped by the reader (empty forward read, fastq.c:192); else 16 + PAMD_FQ_* */
	uint8_t reserved[3];
} pamd_seqid;

typedef struct {
	uint64_t n_records;	/* whole record pairs both texts hold (at most max_pairs) */
	uint64_t n_good;	/* records before the first one the reader fails on: rows [0, n_good) are valid */
	uint64_t n_dropped;	/* of those, records the reader drops (empty forward read): their flen is 0 and flag 1 */
	uint64_t fwd_consumed;	/* bytes of each text the n_records records span: resume the next chunk here */
	uint64_t rev_consumed;
	int32_t stop_code;	/* PAMD_FQ_*: why record n_good failed; PAMD_FQ_END when n_good == n_records */
	int32_t stop_in_reverse;	/* the failing read was the reverse one */
	int32_t seen_under_64;	/* fastq.c:88-90,141-144: feeds the PHRED-offset hint of fastq.c:203-207 */
	uint32_t max_flen, max_rlen;	/* longest reads among the pairs kept (good records that were not dropped) */
	uint32_t reserved;
} pamd_fastq_status;

typedef struct pamd_fastq_parser pamd_fastq_parser;

/* qualmin: the PHRED offset (33 or 64; `-6` selects 64, args_fastq.c).  policy: PAMD_TAG_*. */
pamd_fastq_parser *pamd_fastq_parser_create(int device, int qualmin, int policy);
void pamd_fastq_parser_destroy(pamd_fastq_parser *p);

/* Parse every whole record pair of two FASTQ texts resident in HBM.
 *   fwd_text / rev_text  decompressed FASTQ, 16-byte aligned, each < 4 GiB; a trailing partial record
 *                        is left alone (see fwd_consumed / rev_consumed)
 *   fwd_rows / rev_rows  max_pairs rows of pamd_qual, `*_stride` bases apart (multiple of 8): exactly the
 *                        layout pamd_batch wants; the reverse read is complemented, not reversed
 *                        (iupac_reverse, nt.c:81-114); bases past a read's length are zeroed
 *   flen / rlen          read lengths (<= PAMD_MAX_LEN; longer sequence lines are cut as fastq.c:62 does)
 *   ids                  one pamd_seqid per record
 * Record i of the texts becomes row i.  Records the reference reader would drop stay in place with
 * flen = 0 and ids[i].flag = 1 (the assembler counts such a row as a bad read: subtract n_dropped from
 * its `count` and `badr` counters, or skip flagged rows, to get the reference's numbers).
 * Synchronises on `stream` before returning (the status is read back). */
int pamd_fastq_parse_device(pamd_fastq_parser *p, const char *fwd_text, size_t fwd_bytes, const char *rev_text, size_t rev_bytes, size_t max_pairs, pamd_qual *fwd_rows, size_t fwd_stride, pamd_qual *rev_rows, size_t rev_stride, uint16_t *flen, uint16_t *rlen, pamd_seqid *ids, pamd_fastq_status *status, void *stream);
/* HIP-event time (ms) of the kernels of the last parse call: line index, record parse */
int pamd_fastq_last_kernel_ms(pamd_fastq_parser *p, float *index_ms, float *parse_ms);

/* ---- result formatting ---------------------------------------------------------------------- */
enum { PAMD_OUT_FASTA = 0, PAMD_OUT_FASTQ = 1 };

typedef struct pamd_formatter pamd_formatter;
pamd_formatter *pamd_formatter_create(int device);
void pamd_formatter_destroy(pamd_formatter *f);

/* Format the accepted pairs of an assembled batch, in batch order, as the text panda_output_fasta() /
 * panda_output_fastq() would append for each (output.c:85-126):
 *     >instrument:run:flowcell:lane:tile:x:y:tag;%f\n<bases>\n                     (%f of exp(quality))
 *     @...;%f\n<bases>\n+\n<33 + panda_result_phred(p) per base>\n
 *   results / seq_nt / seq_p   the outputs of pamd_assemble_device() for n_pairs pairs
 *   ids / hdr_text             what pamd_fastq_parse_device() produced and the forward text it points into
 *   out / out_capacity         device text buffer; out_bytes (host) receives the size written
 *   fixups / n_fixups          host array (capacity max_fixups) receiving the byte offsets in `out` of
 *                              "0.dddddd" fields whose sixth decimal the device could not decide against
 *                              its own exp() (|distance to a rounding boundary| < 1e-9; about one pair in
 *                              10^9), each followed by the pair index: the caller re-prints those eight
 *                              characters with the host's printf("%f", exp(quality)).
 * Pairs whose status is not PAMD_ST_OK, whose sequence is empty (output.c:89,107) or whose ids[].flag is
 * non-zero produce nothing.  Synchronises on `stream`. */
int pamd_format_device(pamd_formatter *f, int format, size_t n_pairs, const pamd_pair_result *results, const uint8_t *seq_nt, const double *seq_p, const pamd_seqid *ids, const char *hdr_text, char *out, size_t out_capacity, size_t *out_bytes, uint64_t *fixups, size_t max_fixups, size_t *n_fixups, void *stream);
int pamd_format_last_kernel_ms(pamd_formatter *f, float *ms);

/* ---- files in, file out ------------------------------------------------------------------------------
 * The loop a PANDAseq front end runs around the three device stages, as one call: two FASTQ files -> parse -> assemble ->
 * format -> one FASTA / FASTQ file.  It stands where main.c's chain stands -- a reader thread per input behind a queue
 * (async.c:103-149), workers pulling pairs (pool.c:71-108, mux.c:132-161), a writer thread (writer.c) -- with the workers
 * replaced by the device and the host reduced to keeping both directions of the PCIe link busy: reader threads pread() blocks
 * of either file into pinned memory, a copier thread per file sends them up on a stream of its own, the calling thread gathers
 * each chunk's text on the device (records straddle blocks; the unconsumed tail of a chunk is carried into the next), runs the
 * stages and sends the text back in pieces, writer threads pwrite() the pieces.  An input that is gzip or bzip2 text (by its first two
 * bytes) or not a regular file (a FIFO) is read in order by ONE thread through zlib / libbz2, the way the reference opens every
 * input (fileio.c:70-96; stats->in_bytes then counts the inflated text), at the rate those inflate; the output is always plain text, into
 * a file (pieces written at their offsets) or into something that cannot seek (a pipe, a FIFO: one writer, the pieces in order).
 * The bytes written are those the reference's reader -> assembler -> output.c chain writes for the same input, in input order.
 *   a        a configured assembler (algorithm, primers, modules ... as set by the caller); its counters are not touched:
 *            the run's own are returned in stats->counters (PAMD_CTR_* of pandaseq_amd.h; as with pamd_fastq_parse_device,
 *            subtract stats->dropped from COUNT and BADR for the reference's numbers)
 *   options  pairs_per_chunk 0 = 1 000 000 (at most 3 000 000); qualmin 0 = 33; reader_threads / writer_threads 0 = 3 per file / 1 (writers of ONE file take turns on its lock)
 *   stats    pairs = records handed to the assembler; stop_code = PAMD_FQ_* (PAMD_FQ_END: the input was read to its end) and
 *            stop_record = the record the reader gave up at; seconds_setup = allocations and thread start, seconds_pump = the rest */
typedef struct {
	const char *forward_path, *reverse_path, *out_path;
	int format;		/* PAMD_OUT_FASTA / PAMD_OUT_FASTQ */
	int qualmin, policy;	/* as pamd_fastq_parser_create */
	size_t pairs_per_chunk;
	int reader_threads, writer_threads;
} pamd_pipe_options;
typedef struct {
	uint64_t pairs, dropped, chunks, in_bytes, out_bytes, stop_record;
	int32_t stop_code, reserved;
	double seconds_setup, seconds_pump;
	int64_t counters[PAMD_NUM_COUNTERS];
} pamd_pipe_stats;
int pamd_pipe_run(pamd_assembler *a, int device, const pamd_pipe_options *options, pamd_pipe_stats *stats);

/* The same run over N devices: what panda_run_pool (pool.c:110-181) is to the reference -- a worker per thread there, each with
 * an assembler of its own configured alike (pool.c:142-150), all fed by one mux and writing through one writer -- with a device
 * for a worker and a chunk of `pairs_per_chunk` records for a pair.  The two files are cut at record boundaries (a record is four
 * lines of either file, fastq.c:46-197) found by counting line ends on the host; chunk k goes to devices[k mod n_devices], whose
 * own reader and copier threads bring exactly its bytes up over THAT device's link; every device runs parse -> assemble ->
 * format on its chunks; the text of chunk k is written behind the text of chunks 0 .. k-1, so the output file is byte for byte
 * pamd_pipe_run()'s, and stats->counters is the sum of the chunks' counter blocks (the STAT all-reduce, done on the host because
 * the devices share a process).  A chunk in which the reader gives up ends the run as it ends pamd_pipe_run(): later chunks,
 * already dealt, are dropped.  stats->chunks counts the chunks dealt.
 *   assemblers   n_devices configured assemblers, assemblers[i] created on devices[i] (a device may be named twice: two
 *                workers on one GPU, each with its own assembler)
 * Inputs that are read in order by one thread (gzip / bzip2 text, FIFOs) gain nothing from more devices: such a run is handed to
 * pamd_pipe_run(assemblers[0], devices[0], ...). */
int pamd_pipe_run_multi(pamd_assembler *const *assemblers, const int *devices, int n_devices, const pamd_pipe_options *options, pamd_pipe_stats *stats);

/* The STAT block a reference worker prints when its input ends (pool.c:86-104; OVERLAPS as proxy.c:175-190 prints it), from a
 * counter block -- one assembler's, one run's (pamd_pipe_stats::counters) or a sum over devices:
 *     STAT\tREADS\t<count>  [NOFP] [NORP]  NOALGN  LOWQ  BADR  SLOW  <module>\t<rejected> (module.c:208-216: only if > 0)  OK  OVERLAPS\t<n0> <n1> ...
 * NOFP / NORP only when that primer is set; every line prefixed "<name>\t" when name is not NULL (proxy.c:165-172); TIME and
 * ELAPSED (pool.c:60-65) are the caller's clock and not printed.  Returns the length written (the text is NUL-terminated) or
 * PAMD_EINVAL when `cap` is too small. */
int pamd_stat_block(const int64_t *counters, int has_forward_primer, int has_reverse_primer, const char *const *module_names, int n_modules, const char *name, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
