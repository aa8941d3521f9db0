/*
 * pandaseq_amd_compat.h -- the reference's own per-assembler API for the hot path
 * (pandaseq-assembler.h:40-405 + the types of pandaseq-common.h it needs), implemented on top of
 * the batch ABI of pandaseq_amd.h.  A program written against <pandaseq.h> that only uses the
 * functions below can be re-linked against libpandaseq_amd.so and get the GPU path:
 *
 *   panda_assembler_next()      assembler.c:350-366   pulls a BATCH of pairs from the PandaNextSeq
 *                                                     source, assembles it in one kernel launch and
 *                                                     hands results back one at a time, in source order
 *   panda_assembler_assemble()  assembler.c:368-383   one caller-supplied pair (a batch of one)
 *
 * Struct layouts are ABI-identical to the reference (sizes checked by static asserts in
 * pamd_compat.cpp: panda_qual 2, panda_result 16, panda_seq_identifier 368, panda_result_seq 480).
 * struct panda_assembler / panda_algorithm / panda_module are private in the reference
 * (assembler.h:28, algo.h:28, module.c:84) and are private here too.
 *
 * When this library is linked INTO libpandaseq, build it with
 * -DPAMD_WITH_REFERENCE_OBJECTS: it then defines exactly what the three files it replaces define --
 * assembler.c, assembler_support.c and module.c (module.c works on struct panda_assembler's private
 * fields, module.c:99-220, so it goes with them) -- and leaves every other panda_* name below
 * (algorithm objects, nt / offset / seqid / misc / log-proxy helpers, the FASTQ reader constructors of
 * fastq.c and fileio.c) to the reference's own objects.  tests/test_link_modes_cpu.py holds both sides
 * of that to the reference's sources.
 */
#ifndef PANDASEQ_AMD_COMPAT_H
#define PANDASEQ_AMD_COMPAT_H

#include <stdarg.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdio.h>

#ifndef PANDA_API
#	define PANDA_API 3	/* pandaseq.h:61: the plugin ABI version panda_module_load() demands */
#endif

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PANDA_TAG_LEN
#define PANDA_TAG_LEN 50	/* pandaseq-common.h:45 */
#endif
#define PANDA_DEFAULT_NUM_KMERS 2	/* pandaseq-assembler.h:51 */

typedef char panda_nt;	/* pandaseq-common.h:166 */

typedef struct {	/* pandaseq-common.h:210-219 */
	panda_nt nt;
	char qual;
} panda_qual;

typedef struct {	/* pandaseq-common.h:221-230 */
	panda_nt nt;
	double p;
} panda_result;

typedef struct {	/* pandaseq-common.h:235-244 */
	char instrument[100];
	char run[100];
	char flowcell[100];
	int lane;
	int tile;
	int x;
	int y;
	char tag[PANDA_TAG_LEN];
} panda_seq_identifier;

typedef struct {	/* pandaseq-common.h:277-330 */
	double quality;
	size_t degenerates;
	panda_seq_identifier name;
	panda_result *sequence;
	size_t sequence_length;
	panda_qual const *forward;
	size_t forward_length;
	panda_qual const *reverse;
	size_t reverse_length;
	size_t forward_offset;
	size_t reverse_offset;
	size_t overlap_mismatches;
	size_t overlaps_examined;
	size_t overlap;
	double estimated_overlap_probability;
} panda_result_seq;

typedef struct panda_assembler *PandaAssembler;	/* pandaseq-common.h:66 */
typedef struct panda_algorithm *PandaAlgorithm;	/* pandaseq-common.h:56 */
typedef const struct panda_algorithm_class *PandaAlgorithmClass;	/* pandaseq-common.h:61 */
typedef struct panda_module *PandaModule;	/* pandaseq-common.h:98 */
typedef struct panda_log_proxy *PandaLogProxy;	/* pandaseq-common.h:93; opaque, passed through to callbacks */

/* callback types, pandaseq-common.h:338-522 */
typedef void (*PandaDestroy) (void *user_data);
/* pandaseq-common.h:430-443: fill `buffer` with up to buffer_length bytes; *read = 0 at the end of the data */
typedef bool (*PandaBufferRead) (char *buffer, size_t buffer_length, size_t *read, void *user_data);
/* pandaseq-common.h:165-178 */
typedef enum { PANDA_TAG_PRESENT, PANDA_TAG_ABSENT, PANDA_TAG_OPTIONAL } PandaTagging;
typedef bool (*PandaNextSeq) (panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data);
typedef void (*PandaFailAlign) (PandaAssembler assembler, const panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length, void *user_data);
typedef bool (*PandaCheck) (PandaLogProxy logger, const panda_result_seq *sequence, void *user_data);
typedef bool (*PandaPreCheck) (PandaLogProxy logger, const panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length, void *user_data);
typedef bool (*PandaModuleCallback) (PandaAssembler assembler, PandaModule module, size_t rejected, void *data);
typedef const panda_result_seq *(*PandaAssemble) (void *user_data, panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length);
typedef PandaAlgorithm (*PandaAlgorithmCreate) (const char *args);
typedef double (*PandaComputeMatch) (void *private_data, bool match, char a, char b);
typedef double (*PandaComputeOverlap) (void *private_data, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length, size_t overlap);

struct panda_algorithm_class {	/* pandaseq-common.h:594-605 */
	size_t data_size;
	const char *name;
	PandaAlgorithmCreate create;
	PandaDestroy data_destroy;
	PandaComputeOverlap overlap_probability;	/* the seven built-in classes: host functions (pamd_host_scorers.cpp); batches are scored on the GPU */
	PandaComputeMatch match_probability;
	const double prob_unpaired;
};

/* ---- constructors / lifetime (pandaseq-assembler.h:40-66,183-192) ---- */
/* ---- the FASTQ reader (pandaseq.h:111-127,238-264, pandaseq-assembler.h:77-101), parsing whole chunks of text on
 * the device (pamd_compat_fastq.cpp).  Plain, gzip and bzip2 files (by their first two bytes, fileio.c:70-96; libbz2.so.1 is bound at run time); no separate index-read file (`index` must
 * be NULL). ---- */
PandaNextSeq panda_create_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, PandaBufferRead index, void *index_data, PandaDestroy index_destroy, void **user_data, PandaDestroy *destroy);
PandaBufferRead panda_open_buffer(const char *file_name, PandaLogProxy logger, void **user_data, PandaDestroy *destroy);
PandaNextSeq panda_open_fastq(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, const char *index, void **user_data, PandaDestroy *destroy);
PandaAssembler panda_assembler_new_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy);
PandaAssembler panda_assembler_open_fastq(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy);	/* fileio.c:139 */
PandaAssembler panda_assembler_open(const char *forward, const char *reverse, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy);	/* the name pandaseq-assembler.h:96 declares */
/* which GPU the readers created from now on parse on (default 0), and the PAMD_FQ_* code (pandaseq_amd_io.h) the
 * last reader on this thread stopped with -- the PandaCode the reference's reader would have logged */
void pamd_compat_set_fastq_device(int device);
/* The device reader under its own name, same signature as panda_create_fastq_reader (fastq.c:200-237): what that name IS in
 * the standalone build, and what a program linked with the reference's fastq.c calls to have its text parsed on the GPU. */
PandaNextSeq pamd_compat_create_fastq_reader(PandaBufferRead forward, void *forward_data, PandaDestroy forward_destroy, PandaBufferRead reverse, void *reverse_data, PandaDestroy reverse_destroy, PandaLogProxy logger, unsigned char qualmin, PandaTagging policy, PandaBufferRead index, void *index_data, PandaDestroy index_destroy, void **user_data, PandaDestroy *destroy);
int pamd_compat_last_fastq_stop(void);
/* misc.c:38-41: what PANDA_MAX_LEN expands to (pandaseq-common.h:44); plugins call it */
size_t panda_max_len(void);
PandaAssembler panda_assembler_new(PandaNextSeq next, void *next_data, PandaDestroy next_destroy, PandaLogProxy logger);
PandaAssembler panda_assembler_new_kmer(PandaNextSeq next, void *next_data, PandaDestroy next_destroy, PandaLogProxy logger, size_t num_kmers);
PandaAssembler panda_assembler_ref(PandaAssembler assembler);
void panda_assembler_unref(PandaAssembler assembler);
void panda_assembler_copy_configuration(PandaAssembler dest, PandaAssembler src);	/* :147 */

/* ---- the two entry points of the hot path ---- */
const panda_result_seq *panda_assembler_assemble(PandaAssembler assembler, panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length);	/* :134 */
const panda_result_seq *panda_assembler_next(PandaAssembler assembler);	/* :175 */

/* ---- modules (pandaseq-assembler.h:112-167, pandaseq-module.h) ---- */
PandaModule panda_module_new(const char *name, PandaCheck check, PandaPreCheck precheck, void *user_data, PandaDestroy cleanup);	/* module.c:300-321 */
/* "min_phred:10" / "min_overlapbits:15" -> a module the kernel applies itself (replaces dlopen-ing
 * plugin_min_phred.c / plugin_min_overlapbits.c through module.c:227-298).  Any other "name[:args]" is a
 * plugin file as module.c loads them: name, name.so, then $PANDASEQ_PLUGIN_DIR/name[.so]; its
 * <module>_LTX_api must equal PANDA_API and its <module>_LTX_opener (pandaseq-plugin.h:33) supplies the
 * precheck / check callbacks, which run on the host around the device batch like panda_module_new()'s. */
PandaModule panda_module_load(PandaLogProxy logger, const char *path);
PandaModule panda_module_ref(PandaModule module);
void panda_module_unref(PandaModule module);
const char *panda_module_get_name(PandaModule module);
int panda_module_get_api(PandaModule module);	/* module.c:360-363 */
const char *panda_module_get_args(PandaModule module);	/* pandaseq-module.h:94 (declared there, defined nowhere in the reference) */
const char *panda_module_get_description(PandaModule module);	/* module.c:365-372 */
const char *panda_module_get_usage(PandaModule module);	/* module.c:384-391 */
const char *panda_module_get_version(PandaModule module);	/* module.c:379-382 */
bool panda_assembler_add_module(PandaAssembler assembler, PandaModule module);
size_t panda_assembler_add_modules(PandaAssembler assembler, PandaModule *modules, size_t modules_length);
bool panda_assembler_foreach_module(PandaAssembler assembler, PandaModuleCallback callback, void *callback_data);
void panda_assembler_module_stats(PandaAssembler assembler);

/* ---- algorithms (pandaseq-algorithm.h; algo.c:34-94, algo_*.c) ---- */
extern const struct panda_algorithm_class panda_algorithm_simple_bayes_class;
extern const struct panda_algorithm_class panda_algorithm_pear_class;
extern const struct panda_algorithm_class panda_algorithm_rdp_mle_class;
PandaAlgorithm panda_algorithm_new(PandaAlgorithmClass clazz);
/* the name-sorted registry the command line looks algorithms up in (algo.c:96-133); the seven classes above
 * are registered when the library loads.  A class registered by the caller is listed, but an assembler only
 * accepts the seven the kernel implements (panda_assembler_set_algorithm logs and keeps the old one otherwise). */
extern PandaAlgorithmClass *panda_algorithms;
extern size_t panda_algorithms_length;
void panda_algorithm_register(PandaAlgorithmClass clazz);
PandaAlgorithm panda_algorithm_ref(PandaAlgorithm algo);
void panda_algorithm_unref(PandaAlgorithm algo);
bool panda_algorithm_is_a(PandaAlgorithm algo, PandaAlgorithmClass clazz);
PandaAlgorithmClass panda_algorithm_class(PandaAlgorithm algo);
void *panda_algorithm_data(PandaAlgorithm algo);
/* algo.c:27-32: log probability of one overlap position given the two bases; built-in classes answer from the host copy of the tables */
double panda_algorithm_quality_compare(PandaAlgorithm algorithm, const panda_qual *a, const panda_qual *b);
PandaAlgorithm panda_algorithm_simple_bayes_new(void);
double panda_algorithm_simple_bayes_get_error_estimation(PandaAlgorithm algorithm);
void panda_algorithm_simple_bayes_set_error_estimation(PandaAlgorithm algorithm, double q);
PandaAlgorithm panda_algorithm_pear_new(void);
double panda_algorithm_pear_get_random_base_log_p(PandaAlgorithm algorithm);
void panda_algorithm_pear_set_random_base_log_p(PandaAlgorithm algorithm, double log_p);
PandaAlgorithm panda_algorithm_rdp_mle_new(void);
/* the four remaining scorers (algo_uparse.c, algo_flash.c, algo_ea_util.c, algo_stitch.c; pandaseq-algorithm.h:57-215) */
extern const struct panda_algorithm_class panda_algorithm_uparse_class;
extern const struct panda_algorithm_class panda_algorithm_flash_class;
extern const struct panda_algorithm_class panda_algorithm_ea_util_class;
extern const struct panda_algorithm_class panda_algorithm_stitch_class;
PandaAlgorithm panda_algorithm_uparse_new(void);
double panda_algorithm_uparse_get_error_estimation(PandaAlgorithm algorithm);
void panda_algorithm_uparse_set_error_estimation(PandaAlgorithm algorithm, double q);
PandaAlgorithm panda_algorithm_flash_new(void);
PandaAlgorithm panda_algorithm_ea_util_new(void);
PandaAlgorithm panda_algorithm_stitch_new(void);

/* ---- getters / setters (pandaseq-assembler.h:199-405, assembler_support.c:177-410) ---- */
PandaAlgorithm panda_assembler_get_algorithm(PandaAssembler assembler);
void panda_assembler_set_algorithm(PandaAssembler assembler, PandaAlgorithm algorithm);
long panda_assembler_get_bad_read_count(PandaAssembler assembler);
long panda_assembler_get_count(PandaAssembler assembler);
void panda_assembler_set_fail_alignment(PandaAssembler assembler, PandaFailAlign handler, void *handler_data, PandaDestroy handler_destroy);
long panda_assembler_get_failed_alignment_count(PandaAssembler assembler);
panda_nt *panda_assembler_get_forward_primer(PandaAssembler assembler, size_t *length);
void panda_assembler_set_forward_primer(PandaAssembler assembler, panda_nt *sequence, size_t length);
size_t panda_assembler_get_forward_trim(PandaAssembler assembler);
void panda_assembler_set_forward_trim(PandaAssembler assembler, size_t trim);
size_t panda_assembler_get_longest_overlap(PandaAssembler assembler);
long panda_assembler_get_low_quality_count(PandaAssembler assembler);
int panda_assembler_get_minimum_overlap(PandaAssembler assembler);
void panda_assembler_set_minimum_overlap(PandaAssembler assembler, int overlap);
int panda_assembler_get_maximum_overlap(PandaAssembler assembler);
void panda_assembler_set_maximum_overlap(PandaAssembler assembler, int overlap);
const char *panda_assembler_get_name(PandaAssembler assembler);
void panda_assembler_set_name(PandaAssembler assembler, const char *name);
long panda_assembler_get_no_forward_primer_count(PandaAssembler assembler);
long panda_assembler_get_no_reverse_primer_count(PandaAssembler assembler);
size_t panda_assembler_get_num_kmer(PandaAssembler assembler);
long panda_assembler_get_ok_count(PandaAssembler assembler);
long panda_assembler_get_overlap_count(PandaAssembler assembler, size_t overlap);
bool panda_assembler_get_primers_after(PandaAssembler assembler);
void panda_assembler_set_primers_after(PandaAssembler assembler, bool after);
panda_nt *panda_assembler_get_reverse_primer(PandaAssembler assembler, size_t *length);
void panda_assembler_set_reverse_primer(PandaAssembler assembler, panda_nt *sequence, size_t length);
size_t panda_assembler_get_reverse_trim(PandaAssembler assembler);
void panda_assembler_set_reverse_trim(PandaAssembler assembler, size_t trim);
long panda_assembler_get_slow_count(PandaAssembler assembler);
double panda_assembler_get_threshold(PandaAssembler assembler);
void panda_assembler_set_threshold(PandaAssembler assembler, double threshold);
PandaLogProxy panda_assembler_get_logger(PandaAssembler assembler);
double panda_assembler_get_primer_penalty(PandaAssembler assembler);
void panda_assembler_set_primer_penalty(PandaAssembler assembler, double threshold);

/* ---- per-base helpers plugins call on the results they are handed (pandaseq-nt.h:76-88, nt.c:120-150) ---- */
double panda_quality_log_probability(const panda_qual *q);	/* qual_score[PHREDCLAMP(q->qual)] */
char panda_result_phred(const panda_result *r);			/* the reference's bisection over qual_score, quirks included */
double panda_quality_probability(const panda_qual *q);		/* nt.c:116-119 */
/* offset.c:35-38, :103-112, :124-133: the primer scans for a caller's own rows (plugins call them; the assembler's run on the device) */
double panda_log1mexp(double p);
size_t panda_compute_offset_qual(double threshold, double penalty, bool reverse, const panda_qual *haystack, size_t haystack_length, const panda_nt *needle, size_t needle_length);
size_t panda_compute_offset_result(double threshold, double penalty, bool reverse, const panda_result *haystack, size_t haystack_length, const panda_nt *needle, size_t needle_length);
/* nt.c:46-114,152-173: IUPAC letters <-> base sets */
panda_nt panda_nt_from_ascii(char c);
panda_nt panda_nt_from_ascii_complement(char c);
panda_nt panda_nt_complement(panda_nt nt);
char panda_nt_to_ascii(panda_nt val);
/* proxy.c:141-162, standalone build: to stderr whatever the proxy (the log proxy itself is the reference's) */
void panda_log_proxy_write_str(PandaLogProxy proxy, const char *str);
void panda_log_proxy_write_f(PandaLogProxy proxy, const char *format, ...);
void panda_log_proxy_perror(PandaLogProxy proxy, const char *prefix);
/* seqid.c:46-127: the identifier of a result -- compare, copy, print (parsing header lines is the device parser's job) */
typedef void (*PandaPrintf) (void *data, const char *format, ...);	/* pandaseq-common.h:530-534 */
void panda_seqid_clear(panda_seq_identifier *id);
int panda_seqid_compare(const panda_seq_identifier *one, const panda_seq_identifier *two);
bool panda_seqid_equal(const panda_seq_identifier *one, const panda_seq_identifier *two);
void panda_seqid_copy(const panda_seq_identifier *src, panda_seq_identifier *dest);
void panda_seqid_xprint(const panda_seq_identifier *id, PandaPrintf xprintf, void *x);
void panda_seqid_print(const panda_seq_identifier *id, FILE *file);
const char *panda_seqid_str(const panda_seq_identifier *id);
/* seqid.c:127-285: one header line (without its '@') -> the identifier; the mate number it carries (1; 2 or 3), 0 when it does not
 * parse.  The host form of what the device parser does per record (pamd_io.hip parse_header): for a caller that holds single
 * header strings.  `end_ptr` (may be NULL) receives where parsing stopped.  One difference, where the reference's behaviour is
 * undefined: an instrument, run or flowcell field of 100 characters or more fails here; the reference writes past its 100-byte
 * field for up to 101. */
typedef enum { PANDA_IDFMT_UNKNOWN, PANDA_IDFMT_SRA, PANDA_IDFMT_CASAVA_1_4, PANDA_IDFMT_CASAVA_1_7, PANDA_IDFMT_EBI_SRA, PANDA_IDFMT_CASAVA_CONVERTED } PandaIdFmt;	/* pandaseq-common.h:188-195 */
int panda_seqid_parse(panda_seq_identifier *id, const char *input, PandaTagging policy);
int panda_seqid_parse_fail(panda_seq_identifier *id, const char *input, PandaTagging policy, PandaIdFmt *detected_format, const char **end_ptr);

/* ---- the writer object and the two formatters of a result (pandaseq-writer.h:36-141, writer.c; pandaseq.h:293-304, output.c:85-126):
 * what a caller of panda_assembler_next() does with a record, under the reference's names and signatures.  A writer gathers
 * what a thread appends and hands it to its `write` callback, whole, when that thread commits (records of different threads
 * never interleave, as with the reference's per-thread buffers).  panda_writer_open_file() writes plain files, or bzip2 through
 * libbz2.so.1 bound at run time (writer.c:147-151; NULL where that library cannot be loaded).  The batched formatter -- text made on the device for a whole batch -- is pamd_format_device()
 * (pandaseq_amd_io.h); these are the per-record forms. ---- */
typedef struct panda_writer *PandaWriter;	/* pandaseq-common.h:117 */
typedef void (*PandaBufferWrite) (const char *buffer, size_t buffer_length, void *data);	/* pandaseq-common.h:449-453 */
PandaWriter panda_writer_new(PandaBufferWrite write, void *write_data, PandaDestroy write_destroy);
PandaWriter panda_writer_new_file(FILE *file);
PandaWriter panda_writer_new_null(void);
PandaWriter panda_writer_new_stderr(void);
PandaWriter panda_writer_new_stdout(void);
PandaWriter panda_writer_open_file(const char *filename, bool bzip);
void panda_writer_append(PandaWriter writer, const char *format, ...);
void panda_writer_append_c(PandaWriter writer, char c);
void panda_writer_append_id(PandaWriter writer, const panda_seq_identifier *id);
void panda_writer_append_v(PandaWriter writer, const char *format, va_list va);
void panda_writer_commit(PandaWriter writer);
void panda_writer_set_slave(PandaWriter writer, PandaWriter slave);
PandaWriter panda_writer_get_slave(PandaWriter writer);
void panda_writer_flush(PandaWriter writer);
PandaWriter panda_writer_ref(PandaWriter writer);
void panda_writer_unref(PandaWriter writer);
bool panda_output_fasta(const panda_result_seq *sequence, PandaWriter writer);	/* ">id;%f\n<bases>\n", %f of exp(quality); nothing for an empty sequence */
bool panda_output_fastq(const panda_result_seq *sequence, PandaWriter writer);	/* "@id;%f\n<bases>\n+\n<33 + panda_result_phred>\n" */
int panda_api_version(void);		/* misc.c:33-36 */
char const *panda_version(void);	/* misc.c:28-31 */

/* ---- extensions (not in the reference) ---- */
/* Which GPU a new assembler is created on, in this order: the calling thread's own choice
 * (pamd_compat_set_thread_device; a negative value clears it), else the next entry -- round robin over the process -- of
 * the environment variable PAMD_DEVICES ("0,1,2,3": the embedder that builds one assembler per thread, pool.c:140-160,
 * gets them spread over the GPUs without a code change), else the process default (pamd_compat_set_device, 0). */
void pamd_compat_set_device(int device);
void pamd_compat_set_thread_device(int device);
/* the device the next panda_assembler_new() of the calling thread would use (takes a turn of PAMD_DEVICES when that decides) */
int pamd_compat_device_for_new_assembler(void);
/* how many pairs panda_assembler_next() pulls from the source per batch (default 65536); two batches are in flight */
void pamd_compat_set_batch_pairs(size_t pairs);
/* how assemblers created from now on sum quality and pear / rdp_mle scores (pamd_set_exact_sums): 0 (default) = any order
 * with every decision guarded: statuses, overlaps, merged bases and per-base log-p bit-identical to the reference, `quality`
 * and `estimated_overlap_probability` within 1e-9 (diff.c:26-99 compares bases and log-p, never these two); 1 = in the
 * reference's own order, every output bit bit-identical, about a fifth slower.  PAMD_COMPAT_SUMS=exact in the environment
 * makes 1 the default of a process.  An assembler with a `check` module that runs on the host (a caller's callback, a plugin
 * file) sums in the reference's order by itself -- such a callback may compare `quality` or the score with a threshold of its
 * own -- unless any-order sums were asked for in so many words: this function called with 0, or PAMD_COMPAT_SUMS=fast; with
 * min_phred / min_overlapbits BEHIND a host module (their verdict is then made on the host from those sums) always.  The
 * library says so once on stderr when it applies. */
void pamd_compat_set_exact_sums(int exact);
/* Where panda_assembler_assemble() (pandaseq-assembler.h:137-163, assembler.c:368-383) assembles its ONE pair: 0 (default) = on
 * the caller's thread, through the algorithm class's overlap_probability / match_probability as the reference does (sums in the
 * reference's order: every output bit-identical); 1 = on the device like a batch of one (an upload, two launches and a download
 * per call: for checking the two against each other).  PAMD_COMPAT_PER_PAIR=device makes 1 the default of a process.  Arrays
 * (pamd_compat_assemble_many) and streams (panda_assembler_next) of pairs always go to the device. */
void pamd_compat_set_per_pair_device(int device);
/* The array form of panda_assembler_assemble() (pandaseq-assembler.h:137-163): n pairs through the batches of
 * panda_assembler_next() (up to 65536 pairs each, four in flight) instead of one at a time -- what a caller that already holds its pairs wants
 * (panda_diff's A/B loop, diff.c:26-99: INTEGRATION.md section 2).  `visit` is called for every pair, in order, with its record
 * (NULL when the pair did not assemble; valid during the call only, as panda_assembler_assemble's is until the next call;
 * since round 4 its `forward` / `reverse` point at the library's staged copies of the reads, not at the caller's arrays -- same
 * bytes, same lifetime as the record);
 * counters, modules and the fail-alignment handler behave as if the pairs had been passed one by one.  Returns how many
 * pairs assembled.  An array of at most a few thousand pairs is one piece in one slot; longer ones use up to four slots, each
 * with a device assembler of its own, made the first time they are needed. */
typedef void (*PamdVisit) (size_t index, const panda_result_seq *result, void *user_data);
size_t pamd_compat_assemble_many(PandaAssembler assembler, const panda_seq_identifier *ids, const panda_qual *const *forward, const size_t *forward_length, const panda_qual *const *reverse, const size_t *reverse_length, size_t n, PamdVisit visit, void *user_data);
/* where panda_assembler_next() builds a record's panda_result[] rows: 0 (default) = helper threads, batch by batch, ahead of
 * the caller (a consumer that reads only a record's scalars never pays for the rows); 1 = the caller's thread, when the record
 * is handed over, into one buffer that stays in its cache (faster for a consumer that reads every base, as output.c:85-126
 * does).  PAMD_COMPAT_ROWS=lazy in the environment selects 1. */
void pamd_compat_set_lazy_rows(int lazy);

#ifdef __cplusplus
}
#endif
#endif
