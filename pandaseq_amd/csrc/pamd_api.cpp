// pamd_api.cpp -- the C ABI declared in include/pandaseq_amd.h.
//
// Host side of the hot path: owns the assembler configuration (the fields of struct
// panda_assembler that align()/assemble_seq() read, assembler.h:44-57,72-78), keeps the
// probability tables and primers resident in HBM, sizes the persistent launch and its LDS
// carve-up for each batch, and enqueues the kernel.  There is no CPU implementation behind
// these entry points: without a GPU they fail.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "pamd_internal.h"

namespace {

thread_local char g_error[512] = "";
uint8_t *g_debug_buf = nullptr;	// development aid, see pamd_debug_fetch()

#define HIP_TRY(call)                                                                       \
	do {                                                                                \
		hipError_t e_ = (call);                                                     \
		if (e_ != hipSuccess) {                                                     \
			pamd_set_error("%s: %s", #call, hipGetErrorString(e_));             \
			return PAMD_ENODEVICE;                                              \
		}                                                                           \
	} while (0)

constexpr size_t LDS_LIMIT = 160 * 1024;	// per CU on gfx950
constexpr int MAX_WAVES_PER_WG = 16;

inline uint32_t up16(uint32_t v) {
	return (v + 15u) & ~15u;
}

}	// namespace

void pamd_set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof g_error, fmt, ap);
	va_end(ap);
}

struct pamd_assembler {
	int device = 0;
	int cu_count = 0;
	// configuration
	int algo = PAMD_ALGO_SIMPLE_BAYES;
	double sb_q = 0.36, sb_pmatch = 0, sb_pmismatch = 0;
	double up_q = 0.36, up_pmatch = 0, up_pmismatch = 0;	// uparse (algo_uparse.c:126-135)
	double pear_random_base = 0;
	double threshold = 0;
	int minoverlap = 2;
	int maxoverlap = 0;
	uint8_t primers[912];
	int fprimer_len = 0, rprimer_len = 0;
	int ftrim = 0, rtrim = 0;
	double primer_penalty = 0;
	int n_modules = 0;
	PamdModule modules[PAMD_MAX_MODULES];
	int exact_sums = 1;
	int post_primers = 0;
	int fast_path = 1;		// ordinary pairs go through the lean kernel when the configuration allows (pamd_fast.hip)
	int two_pairs = 0;		// ... two pairs per wave where reads are short (pamd_fast_dual.h): built, parity-green and 22 % SLOWER than a pair per wave
					// on config 4 (profiles/r06_two_pairs_per_wave_ab.txt): off unless pamd_set_two_pairs_per_wave() or PAMD_DUAL=1 asks for it
	double *d_count_table = nullptr;	// ea_util / stitch: every value an overlap's score can take (PamdKernelParams::count_table) ...
	int count_table_algo = -1;	// ... built for this scorer
	uint32_t *d_primer_off = nullptr;	// `-p` / `-q`: the primer kernel's word per pair ...
	size_t primer_cap = 0;	// ... and how many pairs it has room for
	uint32_t *d_defer = nullptr;	// the pairs it leaves to the general kernel ...
	size_t defer_cap = 0;
	unsigned *d_defer_count = nullptr;	// ... and how many; the word behind it: how many of them the general kernel's waves have taken
	bool last_two_kernels = false;
	int last_variant = 0;	// pamd_last_lean_variant()
	double *d_notp = nullptr;
	std::vector<double> h_image;	// host copy of the table image on the device (pamd_table_image)
	uint16_t *d_slots = nullptr;	// log-p -> index of the image (pamd_encode.hip)
	// tables
	PamdTables tabs;
	double *d_tables = nullptr;	// image for the current algorithm
	int d_tables_algo = -1;
	uint8_t *d_primers = nullptr;
	bool primers_dirty = true;
	unsigned long long *d_error = nullptr;
	bool notp_never_finite = false;	// of the table image on the device (upload_tables)
	unsigned long long *d_phase = nullptr;
	void *hb_ptr[9] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };	// pamd_assemble_host staging
	size_t hb_cap[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	void *small_h = nullptr, *small_d = nullptr;	// pamd_assemble_host, a few pairs: one pinned block and its device twin
	size_t small_cap = 0;
	double *d_scratch = nullptr;	// parked score terms (pear / rdp_mle), sized for the largest launch so far
	void *d_fast_scratch = nullptr;	// the lean kernel's rows of table indices (sums in the reference's order), likewise
	size_t fast_scratch_bytes = 0;
	size_t scratch_rows = 0;	// 10 words, only written by -DPAMD_PHASE_TIMING builds
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	bool timed = false;
	hipStream_t last_stream = nullptr;	// the stream of the launch ev_stop was recorded on
	int last_wgs = 0, last_waves = 0, last_lds = 0;
};

extern "C" {

const char *pamd_last_error(void) {
	return g_error;
}

int pamd_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int pamd_tables_get(double *out, size_t n_doubles) {
	if (out == nullptr || n_doubles < PAMD_TABLE_DOUBLES) {
		pamd_set_error("pamd_tables_get: need room for %d doubles", PAMD_TABLE_DOUBLES);
		return PAMD_EINVAL;
	}
	PamdTables *t = new (std::nothrow) PamdTables;
	if (t == nullptr)
		return PAMD_ENOMEM;
	pamd_build_tables(t);
	double *o = out;
	*o++ = t->qual_nn;
	memcpy(o, t->score, sizeof t->score);
	o += PAMD_NQ;
	memcpy(o, t->score_err, sizeof t->score_err);
	o += PAMD_NQ;
	const double *mats[8] = { t->match_sb, t->mismatch_sb, t->match_pear, t->mismatch_pear, t->mismatch_rdp, t->mismatch_rdp_asm, t->match_uparse, t->mismatch_uparse };
	for (int m = 0; m < 8; m++) {
		memcpy(o, mats[m], sizeof(double) * PAMD_TAB);
		o += PAMD_TAB;
	}
	delete t;
	return PAMD_OK;
}

static void set_uparse_q(pamd_assembler *a, double q) {	// algo_uparse.c:126-135
	if (q > 0 && q < 1) {
		a->up_q = q;
		a->up_pmatch = std::log(1 - q * q * (1 - 2 * q + 4 * q * q / 3));
		a->up_pmismatch = std::log(1 - 4 * q / 3 / (2 * q - 4 * q * q / 3));
	}
}

static void set_sb_q(pamd_assembler *a, double q) {	// algo_simple_bayes.c:126-135
	if (q > 0 && q < 1) {
		a->sb_q = q;
		a->sb_pmatch = std::log(0.25 * (1 - 2 * q + q * q));
		a->sb_pmismatch = std::log((3 * q - 2 * q * q) / 18.0);
	}
}

pamd_assembler *pamd_assembler_create(int device) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		pamd_set_error("no HIP device available (%s); the overlap assembler has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
		return nullptr;
	}
	if (device < 0 || device >= n) {
		pamd_set_error("device %d out of range (0..%d)", device, n - 1);
		return nullptr;
	}
	pamd_assembler *a = new (std::nothrow) pamd_assembler;
	if (a == nullptr)
		return nullptr;
	a->device = device;
	hipDeviceProp_t prop;
	if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
		pamd_set_error("cannot query device %d", device);
		delete a;
		return nullptr;
	}
	a->cu_count = prop.multiProcessorCount;
	pamd_build_tables(&a->tabs);
	set_sb_q(a, 0.36);	// algo_simple_bayes.c:113
	set_uparse_q(a, 0.36);	// algo_uparse.c:103
	a->pear_random_base = std::log(0.25);	// algo_pear.c:106
	a->threshold = std::log(0.6);	// assembler_support.c:76
	memset(a->primers, 0, sizeof a->primers);
	if (hipMalloc(&a->d_tables, sizeof(double) * (PAMD_LT_C + PAMD_TAB)) != hipSuccess || hipMalloc(&a->d_notp, sizeof(double) * (PAMD_LT_C + PAMD_TAB)) != hipSuccess || hipMalloc(&a->d_primers, sizeof a->primers) != hipSuccess || hipMalloc(&a->d_slots, PAMD_ENCODE_SLOTS * sizeof(uint16_t)) != hipSuccess || hipMalloc(&a->d_error, sizeof(unsigned long long)) != hipSuccess || hipMalloc(&a->d_defer_count, 2 * sizeof(unsigned)) != hipSuccess || hipMemset(a->d_defer_count, 0, 2 * sizeof(unsigned)) != hipSuccess || hipMalloc(&a->d_phase, 10 * sizeof(unsigned long long)) != hipSuccess || hipMemset(a->d_phase, 0, 10 * sizeof(unsigned long long)) != hipSuccess || hipMemset(a->d_error, 0, sizeof(unsigned long long)) != hipSuccess || hipEventCreate(&a->ev_start) != hipSuccess || hipEventCreate(&a->ev_stop) != hipSuccess) {
		pamd_set_error("device allocation failed on device %d", device);
		pamd_assembler_destroy(a);
		return nullptr;
	}
	return a;
}

void pamd_assembler_destroy(pamd_assembler *a) {
	if (a == nullptr)
		return;
	(void) hipSetDevice(a->device);
	if (a->d_tables)
		(void) hipFree(a->d_tables);
	if (a->d_notp)
		(void) hipFree(a->d_notp);
	if (a->d_primers)
		(void) hipFree(a->d_primers);
	if (a->d_slots)
		(void) hipFree(a->d_slots);
	if (a->d_error)
		(void) hipFree(a->d_error);
	if (a->d_phase)
		(void) hipFree(a->d_phase);
	if (a->d_scratch)
		(void) hipFree(a->d_scratch);
	if (a->d_fast_scratch)
		(void) hipFree(a->d_fast_scratch);
	if (a->d_defer)
		(void) hipFree(a->d_defer);
	if (a->d_primer_off)
		(void) hipFree(a->d_primer_off);
	if (a->d_count_table)
		(void) hipFree(a->d_count_table);
	if (a->d_defer_count)
		(void) hipFree(a->d_defer_count);
	if (a->small_h)
		(void) hipHostFree(a->small_h);
	if (a->small_d)
		(void) hipFree(a->small_d);
	for (void *q : a->hb_ptr)
		if (q)
			(void) hipFree(q);
	if (a->ev_start)
		(void) hipEventDestroy(a->ev_start);
	if (a->ev_stop)
		(void) hipEventDestroy(a->ev_stop);
	delete a;
}

int pamd_set_algorithm(pamd_assembler *a, int algo, double param) {
	if (a == nullptr || algo < 0 || algo > PAMD_ALGO_STITCH) {
		pamd_set_error("pamd_set_algorithm: unknown algorithm %d", algo);
		return PAMD_EUNSUPPORTED;
	}
	a->algo = algo;
	if (algo == PAMD_ALGO_SIMPLE_BAYES) {
		set_sb_q(a, std::isnan(param) ? 0.36 : param);
	} else if (algo == PAMD_ALGO_PEAR) {
		a->pear_random_base = std::isnan(param) ? std::log(0.25) : param;
	} else if (algo == PAMD_ALGO_UPARSE) {
		set_uparse_q(a, std::isnan(param) ? 0.36 : param);
	}
	return PAMD_OK;
}

int pamd_set_threshold(pamd_assembler *a, double threshold) {	// assembler_support.c:391-397
	if (threshold > 0 && threshold < 1)
		a->threshold = std::log(threshold);
	return PAMD_OK;
}

int pamd_set_log_threshold(pamd_assembler *a, double log_threshold) {	// the value assembler.c:334 compares with, as stored
	if (a == nullptr)
		return PAMD_EINVAL;
	if (log_threshold < 0)
		a->threshold = log_threshold;
	return PAMD_OK;
}

int pamd_set_minimum_overlap(pamd_assembler *a, int overlap) {	// assembler_support.c:276-282
	if (overlap > 1 && overlap < 2 * PAMD_MAX_LEN)
		a->minoverlap = overlap;
	return PAMD_OK;
}

int pamd_set_maximum_overlap(pamd_assembler *a, int overlap) {	// assembler_support.c:289-295
	if (overlap >= 0 && overlap < 2 * PAMD_MAX_LEN)
		a->maxoverlap = overlap;
	return PAMD_OK;
}

int pamd_set_forward_primer(pamd_assembler *a, const uint8_t *nt, size_t length) {	// assembler_support.c:201-213
	if (length < PAMD_MAX_LEN) {
		if (length > 0)
			memcpy(a->primers, nt, length);
		a->fprimer_len = (int) length;
		a->ftrim = 0;
		a->primers_dirty = true;
	}
	return PAMD_OK;
}

int pamd_set_reverse_primer(pamd_assembler *a, const uint8_t *nt, size_t length) {	// assembler_support.c:347-359
	if (length < PAMD_MAX_LEN) {
		if (length > 0)
			memcpy(a->primers + 456, nt, length);
		a->rprimer_len = (int) length;
		a->rtrim = 0;
		a->primers_dirty = true;
	}
	return PAMD_OK;
}

int pamd_set_forward_trim(pamd_assembler *a, size_t trim) {	// assembler_support.c:244-249
	a->ftrim = (int) (trim > 65535 ? 65535 : trim);
	a->fprimer_len = 0;
	return PAMD_OK;
}

int pamd_set_reverse_trim(pamd_assembler *a, size_t trim) {	// assembler_support.c:374-379
	a->rtrim = (int) (trim > 65535 ? 65535 : trim);
	a->rprimer_len = 0;
	return PAMD_OK;
}

int pamd_set_primers_after(pamd_assembler *a, int after) {	// assembler_support.c:355-359
	if (a == nullptr)
		return PAMD_EINVAL;
	a->post_primers = after ? 1 : 0;
	return PAMD_OK;
}

int pamd_set_primer_penalty(pamd_assembler *a, double penalty) {	// assembler_support.c:404-410
	if (penalty >= 0 && penalty < 1)
		a->primer_penalty = penalty;
	return PAMD_OK;
}

int pamd_add_module_min_phred(pamd_assembler *a, int phred) {	// plugin_min_phred.c:24-44
	if (a->n_modules >= PAMD_MAX_MODULES || phred < 0 || phred > 127) {
		pamd_set_error("min_phred: value must be 0..127 and at most %d modules", PAMD_MAX_MODULES);
		return PAMD_EINVAL;
	}
	a->modules[a->n_modules].kind = 1;
	a->modules[a->n_modules].phred = phred;
	a->modules[a->n_modules].nats = 0;
	a->n_modules++;
	return PAMD_OK;
}

int pamd_add_module_min_overlapbits(pamd_assembler *a, double bits) {	// plugin_min_overlapbits.c:25-53
	const double nats = bits * 0.69314718055994530942;	// M_LN2
	if (a->n_modules >= PAMD_MAX_MODULES || !(nats >= 0)) {
		pamd_set_error("min_overlapbits: value out of range or more than %d modules", PAMD_MAX_MODULES);
		return PAMD_EINVAL;
	}
	a->modules[a->n_modules].kind = 2;
	a->modules[a->n_modules].phred = 0;
	a->modules[a->n_modules].nats = nats;
	a->n_modules++;
	return PAMD_OK;
}

int pamd_set_exact_sums(pamd_assembler *a, int exact) {
	a->exact_sums = exact ? 1 : 0;
	return PAMD_OK;
}

int pamd_set_two_pairs_per_wave(pamd_assembler *a, int enable) {
	if (a == nullptr)
		return PAMD_EINVAL;
	a->two_pairs = enable ? 1 : 0;
	return PAMD_OK;
}

int pamd_set_fast_path(pamd_assembler *a, int enable) {
	if (a == nullptr)
		return PAMD_EINVAL;
	a->fast_path = enable ? 1 : 0;
	return PAMD_OK;
}

long long pamd_last_deferred(pamd_assembler *a) {
	if (a == nullptr || !a->timed || !a->last_two_kernels)
		return -1;
	unsigned n = 0;
	if (hipSetDevice(a->device) != hipSuccess || hipEventSynchronize(a->ev_stop) != hipSuccess || hipMemcpy(&n, a->d_defer_count, sizeof n, hipMemcpyDeviceToHost) != hipSuccess)
		return -1;
	return (long long) n;
}

int pamd_last_lean_variant(pamd_assembler *a) {
	return a == nullptr ? 0 : a->last_variant;
}

int pamd_assemble_status(pamd_assembler *a, uint64_t *n_overflowed) {
	if (a == nullptr)
		return PAMD_EINVAL;
	HIP_TRY(hipSetDevice(a->device));
	unsigned long long overflow = 0;
	HIP_TRY(hipMemcpy(&overflow, a->d_error, sizeof overflow, hipMemcpyDeviceToHost));
	if (overflow != 0)
		HIP_TRY(hipMemset(a->d_error, 0, sizeof overflow));
	if (n_overflowed)
		*n_overflowed = overflow;
	if (overflow != 0) {
		pamd_set_error("output arena too small: %llu assembled pairs did not fit in seq_capacity (status PAMD_ST_NOSPACE)", overflow);
		return PAMD_EINVAL;
	}
	return PAMD_OK;
}

int pamd_clear_modules(pamd_assembler *a) {
	a->n_modules = 0;
	return PAMD_OK;
}

static int upload_tables(pamd_assembler *a, hipStream_t st) {
	// what follows overwrites device memory an earlier launch of this assembler reads (tables, count table, primers) with
	// blocking copies on the null stream, which do not wait for a caller's non-blocking stream: wait for that launch first
	const bool will_copy = a->d_tables_algo != a->algo || ((a->algo == PAMD_ALGO_EA_UTIL || a->algo == PAMD_ALGO_STITCH) && a->count_table_algo != a->algo) || a->primers_dirty;
	if (will_copy && a->timed)
		HIP_TRY(hipEventSynchronize(a->ev_stop));
	if (a->d_tables_algo != a->algo) {
		std::vector<double> img(PAMD_LT_C + PAMD_TAB, 0.0);
		memcpy(&img[PAMD_LT_SCORE], a->tabs.score, sizeof a->tabs.score);
		memcpy(&img[PAMD_LT_ERR], a->tabs.score_err, sizeof a->tabs.score_err);
		const double *A, *B, *C = nullptr;
		switch (a->algo) {
		case PAMD_ALGO_PEAR:	// algo_pear.c:52-54,67
			A = a->tabs.match_pear;
			B = a->tabs.mismatch_pear;
			break;
		case PAMD_ALGO_RDP_MLE:	// algo_rdp_mle.c:68-70 (scan), :39 (merge)
			A = a->tabs.match_sb;
			B = a->tabs.mismatch_rdp;
			C = a->tabs.mismatch_rdp_asm;
			break;
		case PAMD_ALGO_UPARSE:	// algo_uparse.c:74
			A = a->tabs.match_uparse;
			B = a->tabs.mismatch_uparse;
			break;
		default:	// algo_simple_bayes.c:74, algo_stitch.c:66 (flash / ea_util only read qual_score)
			A = a->tabs.match_sb;
			B = a->tabs.mismatch_sb;
			break;
		}
		img[PAMD_LT_SCORE + 47] = a->tabs.qual_nn;	// the spare slot of the score block: what `-a` calls entry 47
		memcpy(&img[PAMD_LT_A], A, sizeof(double) * PAMD_TAB);
		memcpy(&img[PAMD_LT_B], B, sizeof(double) * PAMD_TAB);
		if (C)
			memcpy(&img[PAMD_LT_C], C, sizeof(double) * PAMD_TAB);
		// synchronous on purpose: `img` is a stack-lifetime host buffer
		HIP_TRY(hipMemcpy(a->d_tables, img.data(), sizeof(double) * img.size(), hipMemcpyHostToDevice));
		// panda_log1mexp (offset.c:35-38) of every entry, with the host's libm as the reference uses it: what a
		// merged base that disagrees with a primer adds in the `-a` scan (offset.c:114-122)
		std::vector<double> notp(img.size());
		for (size_t i = 0; i < img.size(); i++)
			notp[i] = (img[i] > M_LN2) ? std::log1p(-std::exp(-img[i])) : std::log(-std::expm1(-img[i]));
		HIP_TRY(hipMemcpy(a->d_notp, notp.data(), sizeof(double) * notp.size(), hipMemcpyHostToDevice));
		a->notp_never_finite = true;
		for (size_t i = 0; i < (a->algo == PAMD_ALGO_RDP_MLE ? img.size() : (size_t) PAMD_LT_C); i++)
			if (i != (size_t) PAMD_LT_ZERO && std::isfinite(notp[i]))	// (the spare slot holds 0.0 and is no base's log-p)
				a->notp_never_finite = false;
		// log-p -> index: open addressing over the distinct values of the image (the first index of a value wins)
		const size_t n_image = a->algo == PAMD_ALGO_RDP_MLE ? img.size() : (size_t) PAMD_LT_C;
		a->h_image.assign(img.begin(), img.begin() + n_image);
		std::vector<uint16_t> slots(PAMD_ENCODE_SLOTS, 0xFFFFu);
		for (size_t i = 0; i < n_image; i++) {
			unsigned long long bits;
			memcpy(&bits, &img[i], 8);
			unsigned h = (unsigned) ((bits * 0x9E3779B97F4A7C15ull) >> 50) & (PAMD_ENCODE_SLOTS - 1u);
			unsigned walked = 0;
			while (slots[h] != 0xFFFFu && memcmp(&img[slots[h]], &img[i], 8) != 0) {
				h = (h + 1u) & (PAMD_ENCODE_SLOTS - 1u);
				walked++;
			}
			if (walked >= PAMD_ENCODE_PROBES) {	// the device lookup gives up after that many slots (pamd_encode.hip)
				pamd_set_error("table image: entry %zu lies %u slots from its hash slot (the device walks %d); PAMD_ENCODE_SLOTS is too small for these tables", i, walked, PAMD_ENCODE_PROBES);
				return PAMD_EINVAL;
			}
			if (slots[h] == 0xFFFFu)
				slots[h] = (uint16_t) i;
		}
		HIP_TRY(hipMemcpy(a->d_slots, slots.data(), slots.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		a->d_tables_algo = a->algo;
	}
	if ((a->algo == PAMD_ALGO_EA_UTIL || a->algo == PAMD_ALGO_STITCH) && a->count_table_algo != a->algo) {
		// algo_ea_util.c:55, algo_stitch.c:56 for every argument they can meet, with this host's libm
		std::vector<double> t;
		if (a->algo == PAMD_ALGO_EA_UTIL) {
			t.resize((size_t) PAMD_CT_EA * PAMD_CT_EA);
			for (size_t real_overlap = 0; real_overlap < PAMD_CT_EA; real_overlap++)
				for (size_t mismatches = 0; mismatches < PAMD_CT_EA; mismatches++)
					t[real_overlap * PAMD_CT_EA + mismatches] = log((((double) mismatches) * mismatches + 1) / real_overlap);
		} else {
			t.resize((size_t) PAMD_CT_ST * PAMD_CT_ST);
			for (long long d = -450; d <= 450; d++)
				for (size_t len = 0; len < PAMD_CT_ST; len++) {
					const size_t score = (size_t) d;	// matches - mismatches in size_t: wraps when negative
					t[(size_t) (d + 450) * PAMD_CT_ST + len] = log(score / (double) len);
				}
		}
		if (a->d_count_table)
			HIP_TRY(hipFree(a->d_count_table));
		a->d_count_table = nullptr;
		a->count_table_algo = -1;
		HIP_TRY(hipMalloc(&a->d_count_table, t.size() * sizeof(double)));
		HIP_TRY(hipMemcpy(a->d_count_table, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
		a->count_table_algo = a->algo;
	}
	if (a->primers_dirty) {
		HIP_TRY(hipMemcpy(a->d_primers, a->primers, sizeof a->primers, hipMemcpyHostToDevice));
		a->primers_dirty = false;
	}
	(void) st;
	return PAMD_OK;
}

// `error_flag`: the device word this launch counts its overflowing rows into (the assembler's own, or -- the small-batch path
// of pamd_assemble_host -- a word of the block that comes back with the results)
static int assemble_device_impl(pamd_assembler *a, const pamd_batch *b, void *stream, unsigned long long *error_flag) {
	if (a == nullptr || b == nullptr) {
		pamd_set_error("pamd_assemble_device: null argument");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(a->device));
	if (b->n_pairs == 0)
		return PAMD_OK;
	if (b->fwd == nullptr || b->rev == nullptr || b->flen == nullptr || b->rlen == nullptr || b->results == nullptr || b->seq_nt == nullptr || b->seq_p == nullptr) {
		pamd_set_error("pamd_assemble_device: batch has null buffers");
		return PAMD_EINVAL;
	}
	if (b->fwd_stride % 8 || b->rev_stride % 8 || b->fwd_stride == 0 || b->rev_stride == 0 || b->fwd_stride > 456 || b->rev_stride > 456) {
		pamd_set_error("row strides must be multiples of 8 bases in [8,456] (got %zu, %zu)", b->fwd_stride, b->rev_stride);
		return PAMD_EINVAL;
	}
	if (((uintptr_t) b->fwd | (uintptr_t) b->rev | (uintptr_t) b->seq_nt | (uintptr_t) b->seq_p) & 15u) {
		pamd_set_error("fwd, rev, seq_nt and seq_p must be 16-byte aligned");
		return PAMD_EINVAL;
	}
	if (b->seq_stride % 16) {
		pamd_set_error("seq_stride must be a multiple of 16 (got %zu)", b->seq_stride);
		return PAMD_EINVAL;
	}
	if (b->seq_stride == 0 && b->seq_cursor == nullptr) {
		pamd_set_error("compact output (seq_stride == 0) needs seq_cursor");
		return PAMD_EINVAL;
	}
	hipStream_t st = static_cast<hipStream_t>(stream);
	int rc = upload_tables(a, st);
	if (rc != PAMD_OK)
		return rc;

	PamdKernelParams p;
	memset(&p, 0, sizeof p);
	p.fwd = reinterpret_cast<const uint8_t *>(b->fwd);
	p.rev = reinterpret_cast<const uint8_t *>(b->rev);
	p.flen = b->flen;
	p.rlen = b->rlen;
	p.results = b->results;
	p.seq_nt = b->seq_nt;
	p.seq_p = b->seq_p;
	p.seq_cursor = reinterpret_cast<unsigned long long *>(b->seq_cursor);
	p.counters = reinterpret_cast<unsigned long long *>(b->counters);
	p.error_flag = error_flag;
	p.phase_cycles = a->d_phase;
	if (const char *dp = getenv("PAMD_DEBUG_PAIR")) {	// -DPAMD_FK_DEBUG builds: dump the lean kernel's LDS for one pair
		if (g_debug_buf == nullptr && hipMalloc(&g_debug_buf, 4 * 8192) != hipSuccess)
			g_debug_buf = nullptr;
		p.debug_buf = g_debug_buf;
		p.debug_pair = strtoull(dp, nullptr, 10);
	}
	p.tables = a->d_tables;
	p.count_table = a->d_count_table;
	p.notp = a->d_notp;
	p.post_primers = a->post_primers;
	p.primers = a->d_primers;
	p.n_pairs = b->n_pairs;
	// Every wave that merges a pair reserves a chunk of the compact arena (one global atomic per chunk), so a launch
	// can leave up to waves x chunk bases unused: 16384 per wave for large batches, less for small ones and for
	// the general kernel's pass over the lean kernel's deferred pairs (a few per wave).
	// (Measured at 50 M pairs, config 1: 8192, 16384 and 32768 give the same rate; 4096 and 65536 lose 3-5 %.)
	p.arena_chunk = b->n_pairs >= 262144 ? 16384 : 2048;
	p.seq_stride = b->seq_stride;
	p.seq_capacity = b->seq_capacity;
	p.fwd_row_bytes = (uint32_t) b->fwd_stride * 2u;
	p.rev_row_bytes = (uint32_t) b->rev_stride * 2u;
	p.algo = a->algo;
	p.minoverlap = a->minoverlap;
	p.maxoverlap = a->maxoverlap;
	p.fprimer_len = a->fprimer_len;
	p.rprimer_len = a->rprimer_len;
	p.ftrim = a->ftrim;
	p.rtrim = a->rtrim;
	p.n_modules = a->n_modules;
	p.threshold = a->threshold;
	p.primer_penalty = a->primer_penalty;
	p.qual_nn = a->tabs.qual_nn;
	p.sb_pmatch = a->algo == PAMD_ALGO_UPARSE ? a->up_pmatch : a->sb_pmatch;
	p.sb_pmismatch = a->algo == PAMD_ALGO_UPARSE ? a->up_pmismatch : a->sb_pmismatch;
	p.pear_random_base = a->pear_random_base;
	p.exact_sums = a->exact_sums;
	p.two_pairs = (a->two_pairs || (getenv("PAMD_DUAL") && getenv("PAMD_DUAL")[0] == '1')) ? 1 : 0;	// (the environment: measurements only)
	if (a->algo == PAMD_ALGO_PEAR || a->algo == PAMD_ALGO_RDP_MLE) {
		// upper bounds of one term of an overlap's score (PamdKernelParams::many_*): algo_pear.c:47-55 adds -random_base
		// for an N and a table entry otherwise, algo_rdp_mle.c:60-70 a table entry minus qual_nn
		const bool pear = a->algo == PAMD_ALGO_PEAR;
		const double *tm = pear ? a->tabs.match_pear : a->tabs.match_sb, *tx = pear ? a->tabs.mismatch_pear : a->tabs.mismatch_rdp;
		const double off = pear ? 0.0 : a->tabs.qual_nn;
		double mm = -INFINITY, mx = -INFINITY, ms = -INFINITY;
		for (int qa = 0; qa < PAMD_NQ; qa++)
			for (int qb = 0; qb < PAMD_NQ; qb++) {
				const double vm = tm[qa * PAMD_NQ + qb] - off, vx = tx[qa * PAMD_NQ + qb] - off;
				mm = vm > mm ? vm : mm;
				mx = vx > mx ? vx : mx;
				if (qa >= PAMD_STRONG_Q && qb >= PAMD_STRONG_Q)
					ms = vx > ms ? vx : ms;
			}
		p.many_match = mm;
		p.many_strong = ms;
		p.many_all = mm > mx ? mm : mx;
		if (pear && -a->pear_random_base > p.many_all)
			p.many_all = -a->pear_random_base;
	}
	// the lean kernel's primer scan (pamd_fast.hip): the primers as 2-bit codes of the reads' code streams, and what a
	// disagreement costs at least
	for (int which = 0; which < 3; which++) {
		const int plen = which == 0 ? a->fprimer_len : a->rprimer_len;
		const uint8_t *pr = a->primers + (which == 0 ? 0 : 456);
		for (int w = 0; w < 2; w++) {
			uint32_t code = 0, counts = 0;
			for (int k = 0; k < 16; k++) {
				const int x = 16 * w + k;
				const unsigned nt = x < plen ? (pr[which == 2 ? plen - 1 - x : x] & 0xFu) : 15u;	// [2]: the reverse primer read backwards
				const unsigned c = nt == 8 ? 3u : (nt == 4 ? 2u : (nt == 2 ? 1u : 0u));	// misc.h:41
				const unsigned stored = which != 1 ? c : (((c & 1u) << 1) | (c >> 1));	// the reverse read's stream swaps the two bits
				if (nt == 1 || nt == 2 || nt == 4 || nt == 8) {
					code |= stored << (2 * k);
					counts |= 2u << (2 * k);
				}
			}
			p.primer_codes[which].code[w] = code;
			p.primer_codes[which].counts[w] = counts;
		}
	}
	p.notp_never_finite = a->notp_never_finite ? 1 : 0;
	p.primer_qthr = 22;
	p.primer_err_t = a->tabs.score_err[22];
	p.primer_inv_err_t = 1.0 / p.primer_err_t;
	memcpy(p.modules, a->modules, sizeof p.modules);
	for (int m = 0; m < a->n_modules; m++)
		if (a->modules[m].kind == 1)
			p.has_min_phred = 1;

	// LDS carve-up (docs/NOTEBOOK.md 3.2, "LDS carve-up")
	const uint32_t smax = (uint32_t) (b->fwd_stride > b->rev_stride ? b->fwd_stride : b->rev_stride);
	uint32_t mmax = 2 * smax;
	if (mmax > 2 * PAMD_MAX_LEN)
		mmax = 2 * PAMD_MAX_LEN;
	mmax = up16(mmax);
	p.smax = smax;
	p.mmax = mmax;
	p.lds_table_doubles = (a->algo == PAMD_ALGO_RDP_MLE) ? (PAMD_LT_C + PAMD_TAB) : PAMD_LT_C;
	p.lds_ctr_off = up16(p.lds_table_doubles * 8u);
	p.lds_primer_off = up16(p.lds_ctr_off + PAMD_NUM_COUNTERS * 4u);
	p.lds_wave_off = up16(p.lds_primer_off + 912u);
	uint32_t o = 0;
	p.w_fwd = o;
	o += smax * 2u;
	p.w_rev = o;
	o += smax * 2u;
	p.w_fc = o;
	o += 128;
	p.w_rc = o;
	o += 128;
	p.w_fn = o;
	o += 128;
	p.w_rn = o;
	o += 128 + 16;	// the scan may read one word past a stream
	p.w_bits = o;
	o += 128;
	p.w_pend = o;
	if (a->exact_sums && !a->post_primers)	// parked records of 80 bytes: only the exact-sums kernels park pairs in LDS (finish_group);
		o += ((a->algo == PAMD_ALGO_PEAR || a->algo == PAMD_ALGO_RDP_MLE) ? 8 : 16) * 80;	// group_size<ALGO>() of pamd_kernels.hip
	// the merge loops are branch-free: up to 63 lanes past the row's end write into slack (pamd_kernels.hip)
	p.w_nt = o;
	o += mmax + 64u;
	// w_p: the k-mer table while seeding; then the pear / rdp_mle per-base terms of a candidate (up to mmax of them)
	// or a u16 candidate list; with `-a` also the merged row's log-p (+ slack), which otherwise goes straight to HBM
	{
		const uint32_t row = (mmax * 8u > 2048u ? mmax * 8u : 2048u);
		uint32_t slots = 512, shift = 23;
		while (slots * 2u * 4u <= row && slots < 2048u) {
			slots *= 2u;
			shift--;
		}
		p.hash_slots = slots;
		p.hash_shift = shift;
		const bool counting = a->algo != PAMD_ALGO_PEAR && a->algo != PAMD_ALGO_RDP_MLE;
		uint32_t region = slots * 4u > 2048u ? slots * 4u : 2048u;
		if (a->post_primers)
			region = row + 64u * 8u;
		else if (!counting)
			region = row;
		p.w_p = o;
		o += region;
	}
	p.w_pidx = o;
	if (a->post_primers)
		o += mmax * 2u + 64u * 2u;
	p.lds_wave_bytes = up16(o);

	int waves = (int) ((LDS_LIMIT - p.lds_wave_off) / p.lds_wave_bytes);
	if (waves > MAX_WAVES_PER_WG)
		waves = MAX_WAVES_PER_WG;
	if (const char *cap = getenv("PAMD_MAX_WAVES")) {	// occupancy experiments only
		const int c = atoi(cap);
		if (c >= 1 && c < waves)
			waves = c;
	}
	if (waves < 1) {
		pamd_set_error("batch does not fit in LDS");
		return PAMD_EINVAL;
	}
	size_t want_wgs = (b->n_pairs + (size_t) waves - 1) / (size_t) waves;
	int wgs = a->cu_count > 0 ? a->cu_count : 256;
	if ((size_t) wgs > want_wgs)
		wgs = (int) want_wgs;
	const size_t lds_bytes = p.lds_wave_off + (size_t) waves * p.lds_wave_bytes;
	a->last_wgs = wgs;
	a->last_waves = waves;
	a->last_lds = (int) lds_bytes;

	if ((a->algo == PAMD_ALGO_PEAR || a->algo == PAMD_ALGO_RDP_MLE) && a->exact_sums) {
		const size_t rows = (size_t) wgs * (size_t) waves * 8u;	// group_size<pear / rdp_mle>() parked rows per wave
		if (rows > a->scratch_rows) {
			if (a->d_scratch)
				HIP_TRY(hipFree(a->d_scratch));
			a->d_scratch = nullptr;
			a->scratch_rows = 0;
			HIP_TRY(hipMalloc(&a->d_scratch, rows * 456u * sizeof(double)));
			a->scratch_rows = rows;
		}
		p.score_scratch = a->d_scratch;
	}
	// Two kernels (pamd_fast.hip): the lean one takes every pair and hands the unusual ones, by index, to the general
	// one, which runs second on the same stream and reads how many there are from device memory.
	const char *nofast = getenv("PAMD_NO_FAST");	// development aid
	const int variant = (a->fast_path && !(nofast && nofast[0] == '1')) ? pamd_fast_supports(p) : 0;
	const bool two = variant != 0;
	if (two && a->defer_cap < b->n_pairs) {
		if (a->d_defer)
			HIP_TRY(hipFree(a->d_defer));
		a->d_defer = nullptr;
		a->defer_cap = 0;
		HIP_TRY(hipMalloc(&a->d_defer, b->n_pairs * sizeof(uint32_t)));
		a->defer_cap = b->n_pairs;
	}
	// `-p` / `-q`: the primers are found by a kernel of their own, first (pamd_primer.hip)
	const bool primers_first = !a->post_primers && (a->fprimer_len > 0 || a->rprimer_len > 0);
	if (primers_first && a->primer_cap < b->n_pairs) {
		if (a->d_primer_off)
			HIP_TRY(hipFree(a->d_primer_off));
		a->d_primer_off = nullptr;
		a->primer_cap = 0;
		HIP_TRY(hipMalloc(&a->d_primer_off, (2 * b->n_pairs + 4) * sizeof(uint32_t)));	// a word per pair, the list of pairs to redo, its count, the work counter
		a->primer_cap = b->n_pairs;
	}
	p.primer_offsets = primers_first ? a->d_primer_off : nullptr;
	// the general kernel's work queue (a counter)
	p.pair_cursor = b->n_pairs <= 0xFF000000ull ? a->d_defer_count + 1 : nullptr;	// (a 32-bit counter that every wave overshoots by up to sixteen; beyond it fixed shares)
	// One launch in flight per assembler: a launch owns the assembler's deferred list, work counters, per-wave records and
	// primer words, and zeroes them on its stream.  A launch on another stream than the previous one waits for it.
	if (a->timed && a->last_stream != st)
		HIP_TRY(hipStreamWaitEvent(st, a->ev_stop, 0));
	HIP_TRY(hipEventRecord(a->ev_start, st));
	HIP_TRY(hipMemsetAsync(a->d_defer_count, 0, 2 * sizeof(unsigned), st));
	// (the lean kernel finds the primers of the pairs it keeps itself; the primer kernel then serves the pairs it deferred)
	const bool lean_primers = two && variant == PAMD_FAST_PRIM;
	if (primers_first && !lean_primers) {
		rc = pamd_launch_primers(p, a->cu_count, stream, nullptr, nullptr);
		if (rc != PAMD_OK)
			return rc;
	}
	if (two) {

		PamdKernelParams pf = p;
		pf.defer_list = a->d_defer;
		pf.defer_count = a->d_defer_count;
		int fwaves = 0, fwgs = 0;
		size_t flds = 0;
		int cap = MAX_WAVES_PER_WG;
		if (getenv("PAMD_MAX_WAVES") && waves < cap)	// occupancy experiments only
			cap = waves;
		pamd_fast_geometry(pf, a->cu_count, LDS_LIMIT, cap, &fwgs, &fwaves, &flds);
		if (const size_t want = pamd_fast_scratch_bytes(pf, fwgs, fwaves)) {	// (a buffer of its own: the general kernel's second launch may use d_scratch)
			if (want > a->fast_scratch_bytes) {
				if (a->d_fast_scratch)
					HIP_TRY(hipFree(a->d_fast_scratch));
				a->d_fast_scratch = nullptr;
				a->fast_scratch_bytes = 0;
				HIP_TRY(hipMalloc(&a->d_fast_scratch, want));
				a->fast_scratch_bytes = want;
			}
			pf.score_scratch = static_cast<double *>(a->d_fast_scratch);
		}
		rc = pamd_launch_fast(pf, fwgs, fwaves, flds, stream);
		if (rc != PAMD_OK)
			return rc;
		a->last_wgs = fwgs;
		a->last_waves = fwaves;
		a->last_lds = (int) flds;
		p.pair_list = a->d_defer;
		p.pair_count = a->d_defer_count;
		p.arena_chunk = b->n_pairs >= 262144 ? 4096 : 1024;
		if (primers_first && lean_primers) {
			rc = pamd_launch_primers(p, a->cu_count, stream, a->d_defer, a->d_defer_count);
			if (rc != PAMD_OK)
				return rc;
		}
	}
	a->last_two_kernels = two;
	a->last_variant = two ? (variant | ((variant == PAMD_FAST_SHORT && smax <= 192) ? PAMD_LEAN_NARROW : 0) | (pamd_fast_dual(p) ? PAMD_LEAN_DUAL : 0)) : 0;
	rc = pamd_launch_assemble(p, wgs, waves, lds_bytes, stream);
	if (rc != PAMD_OK)
		return rc;
	HIP_TRY(hipEventRecord(a->ev_stop, st));
	a->timed = true;
	a->last_stream = st;
	return PAMD_OK;
}

int pamd_assemble_device(pamd_assembler *a, const pamd_batch *b, void *stream) {
	return assemble_device_impl(a, b, stream, a != nullptr ? a->d_error : nullptr);
}

int pamd_assemble_device_range(pamd_assembler *a, const pamd_batch *b, size_t first, size_t count, void *stream) {
	if (a == nullptr || b == nullptr || first > b->n_pairs || count > b->n_pairs - first) {
		pamd_set_error("pamd_assemble_device_range: range outside the batch");
		return PAMD_EINVAL;
	}
	pamd_batch r = *b;
	r.n_pairs = count;
	r.fwd = b->fwd + first * b->fwd_stride;
	r.rev = b->rev + first * b->rev_stride;
	r.flen = b->flen + first;
	r.rlen = b->rlen + first;
	r.results = b->results + first;
	if (b->seq_stride != 0) {	// strided rows: pair i of the batch keeps its row
		r.seq_nt = b->seq_nt + first * b->seq_stride;
		r.seq_p = b->seq_p + first * b->seq_stride;
	}
	return pamd_assemble_device(a, &r, stream);
}

int pamd_encode_rows_device(pamd_assembler *a, const double *seq_p, size_t n_bases, uint16_t *seq_idx, void *stream) {
	if (a == nullptr || seq_p == nullptr || seq_idx == nullptr || ((uintptr_t) seq_p & 15u) || ((uintptr_t) seq_idx & 3u)) {
		pamd_set_error("pamd_encode_rows_device: null or misaligned argument");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(a->device));
	int rc = upload_tables(a, static_cast<hipStream_t>(stream));
	if (rc != PAMD_OK)
		return rc;
	return pamd_launch_encode(seq_p, seq_idx, n_bases, a->d_tables, (unsigned) a->h_image.size(), a->d_slots, stream);
}

int pamd_table_image(pamd_assembler *a, double *out, size_t n_doubles, size_t *used) {
	if (a == nullptr || out == nullptr) {
		pamd_set_error("pamd_table_image: null argument");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(a->device));
	int rc = upload_tables(a, nullptr);
	if (rc != PAMD_OK)
		return rc;
	if (n_doubles < a->h_image.size()) {
		pamd_set_error("pamd_table_image: need room for %zu doubles", a->h_image.size());
		return PAMD_EINVAL;
	}
	memcpy(out, a->h_image.data(), a->h_image.size() * sizeof(double));
	if (used)
		*used = a->h_image.size();
	return PAMD_OK;
}

float pamd_last_kernel_ms(pamd_assembler *a) {
	if (a == nullptr || !a->timed)
		return -1.0f;
	float ms = -1.0f;
	if (hipEventSynchronize(a->ev_stop) != hipSuccess)
		return -1.0f;
	if (hipEventElapsedTime(&ms, a->ev_start, a->ev_stop) != hipSuccess)
		return -1.0f;
	return ms;
}

int pamd_debug_fetch(void *dst, size_t bytes) {	// development aid (-DPAMD_FK_DEBUG builds): not declared in the public header
	if (g_debug_buf == nullptr || bytes > 4 * 8192)
		return PAMD_EINVAL;
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(dst, g_debug_buf, bytes, hipMemcpyDeviceToHost));
	return PAMD_OK;
}

int pamd_debug_phase_cycles(pamd_assembler *a, unsigned long long *out10, int reset) {
	if (a == nullptr || out10 == nullptr)
		return PAMD_EINVAL;
	HIP_TRY(hipSetDevice(a->device));
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(out10, a->d_phase, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	if (reset)
		HIP_TRY(hipMemset(a->d_phase, 0, 10 * sizeof(unsigned long long)));
	return PAMD_OK;
}

int pamd_last_launch_info(pamd_assembler *a, int *workgroups, int *waves_per_workgroup, int *lds_bytes) {
	if (a == nullptr)
		return PAMD_EINVAL;
	if (workgroups)
		*workgroups = a->last_wgs;
	if (waves_per_workgroup)
		*waves_per_workgroup = a->last_waves;
	if (lds_bytes)
		*lds_bytes = a->last_lds;
	return PAMD_OK;
}

// Device-side staging buffers of the host entry point: grow-only, kept across calls (a hipMalloc /
// hipFree pair per call cost more than the kernel for small batches).
static int host_buf(pamd_assembler *a, int which, size_t bytes, void **out) {
	if (a->hb_cap[which] < bytes) {
		if (a->hb_ptr[which])
			HIP_TRY(hipFree(a->hb_ptr[which]));
		a->hb_ptr[which] = nullptr;
		a->hb_cap[which] = 0;
		const size_t want = bytes + bytes / 4 + 256;
		HIP_TRY(hipMalloc(&a->hb_ptr[which], want));
		a->hb_cap[which] = want;
	}
	*out = a->hb_ptr[which];
	return PAMD_OK;
}

int pamd_assemble_host(pamd_assembler *a, const pamd_batch *hb) {
	if (a == nullptr || hb == nullptr) {
		pamd_set_error("pamd_assemble_host: null argument");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(a->device));
	const size_t n = hb->n_pairs;
	if (n == 0)
		return PAMD_OK;
	const size_t cap = hb->seq_stride ? hb->seq_stride * n : hb->seq_capacity;
	pamd_batch d = *hb;
	void *dfwd, *drev, *dfl, *drl, *dres, *dnt, *dp, *dcur, *dctr;
	int rc;
	// A few pairs (panda_assembler_assemble() sends one: what panda_diff calls per pair, diff.c:26-99): the cost is the number
	// of runtime calls, not the bytes.  Inputs and the zeroed cursor / counters go up in ONE copy from a pinned block laid out
	// like its device twin, results, rows, cursor, counters and the overflow word come back in ONE, and the host waits once.
	{
		auto up16 = [](size_t v) { return (v + 15) & ~(size_t) 15; };
		const size_t o_fwd = 0, o_rev = o_fwd + up16(n * hb->fwd_stride * 2), o_fl = o_rev + up16(n * hb->rev_stride * 2), o_rl = o_fl + up16(n * 2);
		const size_t o_cur = o_rl + up16(n * 2), o_ctr = o_cur + 16, o_err = o_ctr + up16(PAMD_NUM_COUNTERS * 8), o_res = o_err + 16;
		const size_t o_nt = o_res + up16(n * sizeof(pamd_pair_result)), o_p = o_nt + up16(cap + 16), total = o_p + up16((cap + 16) * sizeof(double));
		if (total <= ((size_t) 1 << 20)) {
			if (a->small_cap < total) {
				if (a->small_h)
					HIP_TRY(hipHostFree(a->small_h));
				if (a->small_d)
					HIP_TRY(hipFree(a->small_d));
				a->small_h = a->small_d = nullptr;
				a->small_cap = 0;
				const size_t want = total + total / 2 + 4096;
				HIP_TRY(hipHostMalloc(&a->small_h, want, hipHostMallocDefault));
				HIP_TRY(hipMalloc(&a->small_d, want));
				a->small_cap = want;
			}
			char *h = static_cast<char *>(a->small_h), *dv = static_cast<char *>(a->small_d);
			memcpy(h + o_fwd, hb->fwd, n * hb->fwd_stride * 2);
			memcpy(h + o_rev, hb->rev, n * hb->rev_stride * 2);
			memcpy(h + o_fl, hb->flen, n * 2);
			memcpy(h + o_rl, hb->rlen, n * 2);
			memset(h + o_cur, 0, o_res - o_cur);	// cursor, counters, overflow word
			HIP_TRY(hipMemcpyAsync(dv, h, o_res, hipMemcpyHostToDevice, nullptr));
			d.fwd = reinterpret_cast<const pamd_qual *>(dv + o_fwd);
			d.rev = reinterpret_cast<const pamd_qual *>(dv + o_rev);
			d.flen = reinterpret_cast<const uint16_t *>(dv + o_fl);
			d.rlen = reinterpret_cast<const uint16_t *>(dv + o_rl);
			d.results = reinterpret_cast<pamd_pair_result *>(dv + o_res);
			d.seq_nt = reinterpret_cast<uint8_t *>(dv + o_nt);
			d.seq_p = reinterpret_cast<double *>(dv + o_p);
			d.seq_cursor = reinterpret_cast<uint64_t *>(dv + o_cur);
			d.counters = reinterpret_cast<int64_t *>(dv + o_ctr);
			d.seq_capacity = cap;
			rc = assemble_device_impl(a, &d, nullptr, reinterpret_cast<unsigned long long *>(dv + o_err));	// (this launch reports an overflow into the block)
			if (rc != PAMD_OK)
				return rc;
			HIP_TRY(hipMemcpyAsync(h + o_cur, dv + o_cur, total - o_cur, hipMemcpyDeviceToHost, nullptr));
			HIP_TRY(hipStreamSynchronize(nullptr));
			unsigned long long cur = 0, overflow = 0;
			memcpy(&cur, h + o_cur, 8);
			memcpy(&overflow, h + o_err, 8);
			const size_t used = hb->seq_stride == 0 ? (cur < cap ? (size_t) cur : cap) : cap;
			memcpy(hb->results, h + o_res, n * sizeof(pamd_pair_result));
			memcpy(hb->seq_nt, h + o_nt, used);
			memcpy(hb->seq_p, h + o_p, used * sizeof(double));
			if (hb->seq_cursor)
				memcpy(hb->seq_cursor, &cur, 8);
			if (hb->counters) {
				const int64_t *c = reinterpret_cast<const int64_t *>(h + o_ctr);
				for (int i = 0; i < PAMD_NUM_COUNTERS; i++)
					hb->counters[i] += c[i];
			}
			if (overflow) {
				pamd_set_error("output arena too small: %llu assembled pairs did not fit in seq_capacity=%zu bases", overflow, cap);
				return PAMD_EINVAL;
			}
			return PAMD_OK;
		}
	}
	if ((rc = host_buf(a, 0, n * hb->fwd_stride * 2, &dfwd)) || (rc = host_buf(a, 1, n * hb->rev_stride * 2, &drev)) || (rc = host_buf(a, 2, n * 2, &dfl)) || (rc = host_buf(a, 3, n * 2, &drl)) || (rc = host_buf(a, 4, n * sizeof(pamd_pair_result), &dres)) || (rc = host_buf(a, 5, cap + 16, &dnt)) || (rc = host_buf(a, 6, (cap + 16) * sizeof(double), &dp)) || (rc = host_buf(a, 7, 8, &dcur)) || (rc = host_buf(a, 8, PAMD_NUM_COUNTERS * 8, &dctr)))
		return rc;
	HIP_TRY(hipMemcpy(dfwd, hb->fwd, n * hb->fwd_stride * 2, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(drev, hb->rev, n * hb->rev_stride * 2, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(dfl, hb->flen, n * 2, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(drl, hb->rlen, n * 2, hipMemcpyHostToDevice));
	HIP_TRY(hipMemset(dcur, 0, 8));
	HIP_TRY(hipMemset(dctr, 0, PAMD_NUM_COUNTERS * 8));
	HIP_TRY(hipMemset(a->d_error, 0, 8));
	d.fwd = static_cast<const pamd_qual *>(dfwd);
	d.rev = static_cast<const pamd_qual *>(drev);
	d.flen = static_cast<const uint16_t *>(dfl);
	d.rlen = static_cast<const uint16_t *>(drl);
	d.results = static_cast<pamd_pair_result *>(dres);
	d.seq_nt = static_cast<uint8_t *>(dnt);
	d.seq_p = static_cast<double *>(dp);
	d.seq_cursor = static_cast<uint64_t *>(dcur);
	d.counters = static_cast<int64_t *>(dctr);
	d.seq_capacity = cap;
	rc = pamd_assemble_device(a, &d, nullptr);
	if (rc != PAMD_OK)
		return rc;
	HIP_TRY(hipDeviceSynchronize());
	uint64_t overflow = 0;
	(void) pamd_assemble_status(a, &overflow);
	HIP_TRY(hipMemcpy(hb->results, dres, n * sizeof(pamd_pair_result), hipMemcpyDeviceToHost));
	// compact arena: only the part the launch used comes back
	size_t used = cap;
	if (hb->seq_stride == 0) {
		unsigned long long cur = 0;
		HIP_TRY(hipMemcpy(&cur, dcur, 8, hipMemcpyDeviceToHost));
		used = cur < cap ? (size_t) cur : cap;
	}
	HIP_TRY(hipMemcpy(hb->seq_nt, dnt, used, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(hb->seq_p, dp, used * sizeof(double), hipMemcpyDeviceToHost));
	if (hb->seq_cursor)
		HIP_TRY(hipMemcpy(hb->seq_cursor, dcur, 8, hipMemcpyDeviceToHost));
	if (hb->counters) {
		std::vector<int64_t> tmp(PAMD_NUM_COUNTERS);
		HIP_TRY(hipMemcpy(tmp.data(), dctr, PAMD_NUM_COUNTERS * 8, hipMemcpyDeviceToHost));
		for (int i = 0; i < PAMD_NUM_COUNTERS; i++)
			hb->counters[i] += tmp[i];
	}
	if (overflow) {
		pamd_set_error("output arena too small: %llu assembled pairs did not fit in seq_capacity=%zu bases", (unsigned long long) overflow, cap);
		return PAMD_EINVAL;
	}
	return PAMD_OK;
}

int pamd_synth_generate(int device, const pamd_synth_spec *spec, const uint32_t *thresholds_host, uint64_t first, size_t n, pamd_qual *fwd, pamd_qual *rev, size_t stride, uint16_t *flen, uint16_t *rlen, void *stream) {
	if (spec == nullptr || thresholds_host == nullptr || fwd == nullptr || rev == nullptr || flen == nullptr || rlen == nullptr) {
		pamd_set_error("pamd_synth_generate: null argument");
		return PAMD_EINVAL;
	}
	if (spec->read_len == 0 || spec->read_len > PAMD_MAX_LEN || stride < spec->read_len || spec->tmin < spec->read_len || spec->tmax < spec->tmin || spec->tmax > 1024 || spec->fprimer_len > 32 || spec->rprimer_len > 32) {
		pamd_set_error("pamd_synth_generate: spec out of range");
		return PAMD_EINVAL;
	}
	HIP_TRY(hipSetDevice(device));
	PamdSynthParams p;
	p.spec = *spec;
	memcpy(p.thr, thresholds_host, sizeof p.thr);
	p.first = first;
	p.n = n;
	p.fwd = reinterpret_cast<uint8_t *>(fwd);
	p.rev = reinterpret_cast<uint8_t *>(rev);
	p.flen = flen;
	p.rlen = rlen;
	p.stride = (uint32_t) stride;
	return pamd_launch_synth(p, stream);
}

}	// extern "C"
