// pamd_compat.cpp -- the reference's per-assembler API (include/pandaseq_amd_compat.h) on top of the
// batch ABI.  Host-side bookkeeping only: pulling pairs from the PandaNextSeq source, replaying
// per-pair callbacks and counters in source order (assembler.c:252-366), and the getters/setters of
// assembler_support.c.  All assembly arithmetic happens in the HIP kernel; nothing here scores,
// merges or sums anything.
#include <hip/hip_runtime_api.h>
#include <dirent.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <sched.h>
#include <atomic>
#include <cassert>
#include <cctype>
#include <emmintrin.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "pamd_internal.h"
#include "pamd_bz2.h"
#include "pamd_host_scorers.h"

static_assert(sizeof(panda_qual) == 2, "panda_qual ABI");
static_assert(sizeof(panda_result) == 16 && offsetof(panda_result, p) == 8, "panda_result ABI");
static_assert(sizeof(panda_seq_identifier) == 368, "panda_seq_identifier ABI");
static_assert(offsetof(panda_seq_identifier, tag) - offsetof(panda_seq_identifier, lane) == 16, "lane, tile, x, y are four adjacent ints");
static_assert(sizeof(panda_result_seq) == 480 && offsetof(panda_result_seq, sequence) == 384 && offsetof(panda_result_seq, overlap) == 464, "panda_result_seq ABI");
static_assert(sizeof(panda_qual) == sizeof(pamd_qual), "panda_qual == pamd_qual");

namespace {
// Which GPU a new assembler lives on: the calling thread's own choice (pamd_compat_set_thread_device), else the next
// entry of PAMD_DEVICES ("0,1,2,3": an embedder that builds one assembler per thread, as pool.c:140-160 does, then
// spreads them over the GPUs without a code change), else the process default (pamd_compat_set_device).
std::atomic<int> g_device{0};
thread_local int t_device = -1;
std::atomic<unsigned> g_device_rr{0};
std::atomic<size_t> g_batch_pairs{65536};
// Summation mode of the assemblers this layer creates (pamd_set_exact_sums, include/pandaseq_amd.h).  Default: any order with
// exactness guards -- every decision, merged base and per-base log-p bit-identical to the reference, `quality` and
// `estimated_overlap_probability` within 1e-9 (the tolerance this path is specified to is 1e-6, and panda_diff, diff.c:26-99,
// compares bases and log-p, never those two scalars).  PAMD_COMPAT_SUMS=exact in the environment, or
// pamd_compat_set_exact_sums(1), selects sums in the reference's own order (every output bit bit-identical, ~20 % slower).
// Where panda_assembler_next() builds a record's panda_result[] (16 bytes per base): 0 = by helper threads, batch by batch,
// ahead of the caller (streamed to memory: a consumer that only looks at a record's scalars never touches the rows);
// 1 = on the caller's thread when the record is handed over, into one buffer that stays in its cache (a consumer that reads
// every base reads it from there instead of from memory other threads wrote).  PAMD_COMPAT_ROWS=lazy selects 1.
std::atomic<int> g_streaming{0};	// assemblers that have batches in flight (panda_assembler_next was called on them)

// CPUs this process may actually use: the affinity mask, capped by the cgroup's quota (cpu.max: "<quota> <period>" or "max")
unsigned effective_cpus() {
	static const unsigned n = [] {
		unsigned cpus = std::thread::hardware_concurrency();
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
			cpus = (unsigned) CPU_COUNT(&set);
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			char quota[32] = "";
			double period = 0;
			if (fscanf(f, "%31s %lf", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
				const double q = atof(quota) / period;
				if (q >= 1 && q < cpus)
					cpus = (unsigned) (q + 0.5);
			}
			fclose(f);
		}
		return cpus < 1 ? 1u : cpus;
	}();
	return n;
}

std::atomic<int> g_lazy_rows{-1};
int compat_lazy_rows() {
	int v = g_lazy_rows.load();
	if (v < 0) {
		const char *e = getenv("PAMD_COMPAT_ROWS");
		v = (e != nullptr && strcmp(e, "lazy") == 0) ? 1 : 0;
		g_lazy_rows.store(v);
	}
	return v;
}
std::atomic<int> g_per_pair_device{-1};
int compat_per_pair_device() {
	int v = g_per_pair_device.load();
	if (v < 0) {
		const char *e = getenv("PAMD_COMPAT_PER_PAIR");
		v = (e != nullptr && strcmp(e, "device") == 0) ? 1 : 0;
		g_per_pair_device.store(v);
	}
	return v;
}
std::atomic<int> g_exact_sums{-1};
std::atomic<bool> g_fast_sums_asked{false};	// any-order sums asked for in so many words (push_config: a host `check` then does not force the reference's order)
std::atomic<bool> g_forced_said{false};
int compat_exact_sums() {
	int v = g_exact_sums.load();
	if (v < 0) {
		const char *e = getenv("PAMD_COMPAT_SUMS");
		v = (e != nullptr && strcmp(e, "exact") == 0) ? 1 : 0;
		if (e != nullptr && strcmp(e, "fast") == 0)
			g_fast_sums_asked.store(true);
		g_exact_sums.store(v);
	}
	return v;
}
std::mutex g_ref_lock;	// one lock for all refcounts, like module.c:38-41

int pick_device() {
	if (t_device >= 0)
		return t_device;
	if (const char *list = getenv("PAMD_DEVICES")) {
		std::vector<int> ids;
		for (const char *c = list; *c;) {
			char *end = nullptr;
			const long v = strtol(c, &end, 10);
			if (end == c)
				break;
			ids.push_back((int) v);
			c = *end == ',' ? end + 1 : end;
		}
		if (!ids.empty())
			return ids[g_device_rr.fetch_add(1) % ids.size()];
	}
	return g_device.load();
}
constexpr size_t ROW = 456;	// staging row stride in bases (>= PANDA_MAX_LEN, multiple of 8)
constexpr size_t OUT_ROW = 912;	// strided output row (>= 2 * PANDA_MAX_LEN, multiple of 16)
}

// ---------------------------------------------------------------- algorithms (algo.c, algo_*.c)
#ifndef PAMD_WITH_REFERENCE_OBJECTS
struct panda_algorithm {	// algo.h:28-35
	const struct panda_algorithm_class *clazz;
	size_t refcnt;
	double data[3];	// simple_bayes: q, pmatch, pmismatch (algo_simple_bayes.c:27-31); pear: random_base
};

static PandaAlgorithm sb_create(const char *arg) {	// algo_simple_bayes.c:77-98
	PandaAlgorithm a = panda_algorithm_simple_bayes_new();
	if (arg != nullptr) {
		char *end;
		double q = strtod(arg, &end);
		if (*end != '\0' || q < 0 || q > 1) {
			panda_algorithm_unref(a);
			return nullptr;
		}
		panda_algorithm_simple_bayes_set_error_estimation(a, q);
	}
	return a;
}

static PandaAlgorithm pear_create(const char *arg) {	// algo_pear.c:70-91
	PandaAlgorithm a = panda_algorithm_pear_new();
	if (arg != nullptr) {
		char *end;
		double p = strtod(arg, &end);
		if (*end != '\0' || p < 0 || p > 1) {
			panda_algorithm_unref(a);
			return nullptr;
		}
		panda_algorithm_pear_set_random_base_log_p(a, std::log(p));
	}
	return a;
}

static PandaAlgorithm rdp_create(const char *arg) {	// algo_rdp_mle.c:76-82
	if (arg == nullptr || strlen(arg) == 0)
		return panda_algorithm_rdp_mle_new();
	return nullptr;
}

static PandaAlgorithm uparse_create(const char *arg) {	// algo_uparse.c:77-98
	PandaAlgorithm a = panda_algorithm_uparse_new();
	if (arg != nullptr) {
		char *end;
		double q = strtod(arg, &end);
		if (*end != '\0' || q < 0 || q > 1) {
			panda_algorithm_unref(a);
			return nullptr;
		}
		panda_algorithm_uparse_set_error_estimation(a, q);
	}
	return a;
}

// algo_flash.c:82-89, algo_ea_util.c:69-77, algo_stitch.c:69-77: "No arguments allowed"
static PandaAlgorithm flash_create(const char *arg) {
	return arg == nullptr ? panda_algorithm_flash_new() : nullptr;
}

static PandaAlgorithm ea_util_create(const char *arg) {
	return arg == nullptr ? panda_algorithm_ea_util_new() : nullptr;
}

static PandaAlgorithm stitch_create(const char *arg) {
	return arg == nullptr ? panda_algorithm_stitch_new() : nullptr;
}

extern "C" {
// Every class object is complete (pandaseq-common.h:594-605): `overlap_probability` / `match_probability` are the host
// functions of pamd_host_scorers.cpp -- what a caller or a plugin gets when it scores through a class object, and what the
// per-pair entry point runs on; batches and streams of pairs are scored by the kernels (device_algo_of() below picks them by
// class, never by these pointers).
#define PAMD_CLASS_FNS(ALGO)                                                                                                                                        \
	static double ALGO##_overlap_fn(void *d, const panda_qual *f, size_t fl, const panda_qual *r, size_t rl, size_t ov) { return pamd_host_scorers[ALGO].overlap_probability(d, f, fl, r, rl, ov); } \
	static double ALGO##_match_fn(void *d, bool m, char a, char b) { return pamd_host_scorers[ALGO].match_probability(d, m, a, b); }
PAMD_CLASS_FNS(PAMD_ALGO_SIMPLE_BAYES)
PAMD_CLASS_FNS(PAMD_ALGO_PEAR)
PAMD_CLASS_FNS(PAMD_ALGO_RDP_MLE)
PAMD_CLASS_FNS(PAMD_ALGO_UPARSE)
PAMD_CLASS_FNS(PAMD_ALGO_FLASH)
PAMD_CLASS_FNS(PAMD_ALGO_EA_UTIL)
PAMD_CLASS_FNS(PAMD_ALGO_STITCH)
#undef PAMD_CLASS_FNS
const struct panda_algorithm_class panda_algorithm_uparse_class = { 3 * sizeof(double), "uparse", uparse_create, nullptr, PAMD_ALGO_UPARSE_overlap_fn, PAMD_ALGO_UPARSE_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_flash_class = { 0, "flash", flash_create, nullptr, PAMD_ALGO_FLASH_overlap_fn, PAMD_ALGO_FLASH_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_ea_util_class = { 0, "ea_util", ea_util_create, nullptr, PAMD_ALGO_EA_UTIL_overlap_fn, PAMD_ALGO_EA_UTIL_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_stitch_class = { 0, "stitch", stitch_create, nullptr, PAMD_ALGO_STITCH_overlap_fn, PAMD_ALGO_STITCH_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_simple_bayes_class = { 3 * sizeof(double), "simple_bayesian", sb_create, nullptr, PAMD_ALGO_SIMPLE_BAYES_overlap_fn, PAMD_ALGO_SIMPLE_BAYES_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_pear_class = { sizeof(double), "pear", pear_create, nullptr, PAMD_ALGO_PEAR_overlap_fn, PAMD_ALGO_PEAR_match_fn, -1.38629 };
const struct panda_algorithm_class panda_algorithm_rdp_mle_class = { 0, "rdp_mle", rdp_create, nullptr, PAMD_ALGO_RDP_MLE_overlap_fn, PAMD_ALGO_RDP_MLE_match_fn, -1.38629 };

PandaAlgorithm panda_algorithm_new(PandaAlgorithmClass clazz) {	// algo.c:85-94
	PandaAlgorithm a = static_cast<PandaAlgorithm>(calloc(1, sizeof(struct panda_algorithm)));
	a->clazz = clazz;
	a->refcnt = 1;
	return a;
}

PandaAlgorithm panda_algorithm_ref(PandaAlgorithm a) {
	std::lock_guard<std::mutex> g(g_ref_lock);
	a->refcnt++;
	return a;
}

void panda_algorithm_unref(PandaAlgorithm a) {
	if (a == nullptr)
		return;
	size_t c;
	{
		std::lock_guard<std::mutex> g(g_ref_lock);
		c = --a->refcnt;
	}
	if (c == 0)
		free(a);
}

bool panda_algorithm_is_a(PandaAlgorithm a, PandaAlgorithmClass clazz) {	// algo.c:44-48
	return a != nullptr && a->clazz == clazz;
}

PandaAlgorithmClass panda_algorithm_class(PandaAlgorithm a) {
	return a->clazz;
}

void *panda_algorithm_data(PandaAlgorithm a) {
	return a->data;
}

PandaAlgorithm panda_algorithm_simple_bayes_new(void) {	// algo_simple_bayes.c:110-115
	PandaAlgorithm a = panda_algorithm_new(&panda_algorithm_simple_bayes_class);
	panda_algorithm_simple_bayes_set_error_estimation(a, 0.36);
	return a;
}

double panda_algorithm_simple_bayes_get_error_estimation(PandaAlgorithm a) {
	return panda_algorithm_is_a(a, &panda_algorithm_simple_bayes_class) ? a->data[0] : -1;
}

void panda_algorithm_simple_bayes_set_error_estimation(PandaAlgorithm a, double q) {	// algo_simple_bayes.c:126-135
	if (q > 0 && q < 1 && panda_algorithm_is_a(a, &panda_algorithm_simple_bayes_class)) {
		a->data[0] = q;
		a->data[1] = std::log(0.25 * (1 - 2 * q + q * q));
		a->data[2] = std::log((3 * q - 2 * q * q) / 18.0);
	}
}

PandaAlgorithm panda_algorithm_pear_new(void) {	// algo_pear.c:103-108
	PandaAlgorithm a = panda_algorithm_new(&panda_algorithm_pear_class);
	a->data[0] = std::log(0.25);
	return a;
}

double panda_algorithm_pear_get_random_base_log_p(PandaAlgorithm a) {
	return panda_algorithm_is_a(a, &panda_algorithm_pear_class) ? a->data[0] : 1;
}

void panda_algorithm_pear_set_random_base_log_p(PandaAlgorithm a, double log_p) {
	if (panda_algorithm_is_a(a, &panda_algorithm_pear_class))
		a->data[0] = log_p;
}

PandaAlgorithm panda_algorithm_rdp_mle_new(void) {
	return panda_algorithm_new(&panda_algorithm_rdp_mle_class);
}

PandaAlgorithm panda_algorithm_uparse_new(void) {	// algo_uparse.c:104-109
	PandaAlgorithm a = panda_algorithm_new(&panda_algorithm_uparse_class);
	panda_algorithm_uparse_set_error_estimation(a, 0.36);
	return a;
}

double panda_algorithm_uparse_get_error_estimation(PandaAlgorithm a) {	// algo_uparse.c:111-118
	return panda_algorithm_is_a(a, &panda_algorithm_uparse_class) ? a->data[0] : -1;
}

void panda_algorithm_uparse_set_error_estimation(PandaAlgorithm a, double q) {	// algo_uparse.c:120-135
	if (q > 0 && q < 1 && panda_algorithm_is_a(a, &panda_algorithm_uparse_class)) {
		a->data[0] = q;
		a->data[1] = std::log(1 - q * q * (1 - 2 * q + 4 * q * q / 3));
		a->data[2] = std::log(1 - 4 * q / 3 / (2 * q - 4 * q * q / 3));
	}
}

PandaAlgorithm panda_algorithm_flash_new(void) {	// algo_flash.c:101-104
	return panda_algorithm_new(&panda_algorithm_flash_class);
}

PandaAlgorithm panda_algorithm_ea_util_new(void) {	// algo_ea_util.c:89-92
	return panda_algorithm_new(&panda_algorithm_ea_util_class);
}

PandaAlgorithm panda_algorithm_stitch_new(void) {	// algo_stitch.c:89-92
	return panda_algorithm_new(&panda_algorithm_stitch_class);
}

// the registry (algo.c:96-133): a name-sorted array the front end's -A option searches
PandaAlgorithmClass *panda_algorithms = nullptr;
size_t panda_algorithms_length = 0;
static size_t g_algorithms_size = 0;

void panda_algorithm_register(PandaAlgorithmClass clazz) {
	for (size_t it = 0; it < panda_algorithms_length; it++)
		if (panda_algorithms[it] == clazz)
			return;
	if (panda_algorithms_length == g_algorithms_size) {
		g_algorithms_size = g_algorithms_size ? 2 * g_algorithms_size : 16;
		panda_algorithms = static_cast<PandaAlgorithmClass *>(realloc(panda_algorithms, g_algorithms_size * sizeof(PandaAlgorithmClass)));
	}
	// insertion keeps the array sorted by name
	size_t at = panda_algorithms_length++;
	while (at > 0 && strcmp(panda_algorithms[at - 1]->name, clazz->name) > 0) {
		panda_algorithms[at] = panda_algorithms[at - 1];
		at--;
	}
	panda_algorithms[at] = clazz;
}

__attribute__((constructor)) static void register_builtin_algorithms(void) {	// algo.c:123-133
	panda_algorithm_register(&panda_algorithm_ea_util_class);
	panda_algorithm_register(&panda_algorithm_flash_class);
	panda_algorithm_register(&panda_algorithm_pear_class);
	panda_algorithm_register(&panda_algorithm_rdp_mle_class);
	panda_algorithm_register(&panda_algorithm_simple_bayes_class);
	panda_algorithm_register(&panda_algorithm_stitch_class);
	panda_algorithm_register(&panda_algorithm_uparse_class);
}
}	// extern "C"
#endif	// PAMD_WITH_REFERENCE_OBJECTS (the algorithm objects)

// ---------------------------------------------------------------- modules (module.c)
// Ours in every build: the reference's module.c includes the private assembler.h and works on `struct panda_assembler`'s
// fields (module.c:99-220: `assembler->modules`, `->rejected`, `->logger`, `->mutex`), so it leaves the link together with
// assembler.c and assembler_support.c, and there is no public accessor for a module's callbacks.
struct panda_module {
	size_t refcnt;
	PandaCheck check;
	PandaPreCheck precheck;
	PandaDestroy destroy;
	void *user_data;
	char *name;
	char *args;	// the text after ':' in panda_module_load's path (module.c:284), NULL when there was none
	void *handle;	// dlopen handle of a plugin file; NULL for panda_module_new and the two fused modules
	int api;
	const char **version;
	int fused_kind;	// 0 = host callbacks; 1 = min_phred, 2 = min_overlapbits (run inside the kernel)
	int fused_phred;
	double fused_bits;
};

// A plugin is a shared object exporting <module>_LTX_api, <module>_LTX_opener and optionally _LTX_desc / _LTX_usage /
// _LTX_version (pandaseq-plugin.h:33-35).  module.c:227-298 finds it with libltdl's lt_dlopenext(); here it is
// dlopen() on the name as given, then with ".so" appended, then under $PANDASEQ_PLUGIN_DIR.  libltdl looks a symbol up
// as "<module>_LTX_<sym>" before "<sym>", with <module> the file's base name less "lib" and the extension.
static void *plugin_symbol(void *handle, const std::string &module, const char *sym) {
	void *p = dlsym(handle, (module + "_LTX_" + sym).c_str());
	if (p == nullptr && module.compare(0, 3, "lib") == 0)
		p = dlsym(handle, (module.substr(3) + "_LTX_" + sym).c_str());
	return p != nullptr ? p : dlsym(handle, sym);
}

static std::string plugin_module_name(const std::string &path) {
	size_t slash = path.rfind('/');
	std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
	size_t dot = base.find('.');
	if (dot != std::string::npos)
		base.resize(dot);
	for (char &c : base)
		if (!isalnum((unsigned char) c))
			c = '_';
	return base;
}

typedef bool (*PluginOpen) (PandaLogProxy logger, const char *args, PandaPreCheck *precheck, PandaCheck *check, void **user_data, PandaDestroy *destroy);

static bool load_plugin_file(PandaModule m, PandaLogProxy logger, const std::string &name, const char *args) {
	std::vector<std::string> tries = { name, name + ".so" };
	const char *dir = getenv("PANDASEQ_PLUGIN_DIR");
	if (dir != nullptr && name.find('/') == std::string::npos) {
		tries.push_back(std::string(dir) + "/" + name);
		tries.push_back(std::string(dir) + "/" + name + ".so");
	}
	void *handle = nullptr;
	std::string why;
	for (const std::string &t : tries) {
		// a bare name must not be searched for along the loader's default path: only files
		if (t.find('/') == std::string::npos)
			handle = dlopen(("./" + t).c_str(), RTLD_NOW | RTLD_LOCAL);
		else
			handle = dlopen(t.c_str(), RTLD_NOW | RTLD_LOCAL);
		if (handle != nullptr)
			break;
		const char *e = dlerror();
		if (why.empty() && e != nullptr)
			why = e;
	}
	if (handle == nullptr) {
		fprintf(stderr, "Could not open module %s: %s\n", name.c_str(), why.c_str());
		pamd_set_error("could not open module %s: %s", name.c_str(), why.c_str());
		return false;
	}
	const std::string module = plugin_module_name(name);
	int *api = static_cast<int *>(plugin_symbol(handle, module, "api"));
	if (api == nullptr || *api != PANDA_API) {	// module.c:263-270
		fprintf(stderr, "Invalid API in %s (%d != %d). Are you sure this module was compiled for this version of PANDAseq?\n", name.c_str(), api == nullptr ? -1 : *api, PANDA_API);
		pamd_set_error("invalid API in module %s", name.c_str());
		dlclose(handle);
		return false;
	}
	PluginOpen opener = reinterpret_cast<PluginOpen>(plugin_symbol(handle, module, "opener"));
	if (opener == nullptr) {
		fprintf(stderr, "Could not find opener function in %s\n", name.c_str());
		pamd_set_error("could not find opener function in %s", name.c_str());
		dlclose(handle);
		return false;
	}
	PandaCheck check = nullptr;
	PandaPreCheck precheck = nullptr;
	void *user_data = nullptr;
	PandaDestroy destroy = nullptr;
	if (!opener(logger, args, &precheck, &check, &user_data, &destroy) || (precheck == nullptr && check == nullptr)) {	// module.c:284
		pamd_set_error("module %s refused its arguments", name.c_str());
		dlclose(handle);
		return false;
	}
	m->handle = handle;
	m->api = *api;
	m->check = check;
	m->precheck = precheck;
	m->user_data = user_data;
	m->destroy = destroy;
	m->version = static_cast<const char **>(plugin_symbol(handle, module, "version"));
	return true;
}

extern "C" {
PandaModule panda_module_new(const char *name, PandaCheck check, PandaPreCheck precheck, void *user_data, PandaDestroy cleanup) {	// module.c:300-321
	if (check == nullptr && precheck == nullptr)
		return nullptr;
	PandaModule m = static_cast<PandaModule>(calloc(1, sizeof(struct panda_module)));
	m->refcnt = 1;
	m->check = check;
	m->precheck = precheck;
	m->destroy = cleanup;
	m->user_data = user_data;
	m->name = strdup(name);
	m->api = PANDA_API;
	return m;
}

PandaModule panda_module_load(PandaLogProxy logger, const char *path) {
	(void) logger;
	if (path == nullptr)
		return nullptr;
	// "name:args" (module.c:244-247); only the two modules the kernel implements are known here
	const char *colon = strchr(path, ':');
	std::string name = colon ? std::string(path, colon - path) : std::string(path);
	const char *args = colon ? colon + 1 : nullptr;
	PandaModule m = static_cast<PandaModule>(calloc(1, sizeof(struct panda_module)));
	m->refcnt = 1;
	m->name = strdup(name.c_str());
	m->args = args ? strdup(args) : nullptr;
	m->api = PANDA_API;
	if (name == "min_phred") {	// plugin_min_phred.c:24-44
		char *end = nullptr;
		long v = (args && *args) ? strtol(args, &end, 10) : -1;
		if (args == nullptr || *args == '\0' || (end && *end != '\0') || v < 0 || v > 127) {
			panda_module_unref(m);
			return nullptr;
		}
		m->fused_kind = 1;
		m->fused_phred = (int) v;
	} else if (name == "min_overlapbits") {	// plugin_min_overlapbits.c:25-53
		double bits = 15;
		if (args != nullptr) {
			char *end = nullptr;
			bits = strtod(args, &end);
			if (*end != '\0' || bits < 0) {
				panda_module_unref(m);
				return nullptr;
			}
		}
		m->fused_kind = 2;
		m->fused_bits = bits;
	} else if (!load_plugin_file(m, logger, name, args)) {
		panda_module_unref(m);
		return nullptr;
	}
	return m;
}

PandaModule panda_module_ref(PandaModule m) {
	std::lock_guard<std::mutex> g(g_ref_lock);
	m->refcnt++;
	return m;
}

void panda_module_unref(PandaModule m) {
	if (m == nullptr)
		return;
	size_t c;
	{
		std::lock_guard<std::mutex> g(g_ref_lock);
		c = --m->refcnt;
	}
	if (c == 0) {
		if (m->destroy != nullptr && m->user_data != nullptr)
			m->destroy(m->user_data);
		free(m->name);
		free(m->args);
		if (m->handle != nullptr)
			dlclose(m->handle);
		free(m);
	}
}

const char *panda_module_get_name(PandaModule m) {
	return m->name;
}

int panda_module_get_api(PandaModule m) {	// module.c:360-363
	return m->api;
}

const char *panda_module_get_args(PandaModule m) {	// pandaseq-module.h: the arguments the module was loaded with
	return m->args;
}

static const char *module_string(PandaModule m, const char *sym) {
	if (m->handle == nullptr)
		return nullptr;
	const char **v = static_cast<const char **>(plugin_symbol(m->handle, plugin_module_name(m->name), sym));
	return v == nullptr ? nullptr : *v;
}

const char *panda_module_get_description(PandaModule m) {	// module.c:365-372
	if (m->fused_kind == 1)
		return "Require every base of the assembled sequence to reach a minimum PHRED score";
	if (m->fused_kind == 2)
		return "Require a minimum amount of information in the overlap";
	return module_string(m, "desc");
}

const char *panda_module_get_usage(PandaModule m) {	// module.c:384-391
	if (m->fused_kind == 1)
		return "min_phred:value";
	if (m->fused_kind == 2)
		return "min_overlapbits:15";
	return module_string(m, "usage");
}

const char *panda_module_get_version(PandaModule m) {	// module.c:379-382
	if (m->fused_kind != 0)
		return "1.0";	// VER_INFO of plugin_min_phred.c:6 / plugin_min_overlapbits.c:15
	return m->version == nullptr ? nullptr : *m->version;
}

#ifdef PAMD_WITH_REFERENCE_OBJECTS
// module.c:393-451, the one internal name of module.c a file that STAYS in the link calls (args.c:273, the front end's help):
// the modules this library knows -- the two that run inside the kernel, then the plugin files of $PANDASEQ_PLUGIN_DIR.
void module_show_all(void) {
	fprintf(stderr, "\nKnown modules:\n");
	fprintf(stderr, "min_phred(1.0): %s\n\tUsage: %s\n", "Require every base of the assembled sequence to reach a minimum PHRED score", "min_phred:value");
	fprintf(stderr, "min_overlapbits(1.0): %s\n\tUsage: %s\n", "Require a minimum amount of information in the overlap", "min_overlapbits:15");
	const char *dir = getenv("PANDASEQ_PLUGIN_DIR");
	DIR *d = dir != nullptr ? opendir(dir) : nullptr;
	if (d == nullptr)
		return;
	while (struct dirent *e = readdir(d)) {
		const std::string file = e->d_name;
		if (file.size() < 4 || file.compare(file.size() - 3, 3, ".so") != 0)
			continue;
		void *handle = dlopen((std::string(dir) + "/" + file).c_str(), RTLD_LAZY | RTLD_LOCAL);
		if (handle == nullptr)
			continue;
		const std::string module = plugin_module_name(file);
		const int *api = static_cast<const int *>(plugin_symbol(handle, module, "api"));
		if (api == nullptr || *api != PANDA_API) {
			fprintf(stderr, "Invalid API in %s (%d != %d).\n", file.c_str(), api == nullptr ? -1 : *api, PANDA_API);
		} else {
			const char **version = static_cast<const char **>(plugin_symbol(handle, module, "version"));
			const char **desc = static_cast<const char **>(plugin_symbol(handle, module, "desc"));
			const char **usage = static_cast<const char **>(plugin_symbol(handle, module, "usage"));
			fprintf(stderr, "%s", module.c_str());
			if (version != nullptr && *version != nullptr)
				fprintf(stderr, "(%s)", *version);
			fprintf(stderr, desc != nullptr && *desc != nullptr ? ": %s\n" : "%s\n", desc != nullptr && *desc != nullptr ? *desc : "");
			if (usage != nullptr && *usage != nullptr)
				fprintf(stderr, "\tUsage: %s\n", *usage);
		}
		dlclose(handle);
	}
	closedir(d);
}
#endif
}	// extern "C"

// ---------------------------------------------------------------- the assembler
struct panda_assembler {
	size_t refcnt = 1;
	PandaNextSeq next = nullptr;
	void *next_data = nullptr;
	PandaDestroy next_destroy = nullptr;
	PandaFailAlign noalgn = nullptr;
	void *noalgn_data = nullptr;
	PandaDestroy noalgn_destroy = nullptr;
	PandaLogProxy logger = nullptr;
	pamd_assembler *dev = nullptr;
	bool dev_dirty = true;	// configuration changed since it was last pushed to `dev`
	// configuration (assembler.h:44-57,72-78)
	double threshold = std::log(0.6);
	int minoverlap = 2, maxoverlap = 0;
	size_t num_kmers = 2;
	PandaAlgorithm algo = nullptr;
	panda_nt forward_primer[PAMD_MAX_LEN], reverse_primer[PAMD_MAX_LEN];
	size_t forward_primer_length = 0, reverse_primer_length = 0, forward_trim = 0, reverse_trim = 0;
	bool post_primers = false;
	double primer_penalty = 0;
	char name[PAMD_MAX_LEN] = "";
	std::vector<PandaModule> modules;
	std::vector<size_t> rejected;
	size_t fused_prefix = 0;	// modules[0..fused_prefix) run inside the kernel
	// counters (assembler.h:59-67,75-76)
	long nofpcount = 0, norpcount = 0, okcount = 0, lowqcount = 0, noalgncount = 0, badreadcount = 0, slowcount = 0, count = 0;
	long overlapcount[2 * PAMD_MAX_LEN] = { 0 };
	size_t longest_overlap = 0;
	// the result handed back (valid until the next call)
	panda_result_seq result;
	panda_result result_seq[2 * PAMD_MAX_LEN];
	// batch staging (host)
	std::vector<pamd_qual> h_fwd, h_rev;
	std::vector<uint16_t> h_flen, h_rlen;
	std::vector<panda_seq_identifier> h_id;
	std::vector<int> h_pre;	// -2 = sent to the device, -1 = short read (BADR), >= 0 = rejected by precheck module k
	std::vector<pamd_pair_result> h_res;
	std::vector<uint8_t> h_nt;
	std::vector<double> h_p;
	size_t batch_n = 0, batch_pos = 0;
	bool eof = false;
	// panda_assembler_next(): two batches in flight (see "the pipelined source path" below)
	static constexpr int NSLOT = 4;	// batches in flight: upload, kernels, download and expansion of different batches overlap
	struct pamd_slot *slots[NSLOT] = { nullptr, nullptr, nullptr, nullptr };
	pamd_assembler *devs[NSLOT] = { nullptr, nullptr, nullptr, nullptr };	// one device assembler per slot (a launch owns its assembler's deferred list); devs[0] == dev
	int cur = 0;
	bool primed = false;
	std::vector<double> image;	// host copy of the device's table image: log-p comes back as indices into it
	double t_fill = 0, t_wait = 0, t_bg = 0;	// PAMD_COMPAT_STATS=1: seconds pulling + staging, blocked on a batch, and of background work
	long n_batches = 0;
	int device = 0;
};

namespace {

// which kernel scorer an algorithm object is (-1: none: a class registered by the caller), and its parameter
int device_algo_of(PandaAlgorithm algorithm, double *param_out = nullptr) {
	int algo = -1;
	double param = NAN;
	if (algorithm == nullptr) {
		algo = -1;
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_simple_bayes_class)) {
		algo = PAMD_ALGO_SIMPLE_BAYES;
		param = panda_algorithm_simple_bayes_get_error_estimation(algorithm);
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_pear_class)) {
		algo = PAMD_ALGO_PEAR;
		param = panda_algorithm_pear_get_random_base_log_p(algorithm);
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_rdp_mle_class)) {
		algo = PAMD_ALGO_RDP_MLE;
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_uparse_class)) {
		algo = PAMD_ALGO_UPARSE;
		param = panda_algorithm_uparse_get_error_estimation(algorithm);
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_flash_class)) {
		algo = PAMD_ALGO_FLASH;
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_ea_util_class)) {
		algo = PAMD_ALGO_EA_UTIL;
	} else if (panda_algorithm_is_a(algorithm, &panda_algorithm_stitch_class)) {
		algo = PAMD_ALGO_STITCH;
	}
	if (param_out != nullptr)
		*param_out = param;
	return algo;
}

// pairs of this assembler are assembled one by one on the caller's thread (host_assemble): an algorithm class of the caller's, or
// a k-mer table of other than two positions per k-mer (the kernels' seeding is the default's, num_kmers == 2)
bool host_routed(PandaAssembler a) {
	return device_algo_of(a->algo) < 0 || a->num_kmers != 2;
}

int push_config(PandaAssembler a) {
	if (a->dev == nullptr) {	// (pamd_compat_host_only_assembler_new: a test hook; nothing but the per-pair host path works on it)
		pamd_set_error("this assembler has no device: only panda_assembler_assemble() runs on it");
		return PAMD_ENODEVICE;
	}
	if (!a->dev_dirty)
		return PAMD_OK;
	double param = NAN;
	const int algo = device_algo_of(a->algo, &param);
	if (algo < 0) {	// (panda_assembler_set_algorithm admits such a class only with its own functions, and then the host path runs)
		pamd_set_error("algorithm %s has no device implementation", a->algo ? panda_algorithm_class(a->algo)->name : "(null)");
		return PAMD_EUNSUPPORTED;
	}
	// only a leading run of kernel-implemented modules can be fused without changing which module
	// gets the blame for a rejection (module.c:128-134 stops at the first failure)
	a->fused_prefix = 0;
	for (PandaModule m : a->modules) {
		if (m->fused_kind == 0 || a->fused_prefix >= PAMD_MAX_MODULES)
			break;
		a->fused_prefix++;
	}
	// (a kernel-implemented module that FOLLOWS a host-callback module runs on the host too, from the same definitions:
	// host_module_check() below)
	// A `check` that runs on the host reads `quality` and `estimated_overlap_probability` and may compare them with a threshold
	// of its own -- min_overlapbits behind a host module does, a caller's callback may --: the kernel's guard (a fused verdict
	// within 1e-6 of its threshold is re-made from sums in the reference's order) does not reach it, so such an assembler sums
	// in the reference's order throughout.
	// Two cases.  min_phred / min_overlapbits behind a host module: the verdict is THIS library's, made on the host from the
	// record's sums -- reference order always.  A caller's own `check`: reference order unless the caller has asked for any-order
	// sums in so many words (pamd_compat_set_exact_sums(0) called, or PAMD_COMPAT_SUMS=fast): a callback that never looks at
	// `quality` or the score pays a fifth of the device's rate for nothing otherwise.  Said once on stderr when it applies.
	bool fused_behind_host = false, host_check = false;
	for (size_t m = a->fused_prefix; m < a->modules.size(); m++) {
		fused_behind_host |= a->modules[m]->fused_kind != 0;
		host_check |= a->modules[m]->check != nullptr;
	}
	const bool forced = !compat_exact_sums() && (fused_behind_host || (host_check && !g_fast_sums_asked.load()));
	if (forced && !g_forced_said.exchange(true))
		fprintf(stderr, "pandaseq_amd: a check module runs on the host behind the device: sums in the reference's order for this assembler (%s)\n",
			fused_behind_host ? "min_phred / min_overlapbits follows a host module" : "pamd_compat_set_exact_sums(0) or PAMD_COMPAT_SUMS=fast keeps any-order sums");
	const int exact = (compat_exact_sums() || forced) ? 1 : 0;
	for (pamd_assembler *d : a->devs) {
		if (d == nullptr)
			continue;
		pamd_set_exact_sums(d, exact);
		pamd_set_algorithm(d, algo, param);
		pamd_set_primers_after(d, a->post_primers ? 1 : 0);
		pamd_set_log_threshold(d, a->threshold);	// the stored logarithm itself (assembler_support.c:395): no exp / log round trip
		pamd_set_minimum_overlap(d, a->minoverlap);
		pamd_set_maximum_overlap(d, a->maxoverlap);
		pamd_set_primer_penalty(d, a->primer_penalty);
		if (a->forward_primer_length > 0)
			pamd_set_forward_primer(d, reinterpret_cast<const uint8_t *>(a->forward_primer), a->forward_primer_length);
		else
			pamd_set_forward_trim(d, a->forward_trim);
		if (a->reverse_primer_length > 0)
			pamd_set_reverse_primer(d, reinterpret_cast<const uint8_t *>(a->reverse_primer), a->reverse_primer_length);
		else
			pamd_set_reverse_trim(d, a->reverse_trim);
		pamd_clear_modules(d);
		for (size_t m = 0; m < a->fused_prefix; m++) {
			if (a->modules[m]->fused_kind == 1)
				pamd_add_module_min_phred(d, a->modules[m]->fused_phred);
			else
				pamd_add_module_min_overlapbits(d, a->modules[m]->fused_bits);
		}
	}	// every slot's device assembler
	{
		a->image.assign(8192, 0.0);
		size_t used = 0;
		const int rc = pamd_table_image(a->dev, a->image.data(), a->image.size(), &used);
		if (rc != PAMD_OK)
			return rc;
		a->image.resize(used);
	}
	a->dev_dirty = false;
	return PAMD_OK;
}

// Run the staged pairs [0, batch_n) through the device.
int run_batch(PandaAssembler a) {
	int rc = push_config(a);
	if (rc != PAMD_OK)
		return rc;
	std::vector<size_t> sent;
	for (size_t i = 0; i < a->batch_n; i++)
		if (a->h_pre[i] == -2)
			sent.push_back(i);
	const size_t n = sent.size();
	if (n == 0)
		return PAMD_OK;
	// compact the pairs that go to the device to the front of the staging rows (order kept)
	for (size_t k = 0; k < n; k++) {
		const size_t i = sent[k];
		if (i != k) {
			memcpy(&a->h_fwd[k * ROW], &a->h_fwd[i * ROW], ROW * sizeof(pamd_qual));
			memcpy(&a->h_rev[k * ROW], &a->h_rev[i * ROW], ROW * sizeof(pamd_qual));
			a->h_flen[k] = a->h_flen[i];
			a->h_rlen[k] = a->h_rlen[i];
		}
	}
	a->h_res.resize(n);
	a->h_nt.resize(n * OUT_ROW);
	a->h_p.resize(n * OUT_ROW);
	pamd_batch b;
	memset(&b, 0, sizeof b);
	b.n_pairs = n;
	b.fwd = a->h_fwd.data();
	b.rev = a->h_rev.data();
	b.fwd_stride = b.rev_stride = ROW;
	b.flen = a->h_flen.data();
	b.rlen = a->h_rlen.data();
	b.results = a->h_res.data();
	b.seq_nt = a->h_nt.data();
	b.seq_p = a->h_p.data();
	b.seq_stride = OUT_ROW;
	b.seq_capacity = n * OUT_ROW;
	rc = pamd_assemble_host(a->dev, &b);
	if (rc != PAMD_OK)
		return rc;
	// map results back: h_pre[i] == -2 pairs take results in order; remember the slot in h_pre as -(k+3)
	for (size_t k = 0; k < n; k++)
		a->h_pre[sent[k]] = -(int) k - 3;
	return PAMD_OK;
}

// min_phred / min_overlapbits on the host, for the case where one of them follows a host-callback module (the kernel can
// only take a leading run of them without changing which module a rejection is blamed on, module.c:128-134)
bool host_module_check(PandaModule mod, const panda_result_seq *res) {
	if (mod->fused_kind == 1) {	// plugin_min_phred.c:8-22
		for (size_t j = 0; j < res->sequence_length; j++)
			if (panda_result_phred(&res->sequence[j]) < mod->fused_phred)
				return false;
		return true;
	}
	return mod->fused_bits * 0.69314718055994530942 <= res->estimated_overlap_probability;	// plugin_min_overlapbits.c:17-23
}

// ---- the host path: algorithm classes whose scoring the kernel does not implement ---------------------------------------
// A class registered by the caller (algo.c:106-120) scores an overlap and a merged base through its own function pointers
// (pandaseq-common.h:594-605), which cannot run on the device.  An assembler given such an algorithm assembles on the caller's
// thread, pair by pair, as assemble_seq() / align() do (assembler.c:48-348), calling those pointers where the reference calls
// them.
const PamdTables &host_tables() {	// (built on first use; afterwards one guard-byte test per call)
	static const PamdTables *const t = [] {
		PamdTables *p = new PamdTables;
		pamd_build_tables(p);
		return p;
	}();
	return *t;
}

inline int host_clampq(int q) {	// PHREDCLAMP, prob.h:23
	return q > 46 ? 46 : (q < 0 ? 0 : q);
}

}	// namespace

// algo.c:27-32: the log probability one overlap position contributes to the merged base, for any algorithm object: asked of
// the class's own function (the seven built-in classes': pamd_host_scorers.cpp).
#ifndef PAMD_WITH_REFERENCE_OBJECTS
double panda_algorithm_quality_compare(PandaAlgorithm algorithm, const panda_qual *a, const panda_qual *b) {
	return panda_algorithm_class(algorithm)->match_probability(panda_algorithm_data(algorithm), (a->nt & b->nt) != 0, a->qual, b->qual);
}
#endif

namespace {

inline unsigned kmer_code(panda_nt nt) {	// misc.h:41
	return nt == 8 ? 3u : (nt == 4 ? 2u : (nt == 2 ? 1u : 0u));
}

// calls emit(kmer, position) for every 8-mer the reference's iterator yields (misc.h:41-45): ascending positions for the
// forward read, descending for the reverse one; an N voids the nine windows that hold it
template <typename Emit>
void host_kmers(const panda_qual *seq, size_t len, bool descending, Emit emit) {
	size_t kmer = 0;
	ptrdiff_t bad = 8;
	for (size_t step = 0; step < len; step++) {
		const size_t p = descending ? len - 1 - step : step;
		kmer = ((kmer << 2) | kmer_code(seq[p].nt)) & 0xFFFFu;
		if (seq[p].nt == (panda_nt) 15)
			bad = 8;
		else if (bad > 0)
			bad--;
		else
			emit(kmer, p);
	}
}

// offset.c:47-90: where the primer ends in `n` bases read through base(i) -> (nt, p, notp); 0 = not found
template <typename Base>
size_t host_primer_offset(double threshold, double penalty, size_t n, Base base, const panda_nt *primer, size_t plen) {
	if (plen > n)
		return 0;
	std::vector<double> pr(plen, -INFINITY);
	double bestpr = exp((double) plen * threshold);
	size_t best = 0;
	for (size_t index = 0; index < n; index++) {
		const double last = exp(pr[index % plen] / (double) (index + 1)) - (double) index * penalty;
		if (last > bestpr) {
			bestpr = last;
			best = index + 1;
		}
		pr[index % plen] = 0;
		panda_nt nt;
		double p, notp;
		base(index, nt, p, notp);
		for (ptrdiff_t x = (ptrdiff_t) (plen > index ? index : plen - 1); x >= 0; x--)
			if (primer[x] != (panda_nt) 15)
				pr[(index - (size_t) x) % plen] += ((nt & primer[x]) != 0) ? p : notp;
	}
	return best;
}

struct host_scratch {
	// the forward positions kept per k-mer (assembler.c:94-99), each entry tagged with the pair it was written for
	// (tag << 16 | position; a stale tag reads as "none"), so the table is cleared once per 65 535 pairs and not per pair
	// (65 536 x num_kmers entries, assembler_support.c:82; sized for the largest num_kmers this thread has seen)
	std::vector<uint32_t> seen = std::vector<uint32_t>(2u << 16, 0);
	uint32_t tag = 0;
	std::vector<uint8_t> bits;
};

// align(), assembler.c:48-250
bool host_align(PandaAssembler a, panda_result_seq &res, host_scratch &sc) {
	// num_kmers (panda_assembler_new_kmer, assembler_support.c:80): the reference LOOKS FOR a free slot at kmerseen[(kmer << 1) + j]
	// (assembler.c:94) and WRITES to kmerseen[kmer * num_kmers + j] (:99) -- one and the same entry only for 2, the default.  For any
	// other value the test reads entries of other k-mers; reproduced as written, on one flat table.  (1: the test's index runs to
	// 2 x 65 535 in a table of 65 536 entries -- the reference reads past its allocation; read here as "free".  0: every k-mer is lost.)
	const size_t nk = a->num_kmers;
	const size_t table = nk << 16;
	const PamdTables &T = host_tables();
	const struct panda_algorithm_class *cls = panda_algorithm_class(a->algo);
	void *data = panda_algorithm_data(a->algo);
	const double qual_nn = cls->prob_unpaired;
	const size_t flen = res.forward_length, rlen = res.reverse_length, fo = res.forward_offset, ro = res.reverse_offset, minov = (size_t) a->minoverlap;
	if (minov + fo >= flen || minov + ro >= rlen)
		return false;
	size_t maxov = flen + rlen - minov - fo - ro - 1;
	if (a->maxoverlap == 0)
		maxov = flen < rlen ? flen : rlen;
	else if (maxov > (size_t) a->maxoverlap)
		maxov = (size_t) a->maxoverlap;
	const size_t size = minov <= maxov ? maxov - minov + 1 : 1;
	sc.bits.assign(size, 0);
	if (sc.seen.size() < table) {
		sc.seen.assign(table, 0u);
		sc.tag = 0;
	}
	if (++sc.tag == 0x10000u) {
		std::fill(sc.seen.begin(), sc.seen.end(), 0u);
		sc.tag = 1;
	}
	const uint32_t tag = sc.tag << 16;
	// (a k-mer ends at position >= 8, so a stored position is never 0 and a zeroed entry never carries a live tag)
	host_kmers(res.forward, flen, false, [&](size_t k, size_t p) {
		size_t j = 0;
		for (; j < nk; j++) {
			const size_t probe = (k << 1) + j;
			if (probe >= table || (sc.seen[probe] ^ tag) >= 0x10000u)
				break;
		}
		if (j < nk)
			sc.seen[k * nk + j] = tag | (uint32_t) p;
	});
	bool any = false;
	host_kmers(res.reverse, rlen, true, [&](size_t k, size_t p) {
		for (size_t j = 0; j < nk && (sc.seen[k * nk + j] ^ tag) < 0x10000u; j++) {
			const ptrdiff_t index = (ptrdiff_t) (flen + rlen) - (ptrdiff_t) p - (ptrdiff_t) (sc.seen[k * nk + j] & 0xFFFFu) - (ptrdiff_t) minov - 1;
			if (index >= 0 && (size_t) index < size) {
				sc.bits[(size_t) index] = 1;
				any = true;
			}
		}
	});
	double best = qual_nn * (double) (flen + rlen);
	ptrdiff_t bo = -1;
	res.overlaps_examined = 0;
	for (size_t c = 0; c < size; c++) {
		if (any && !sc.bits[c])
			continue;
		const size_t overlap = c + minov;
		const double pr = cls->overlap_probability(data, res.forward, flen, res.reverse, rlen, overlap);
		if (pr > best) {
			best = pr;
			bo = (ptrdiff_t) overlap;
		}
		res.overlaps_examined++;
	}
	if (res.overlaps_examined == maxov - minov + 1)
		a->slowcount++;
	if (bo < 0)
		return false;
	const ptrdiff_t len = (ptrdiff_t) flen - (ptrdiff_t) fo - bo + (ptrdiff_t) rlen - (ptrdiff_t) ro + 1;
	if (len <= 0 || (size_t) len > 2 * PAMD_MAX_LEN)
		return false;
	res.sequence = a->result_seq;
	res.sequence_length = (size_t) len - 1;
	res.degenerates = 0;
	res.overlap_mismatches = 0;
	const ptrdiff_t df = (ptrdiff_t) flen - (ptrdiff_t) fo - bo, dr = (ptrdiff_t) rlen - (ptrdiff_t) ro - bo;
	const ptrdiff_t vdf = df < 0 ? 0 : df, wdf = df > 0 ? 0 : df, vdr = dr < 0 ? 0 : dr, wdr = dr > 0 ? 0 : dr;
	// PANDA_NT_IS_DEGN (pandaseq-nt.h:55): anything but exactly one of the four bits -- nt 0 counts as degenerate
	auto degenerate = [](panda_nt nt) {	// (the macro's multiply-and-modulo counts the set bits of a 14-bit value; a panda_nt has four)
		const unsigned v = (unsigned int) nt;
		return v < 16u ? __builtin_popcount(v) != 1 : ((v * 0x200040008001ULL & 0x111111111111111ULL) % 0xf) != 1;
	};
	double fq = 0, oq = 0, rq = 0;
	for (ptrdiff_t i = 0; i < vdf; i++) {
		const panda_qual &b = res.forward[(size_t) i + fo];
		const double q = T.score[host_clampq(b.qual)];
		res.sequence[i].nt = b.nt;
		res.sequence[i].p = q;
		res.degenerates += degenerate(b.nt);
		fq += q;
	}
	size_t uf = flen, ur = rlen;	// the B-cliff: trailing runs of quality 2 (assembler.c:176-177)
	while (uf > 0 && res.forward[uf - 1].qual == (char) 2)
		uf--;
	while (ur > 0 && res.reverse[ur - 1].qual == (char) 2)
		ur--;
	for (ptrdiff_t i = 0; i < bo + wdf + wdr; i++) {
		const ptrdiff_t index = vdf + i, fi = (ptrdiff_t) fo + vdf + i, ri = (ptrdiff_t) rlen - i - 1 + wdf;
		if (index < 0 || fi < 0 || ri < 0 || (size_t) fi >= flen || (size_t) ri >= rlen)
			continue;
		const panda_qual &f = res.forward[fi], &r = res.reverse[ri];
		const bool match = (r.nt & f.nt) != 0;
		if (!match)
			res.overlap_mismatches++;
		double q;
		if ((size_t) fi >= uf && (size_t) ri >= ur)
			q = qual_nn;
		else if ((size_t) fi >= uf)
			q = T.score[host_clampq(r.qual)];
		else if ((size_t) ri >= ur)
			q = T.score[host_clampq(f.qual)];
		else
			q = cls->match_probability(data, match, f.qual, r.qual);
		const panda_nt nt = match ? (panda_nt) (r.nt & f.nt) : (f.qual < r.qual ? r.nt : f.nt);
		res.sequence[index].nt = nt;
		res.sequence[index].p = q;
		res.degenerates += degenerate(nt);
		oq += q;
	}
	for (ptrdiff_t i = 0; i < vdr; i++) {
		const ptrdiff_t index = df + bo + i;
		const panda_qual &b = res.reverse[(size_t) ((ptrdiff_t) rlen - bo - i - 1)];
		const double q = T.score[host_clampq(b.qual)];
		rq += q;
		res.sequence[index].nt = b.nt;
		res.sequence[index].p = q;
		res.degenerates += degenerate(b.nt);
	}
	res.quality = (fq + rq + oq) / (double) len;
	res.overlap = (size_t) bo;
	res.estimated_overlap_probability = best;
	return true;
}

// assemble_seq(), assembler.c:252-348
const panda_result_seq *host_assemble(PandaAssembler a, const panda_seq_identifier *id, const panda_qual *f, size_t fl, const panda_qual *r, size_t rl) {
	static thread_local host_scratch sc;
	const PamdTables &T = host_tables();
	panda_result_seq &res = a->result;
	if (id != &res.name)
		res.name = *id;
	res.forward = f;
	res.forward_length = fl;
	res.reverse = r;
	res.reverse_length = rl;
	a->count++;
	if (fl < 2 || rl < 2) {
		a->badreadcount++;
		return nullptr;
	}
	for (size_t m = 0; m < a->modules.size(); m++) {	// module_precheckseq, module.c:138-154
		PandaModule mod = a->modules[m];
		if (mod->precheck != nullptr && !mod->precheck(a->logger, &res.name, f, fl, r, rl, mod->user_data)) {
			a->rejected[m]++;
			return nullptr;
		}
	}
	auto read_base = [&T](const panda_qual *seq) {
		return [seq, &T](size_t i, panda_nt &nt, double &p, double &notp) {	// qual_base_score, offset.c:92-101
			const int q = host_clampq(seq[i].qual);
			nt = seq[i].nt;
			p = T.score[q];
			notp = T.score_err[q];
		};
	};
	res.forward_offset = res.reverse_offset = 0;
	if (!a->post_primers) {
		if (a->forward_primer_length > 0) {
			res.forward_offset = host_primer_offset(a->threshold, a->primer_penalty, fl, read_base(f), a->forward_primer, a->forward_primer_length);
			if (res.forward_offset == 0) {
				a->nofpcount++;
				return nullptr;
			}
			res.forward_offset--;
		} else {
			res.forward_offset = a->forward_trim;
		}
		if (a->reverse_primer_length > 0) {
			res.reverse_offset = host_primer_offset(a->threshold, a->primer_penalty, rl, read_base(r), a->reverse_primer, a->reverse_primer_length);
			if (res.reverse_offset == 0) {
				a->norpcount++;
				return nullptr;
			}
			res.reverse_offset--;
		} else {
			res.reverse_offset = a->reverse_trim;
		}
	}
	if ((fl < rl ? fl : rl) < (size_t) a->minoverlap) {
		a->badreadcount++;
		return nullptr;
	}
	if (!host_align(a, res, sc)) {
		if (a->noalgn != nullptr)
			a->noalgn(a, &res.name, f, fl, r, rl, a->noalgn_data);
		a->noalgncount++;
		return nullptr;
	}
	if (a->post_primers) {	// assembler.c:300-333: the primers are looked for in the merged sequence, the reverse one from its end
		const panda_result *seq = res.sequence;
		const size_t n = res.sequence_length;
		auto merged_base = [seq, n](bool reverse) {
			return [seq, n, reverse](size_t i, panda_nt &nt, double &p, double &notp) {	// result_base_score, offset.c:114-122
				const panda_result &b = seq[reverse ? n - i - 1 : i];
				nt = b.nt;
				p = b.p;
				notp = (b.p > M_LN2) ? log1p(-exp(-b.p)) : log(-expm1(-b.p));	// panda_log1mexp, offset.c:35-38
			};
		};
		if (a->forward_primer_length > 0) {
			res.forward_offset = host_primer_offset(a->threshold, a->primer_penalty, n, merged_base(false), a->forward_primer, a->forward_primer_length);
			if (res.forward_offset == 0) {
				a->nofpcount++;
				return nullptr;
			}
			res.forward_offset--;
		} else {
			res.forward_offset = a->forward_trim;
		}
		if (a->reverse_primer_length > 0) {
			res.reverse_offset = host_primer_offset(a->threshold, a->primer_penalty, n, merged_base(true), a->reverse_primer, a->reverse_primer_length);
			if (res.reverse_offset == 0) {
				a->norpcount++;
				return nullptr;
			}
			res.reverse_offset--;
		} else {
			res.reverse_offset = a->reverse_trim;
		}
		if (res.sequence_length <= res.forward_offset + res.reverse_offset) {
			a->nofpcount++;
			return nullptr;
		}
		res.sequence_length -= res.forward_offset + res.reverse_offset;
		for (size_t it = 0; it < res.sequence_length; it++)
			res.sequence[it] = res.sequence[it + res.forward_offset];
	}
	if (res.quality < a->threshold) {
		a->lowqcount++;
		return nullptr;
	}
	for (size_t m = 0; m < a->modules.size(); m++) {	// module_checkseq, module.c:124-136
		PandaModule mod = a->modules[m];
		if (mod->fused_kind != 0 ? !host_module_check(mod, &res) : (mod->check != nullptr && !mod->check(a->logger, &res, mod->user_data))) {
			a->rejected[m]++;
			return nullptr;
		}
	}
	a->okcount++;
	a->overlapcount[res.overlap]++;
	if (a->longest_overlap < res.overlap)
		a->longest_overlap = res.overlap;
	return &res;
}

// Replay staged pair `i` the way assemble_seq() would have (assembler.c:252-348).  Returns the
// result pointer or NULL if the pair was rejected.  `f`/`r` are the caller-visible read buffers.
const panda_result_seq *replay(PandaAssembler a, size_t i, const panda_qual *f, const panda_qual *r) {
	a->count++;
	const int pre = a->h_pre[i];
	if (pre == -1) {
		a->badreadcount++;
		return nullptr;
	}
	if (pre >= 0) {
		a->rejected[pre]++;
		return nullptr;
	}
	const size_t k = (size_t) (-pre - 3);
	const pamd_pair_result &d = a->h_res[k];
	panda_result_seq &res = a->result;
	res.name = a->h_id[i];
	res.forward = f;
	res.reverse = r;
	res.forward_length = a->h_flen[k];
	res.reverse_length = a->h_rlen[k];
	res.forward_offset = d.fwd_offset;
	res.reverse_offset = d.rev_offset;
	res.overlaps_examined = d.examined;
	if (d.slow)
		a->slowcount++;
	switch (d.status) {
	case PAMD_ST_BADR:
		a->badreadcount++;
		return nullptr;
	case PAMD_ST_NOFP:
		a->nofpcount++;
		return nullptr;
	case PAMD_ST_NORP:
		a->norpcount++;
		return nullptr;
	case PAMD_ST_NOALGN:
		if (a->noalgn != nullptr)	// assembler.c:294-296
			a->noalgn(a, &res.name, f, res.forward_length, r, res.reverse_length, a->noalgn_data);
		a->noalgncount++;
		return nullptr;
	default:
		break;
	}
	res.quality = d.quality;
	res.degenerates = d.degenerates;
	res.sequence = a->result_seq;
	res.sequence_length = d.seq_len;
	res.overlap_mismatches = d.mismatches;
	res.overlap = d.overlap;
	res.estimated_overlap_probability = d.est_overlap_prob;
	if (d.status == PAMD_ST_LOWQ) {
		a->lowqcount++;
		return nullptr;
	}
	if (d.status == PAMD_ST_REJECT) {
		a->rejected[d.reject_module]++;
		return nullptr;
	}
	// accepted by the kernel: materialise panda_result[] (16 B/base AoS) from the two device arrays
	const uint8_t *nt = &a->h_nt[d.seq_offset];
	const double *p = &a->h_p[d.seq_offset];
	for (size_t j = 0; j < d.seq_len; j++) {
		a->result_seq[j].nt = (panda_nt) nt[j];
		a->result_seq[j].p = p[j];
	}
	// remaining (host-callback) check modules, in order (module.c:124-136)
	for (size_t m = a->fused_prefix; m < a->modules.size(); m++) {
		PandaModule mod = a->modules[m];
		if (mod->fused_kind != 0 ? !host_module_check(mod, &res) : (mod->check != nullptr && !mod->check(a->logger, &res, mod->user_data))) {
			a->rejected[m]++;
			return nullptr;
		}
	}
	a->okcount++;
	a->overlapcount[res.overlap]++;
	if (a->longest_overlap < res.overlap)
		a->longest_overlap = res.overlap;
	return &res;
}

// Stage one pair at slot i; decides what assemble_seq() decides before align() on the host side
// only where it must (length < 2, precheck callbacks -- assembler.c:255-261).
void stage_pair(PandaAssembler a, size_t i, const panda_seq_identifier *id, const panda_qual *f, size_t flen, const panda_qual *r, size_t rlen) {
	a->h_id[i] = *id;
	a->h_flen[i] = (uint16_t) flen;
	a->h_rlen[i] = (uint16_t) rlen;
	memcpy(&a->h_fwd[i * ROW], f, flen * sizeof(panda_qual));
	memcpy(&a->h_rev[i * ROW], r, rlen * sizeof(panda_qual));
	int pre = -2;
	if (flen < 2 || rlen < 2) {
		pre = -1;
	} else {
		for (size_t m = 0; m < a->modules.size(); m++) {	// module_precheckseq, module.c:138-154
			PandaModule mod = a->modules[m];
			if (mod->precheck != nullptr && !mod->precheck(a->logger, id, f, flen, r, rlen, mod->user_data)) {
				pre = (int) m;
				break;
			}
		}
	}
	a->h_pre[i] = pre;
}

// ---- the pipelined source path: panda_assembler_next() ------------------------------------------------------------
// The reference overlaps reading with assembly through its prefetching reader thread (async.c:103-149) and hands a
// worker one pair at a time (mux.c:132-161).  Here two batches are in flight: while the caller consumes the results of
// batch k, batch k+1 is uploaded from pinned memory, assembled, its log-probabilities turned into table indices
// (pamd_encode_rows_device: 3 bytes per base cross PCIe instead of 9), downloaded into pinned memory and expanded into
// panda_result[] rows by a few helper threads.  The caller's thread only ever talks to the source (PandaNextSeq and the
// module callbacks are the embedder's code, called from the thread that calls panda_assembler_next, in source order),
// copies each pair's reads once, and fills in the record it returns.
}	// namespace

struct pamd_slot {
	bool lazy = false;	// this batch's rows are built on the caller's thread, record by record (compat_lazy_rows)
	// pinned host memory
	pamd_qual *h_fwd = nullptr, *h_rev = nullptr;
	uint16_t *h_flen = nullptr, *h_rlen = nullptr;
	pamd_pair_result *h_res = nullptr;
	uint8_t *h_nt = nullptr;
	uint16_t *h_idx = nullptr;
	// device memory
	pamd_qual *d_fwd = nullptr, *d_rev = nullptr;
	uint16_t *d_flen = nullptr, *d_rlen = nullptr, *d_idx = nullptr;
	pamd_pair_result *d_res = nullptr;
	uint8_t *d_nt = nullptr;
	double *d_p = nullptr;
	int64_t *d_ctr = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t done_ev = nullptr;	// recorded behind a batch's last copy; waited for without spinning
	size_t cap = 0;		// pairs all of the above are sized for (rows of ROW bases in, OUT_ROW bases out)
	// the batch
	// what is remembered of every pair pulled from the source, compactly (the caller's thread streams through all of
	// it twice, once pulling and once replaying: a 480-byte record per pair was most of what it moved): the identifier's
	// four numbers and four strings back to back, the read lengths
	std::vector<char> ids;	// the batch's packed identifiers ...
	size_t ids_used = 0;	// ... and how much of the buffer they take
	std::vector<uint32_t> id_at;
	std::vector<uint16_t> len_f, len_r;
	std::vector<int> pre;			// -1 short read, >= 0 rejected by precheck module k, <= -3: device pair -(k+3)
	panda_result *rows = nullptr;		// panda_result[] of the pairs the kernel accepted, back to back (never zero-filled)
	size_t rows_cap = 0;
	std::vector<size_t> row_off;		// per device pair: where its row starts in `rows`
	size_t n = 0, n_dev = 0, pos = 0;
	size_t stride = 256;	// bases per staged read row: 256 until a longer read arrives, then ROW
	size_t seq_stride = 512;
	pamd_assembler *dev = nullptr;
	std::future<int> done;
	std::string error;
	double seconds = 0, sec_up = 0, sec_gpu = 0, sec_down = 0, sec_expand = 0;
};

namespace {

// A read into its pinned staging row.  (Plain memcpy: streaming stores, 16 and 32 bytes wide, are faster in isolation --
// 16 ns against 25 ns per pair on the GPU box, tools/probe/stage_copy_probe.cpp -- and twice as slow in the pipeline, where
// the rows' odd tails and the other threads' traffic break their write combining: measured twice, 8.3 M pairs/s against 15.)
void stage_copy(pamd_qual *dst, const panda_qual *src, size_t bases) {
	memcpy(dst, src, bases * sizeof(panda_qual));
}

double now_s() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void slot_free(pamd_slot *s) {
	if (s == nullptr)
		return;
	if (s->done.valid())
		s->done.wait();
	for (void *q : { (void *) s->h_fwd, (void *) s->h_rev, (void *) s->h_flen, (void *) s->h_rlen, (void *) s->h_res, (void *) s->h_nt, (void *) s->h_idx })
		if (q)
			(void) hipHostFree(q);
	for (void *q : { (void *) s->d_fwd, (void *) s->d_rev, (void *) s->d_flen, (void *) s->d_rlen, (void *) s->d_idx, (void *) s->d_res, (void *) s->d_nt, (void *) s->d_p, (void *) s->d_ctr })
		if (q)
			(void) hipFree(q);
	if (s->stream)
		(void) hipStreamDestroy(s->stream);
	if (s->done_ev)
		(void) hipEventDestroy(s->done_ev);
	free(s->rows);
	delete s;
}

bool slot_reserve(PandaAssembler a, pamd_slot *s, size_t pairs) {
	if (s->cap >= pairs)
		return true;
	if (hipSetDevice(a->device) != hipSuccess)
		return false;
	bool ok = true;
	auto pin = [&](auto &ptr, size_t bytes) {
		void *q = nullptr;
		ok = ok && hipHostMalloc(&q, bytes, hipHostMallocNonCoherent) == hipSuccess;	// host-cached; every hand-over is a stream synchronise
		ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(q);
	};
	auto dev = [&](auto &ptr, size_t bytes) {
		void *q = nullptr;
		ok = ok && hipMalloc(&q, bytes) == hipSuccess;
		ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(q);
	};
	// free what is there (no batch is in flight on a slot that is being refilled)
	for (void *q : { (void *) s->h_fwd, (void *) s->h_rev, (void *) s->h_flen, (void *) s->h_rlen, (void *) s->h_res, (void *) s->h_nt, (void *) s->h_idx })
		if (q)
			(void) hipHostFree(q);
	for (void *q : { (void *) s->d_fwd, (void *) s->d_rev, (void *) s->d_flen, (void *) s->d_rlen, (void *) s->d_idx, (void *) s->d_res, (void *) s->d_nt, (void *) s->d_p, (void *) s->d_ctr })
		if (q)
			(void) hipFree(q);
	pin(s->h_fwd, pairs * ROW * sizeof(pamd_qual));
	pin(s->h_rev, pairs * ROW * sizeof(pamd_qual));
	pin(s->h_flen, pairs * 2);
	pin(s->h_rlen, pairs * 2);
	pin(s->h_res, pairs * sizeof(pamd_pair_result));
	pin(s->h_nt, pairs * OUT_ROW);
	pin(s->h_idx, pairs * OUT_ROW * 2);
	dev(s->d_fwd, pairs * ROW * sizeof(pamd_qual));
	dev(s->d_rev, pairs * ROW * sizeof(pamd_qual));
	dev(s->d_flen, pairs * 2);
	dev(s->d_rlen, pairs * 2);
	dev(s->d_res, pairs * sizeof(pamd_pair_result));
	dev(s->d_nt, pairs * OUT_ROW + 16);
	dev(s->d_p, (pairs * OUT_ROW + 16) * sizeof(double));
	dev(s->d_idx, (pairs * OUT_ROW + 16) * 2);
	dev(s->d_ctr, PAMD_NUM_COUNTERS * 8);
	if (s->stream == nullptr)
		ok = ok && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
	if (s->done_ev == nullptr)
		ok = ok && hipEventCreateWithFlags(&s->done_ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
	s->cap = ok ? pairs : 0;
	return ok;
}

// What the background thread of a batch does: upload, assemble, encode, download, expand.
int slot_run(PandaAssembler a, pamd_slot *s) {
#define SLOT_TRY(call)                                                          \
	do {                                                                    \
		hipError_t e_ = (call);                                         \
		if (e_ != hipSuccess) {                                         \
			s->error = std::string(#call ": ") + hipGetErrorString(e_); \
			return PAMD_ENODEVICE;                                  \
		}                                                               \
	} while (0)
	const size_t n = s->n_dev;
	if (n == 0)
		return PAMD_OK;
	const double t_begin = now_s();
	SLOT_TRY(hipSetDevice(a->device));
	const size_t st = s->stride, ss = s->seq_stride;
	SLOT_TRY(hipMemcpyAsync(s->d_fwd, s->h_fwd, n * st * sizeof(pamd_qual), hipMemcpyHostToDevice, s->stream));
	SLOT_TRY(hipMemcpyAsync(s->d_rev, s->h_rev, n * st * sizeof(pamd_qual), hipMemcpyHostToDevice, s->stream));
	SLOT_TRY(hipMemcpyAsync(s->d_flen, s->h_flen, n * 2, hipMemcpyHostToDevice, s->stream));
	SLOT_TRY(hipMemcpyAsync(s->d_rlen, s->h_rlen, n * 2, hipMemcpyHostToDevice, s->stream));
	pamd_batch b;
	memset(&b, 0, sizeof b);
	b.n_pairs = n;
	b.fwd = s->d_fwd;
	b.rev = s->d_rev;
	b.fwd_stride = b.rev_stride = st;
	b.flen = s->d_flen;
	b.rlen = s->d_rlen;
	b.results = s->d_res;
	b.seq_nt = s->d_nt;
	b.seq_p = s->d_p;
	b.seq_stride = ss;	// strided rows: what comes back is n * ss bases, no cursor to wait for
	b.seq_capacity = n * ss;
	b.counters = nullptr;	// the counters are the caller's, kept in source order by replay
	int rc = pamd_assemble_device(s->dev, &b, s->stream);
	if (rc == PAMD_OK)
		rc = pamd_encode_rows_device(s->dev, s->d_p, n * ss, s->d_idx, s->stream);
	if (rc != PAMD_OK) {
		s->error = pamd_last_error();
		return rc;
	}
	SLOT_TRY(hipMemcpyAsync(s->h_res, s->d_res, n * sizeof(pamd_pair_result), hipMemcpyDeviceToHost, s->stream));
	SLOT_TRY(hipMemcpyAsync(s->h_nt, s->d_nt, n * ss, hipMemcpyDeviceToHost, s->stream));
	SLOT_TRY(hipMemcpyAsync(s->h_idx, s->d_idx, n * ss * 2, hipMemcpyDeviceToHost, s->stream));
	// (a blocking event: hipStreamSynchronize spins, and with one assembler per thread -- pool.c:140-167 -- there are four such
	// waits per assembler competing with the callers' threads for the CPUs)
	SLOT_TRY(hipEventRecord(s->done_ev, s->stream));
	SLOT_TRY(hipEventSynchronize(s->done_ev));
#undef SLOT_TRY
	// expand the accepted rows into the reference's 16-byte panda_result (pandaseq-common.h:221-230): nt from the
	// device, p = the table entry the index names -- the very double the kernel wrote
	s->row_off.resize(n + 1);
	size_t total = 0;
	for (size_t k = 0; k < n; k++) {
		s->row_off[k] = total;
		if (s->h_res[k].status == PAMD_ST_OK)
			total += s->h_res[k].seq_len;
	}
	s->row_off[n] = total;
	s->lazy = compat_lazy_rows() != 0;
	if (s->lazy) {	// the rows are built record by record on the caller's thread (slot_replay)
		s->seconds = now_s() - t_begin;
		return PAMD_OK;
	}
	if (s->rows_cap < total) {
		free(s->rows);
		s->rows_cap = total + total / 8 + 4096;
		s->rows = static_cast<panda_result *>(malloc(s->rows_cap * sizeof(panda_result)));
		if (s->rows == nullptr) {
			s->rows_cap = 0;
			s->error = "out of host memory for the merged rows";
			return PAMD_ENOMEM;
		}
	}
	const double *image = a->image.data();
	const size_t n_image = a->image.size();
	// the expansion writes 16 bytes per base: memory-bound, it scales with threads -- those this process may run on (its
	// affinity mask and its cgroup's CPU quota, not the machine's count), shared between the assemblers that are streaming
	// (pool.c:140-167 runs one per thread)
	const int streaming = g_streaming.load();
	unsigned helpers = effective_cpus() / (2u * (unsigned) (streaming < 1 ? 1 : streaming));
	helpers = helpers < 1 ? 1 : (helpers > 32 ? 32 : helpers);
	if (n < 4096)
		helpers = 1;
	std::atomic<int> bad{0};
	static const bool plain_rows = getenv("PAMD_COMPAT_PLAIN_ROWS") != nullptr;
	auto expand = [&](size_t lo, size_t hi) {
		for (size_t k = lo; k < hi; k++) {
			const pamd_pair_result &d = s->h_res[k];
			if (d.status != PAMD_ST_OK)
				continue;
			const uint8_t *nt = s->h_nt + k * ss;
			const uint16_t *ix = s->h_idx + k * ss;
			panda_result *out = s->rows + s->row_off[k];
			for (size_t j = 0; j < d.seq_len; j++) {
				if (ix[j] >= n_image) {
					bad.store(1);
					continue;
				}
				// one 16-byte streaming store per base (nt, padding, p): the rows are written once, 6 KB per pair, and read --
				// if at all -- much later by another thread; through the caches they would cost a read of every line first
				// and evict what the caller's thread is working with
				long long bits;
				memcpy(&bits, &image[ix[j]], sizeof bits);
				if (plain_rows) {	// (PAMD_COMPAT_PLAIN_ROWS=1: A/B measurements only)
					out[j].nt = (panda_nt) nt[j];
					out[j].p = image[ix[j]];
				} else {
					_mm_stream_si128(reinterpret_cast<__m128i *>(out + j), _mm_set_epi64x(bits, (long long) nt[j]));
				}
			}
		}
	};
	static_assert(sizeof(panda_result) == 16 && offsetof(panda_result, nt) == 0 && offsetof(panda_result, p) == 8, "panda_result layout (pandaseq-common.h:221-230)");
	std::vector<std::thread> team;
	for (unsigned t = 1; t < helpers; t++)
		team.emplace_back([&expand, n, t, helpers] {
			expand(n * t / helpers, n * (t + 1) / helpers);
			_mm_sfence();
		});
	expand(0, n / helpers);
	_mm_sfence();
	for (std::thread &t : team)
		t.join();
	if (bad.load()) {
		s->error = "a merged log-probability is not an entry of the table image";
		return PAMD_EINVAL;
	}
	s->seconds = now_s() - t_begin;
	return PAMD_OK;
}

void slot_wait(pamd_slot *s);

// Pull the next batch from the source into slot `s` (the caller's thread) and start it.
void slot_fill_and_submit(PandaAssembler a, pamd_slot *s, size_t batch_pairs = 0) {
	s->n = s->n_dev = s->pos = 0;
	if (a->eof)
		return;
	const size_t want = batch_pairs != 0 ? batch_pairs : g_batch_pairs.load();
	if (!slot_reserve(a, s, want)) {
		fprintf(stderr, "pandaseq_amd: cannot allocate the staging buffers of a %zu-pair batch\n", want);
		abort();
	}
	if (s->pre.size() < want) {
		s->pre.resize(want);
		s->id_at.resize(want);
		s->len_f.resize(want);
		s->len_r.resize(want);
	}
	s->ids_used = 0;
	if (s->ids.size() < want * 40)
		s->ids.resize(want * 40);
	s->stride = 256;
	panda_result_seq &r = a->result;	// the source fills in the identifier here (cache-hot); it is packed away below
	while (s->n < want) {
		const panda_qual *f = nullptr, *rv = nullptr;
		size_t fl = 0, rl = 0;
		if (!a->next(&r.name, &f, &fl, &rv, &rl, a->next_data)) {
			a->eof = true;
			break;
		}
		if (fl > PAMD_MAX_LEN || rl > PAMD_MAX_LEN) {	// assembler.c:359-360
			fprintf(stderr, "pandaseq_amd: read longer than PANDA_MAX_LEN\n");
			abort();
		}
		int pre = -2;
		if (fl < 2 || rl < 2) {
			pre = -1;
		} else {
			for (size_t m = 0; m < a->modules.size(); m++) {	// module_precheckseq, module.c:138-154
				PandaModule mod = a->modules[m];
				if (mod->precheck != nullptr && !mod->precheck(a->logger, &r.name, f, fl, rv, rl, mod->user_data)) {
					pre = (int) m;
					break;
				}
			}
		}
		if (pre == -2) {
			if ((fl > 256 || rl > 256) && s->stride == 256) {	// first long read of the batch: re-lay the rows, last first
				for (size_t k = s->n_dev; k-- > 0;) {
					memmove(s->h_fwd + k * ROW, s->h_fwd + k * 256, 256 * sizeof(pamd_qual));
					memmove(s->h_rev + k * ROW, s->h_rev + k * 256, 256 * sizeof(pamd_qual));
				}
				s->stride = ROW;
			}
			const size_t k = s->n_dev++;
			stage_copy(s->h_fwd + k * s->stride, f, fl);
			stage_copy(s->h_rev + k * s->stride, rv, rl);
			s->h_flen[k] = (uint16_t) fl;
			s->h_rlen[k] = (uint16_t) rl;
			pre = -(int) k - 3;
		}
		{	// pack the identifier: lane, tile, x, y, then the four strings with their terminators (straight into the
			// buffer: room for the longest identifier is kept ahead of the write position)
			const panda_seq_identifier &id = r.name;
			constexpr size_t LONGEST = 16 + sizeof id.instrument + sizeof id.run + sizeof id.flowcell + sizeof id.tag;
			const size_t at = s->ids_used;
			if (s->ids.size() < at + LONGEST)
				s->ids.resize((at + LONGEST) * 2);
			char *const base = s->ids.data() + at;
			char *o = base;
			memcpy(o, &id.lane, 16);	// lane, tile, x, y are adjacent ints (pandaseq-common.h:248-262)
			o += 16;
			auto put = [&o](const char *from, size_t room) {	// a string and its terminator (at most room - 1 characters)
				size_t k = 0;
				while (k + 1 < room && from[k] != 0) {
					o[k] = from[k];
					k++;
				}
				o[k] = 0;
				o += k + 1;
			};
			put(id.instrument, sizeof id.instrument);
			put(id.run, sizeof id.run);
			put(id.flowcell, sizeof id.flowcell);
			put(id.tag, sizeof id.tag);
			s->ids_used = at + (size_t) (o - base);
			s->id_at[s->n] = (uint32_t) at;
		}
		s->len_f[s->n] = (uint16_t) fl;
		s->len_r[s->n] = (uint16_t) rl;
		s->pre[s->n] = pre;
		s->n++;
	}
	s->seq_stride = s->stride == 256 ? 512 : OUT_ROW;
	if (a->dev_dirty) {
		// a setter was called after batches were prefetched: those (up to NSLOT x the batch size pairs) were assembled under
		// the old configuration, as documented; their background work reads the table image and the device assemblers that
		// push_config() rewrites, so it has to be over first
		for (pamd_slot *other : a->slots)
			if (other != nullptr && other != s)
				slot_wait(other);
	}
	if (push_config(a) != PAMD_OK) {
		fprintf(stderr, "pandaseq_amd: %s\n", pamd_last_error());
		abort();	// no CPU fallback: a device failure is fatal, never silently wrong
	}
	if (s->n_dev > 0)
		s->done = std::async(std::launch::async, slot_run, a, s);
}

void slot_wait(pamd_slot *s) {
	if (!s->done.valid())
		return;
	const int rc = s->done.get();
	if (rc != PAMD_OK) {
		fprintf(stderr, "pandaseq_amd: %s\n", s->error.c_str());
		abort();
	}
}

// assemble_seq()'s bookkeeping for pair i of a finished batch (assembler.c:252-348), in source order
const panda_result_seq *slot_replay(PandaAssembler a, pamd_slot *s, size_t i) {
	a->count++;
	const int pre = s->pre[i];
	if (pre == -1) {
		a->badreadcount++;
		return nullptr;
	}
	if (pre >= 0) {
		a->rejected[pre]++;
		return nullptr;
	}
	const size_t k = (size_t) (-pre - 3);
	const pamd_pair_result &d = s->h_res[k];
	panda_result_seq &res = a->result;	// one record, valid until the next call (pandaseq-assembler.h:172-173)
	{	// unpack the identifier
		const char *o = s->ids.data() + s->id_at[i];
		memcpy(&res.name.lane, o, 16);
		o += 16;
		size_t l = strlen(o);
		memcpy(res.name.instrument, o, l + 1);
		o += l + 1;
		l = strlen(o);
		memcpy(res.name.run, o, l + 1);
		o += l + 1;
		l = strlen(o);
		memcpy(res.name.flowcell, o, l + 1);
		o += l + 1;
		l = strlen(o);
		memcpy(res.name.tag, o, l + 1);
	}
	res.forward_length = s->len_f[i];
	res.reverse_length = s->len_r[i];
	res.forward = reinterpret_cast<const panda_qual *>(s->h_fwd + k * s->stride);
	res.reverse = reinterpret_cast<const panda_qual *>(s->h_rev + k * s->stride);
	res.forward_offset = d.fwd_offset;
	res.reverse_offset = d.rev_offset;
	res.overlaps_examined = d.examined;
	if (d.slow)
		a->slowcount++;
	switch (d.status) {
	case PAMD_ST_BADR:
		a->badreadcount++;
		return nullptr;
	case PAMD_ST_NOFP:
		a->nofpcount++;
		return nullptr;
	case PAMD_ST_NORP:
		a->norpcount++;
		return nullptr;
	case PAMD_ST_NOALGN:
		if (a->noalgn != nullptr)	// assembler.c:294-296
			a->noalgn(a, &res.name, res.forward, res.forward_length, res.reverse, res.reverse_length, a->noalgn_data);
		a->noalgncount++;
		return nullptr;
	default:
		break;
	}
	res.quality = d.quality;
	res.degenerates = d.degenerates;
	res.sequence = s->lazy ? a->result_seq : s->rows + s->row_off[k];
	res.sequence_length = d.seq_len;
	res.overlap_mismatches = d.mismatches;
	res.overlap = d.overlap;
	res.estimated_overlap_probability = d.est_overlap_prob;
	if (d.status == PAMD_ST_LOWQ) {
		a->lowqcount++;
		return nullptr;
	}
	if (d.status == PAMD_ST_REJECT) {
		a->rejected[d.reject_module]++;
		return nullptr;
	}
	if (s->lazy) {	// nt from the device, p = the table entry the index names -- the very double the kernel wrote
		const uint8_t *nt = s->h_nt + k * s->seq_stride;
		const uint16_t *ix = s->h_idx + k * s->seq_stride;
		const double *image = a->image.data();
		const size_t n_image = a->image.size();
		for (size_t j = 0; j < d.seq_len; j++) {
			if (ix[j] >= n_image) {
				fprintf(stderr, "pandaseq_amd: a merged log-probability is not an entry of the table image\n");
				abort();
			}
			a->result_seq[j].nt = (panda_nt) nt[j];
			a->result_seq[j].p = image[ix[j]];
		}
	}
	for (size_t m = a->fused_prefix; m < a->modules.size(); m++) {	// host-callback check modules, in order (module.c:124-136)
		PandaModule mod = a->modules[m];
		if (mod->fused_kind != 0 ? !host_module_check(mod, &res) : (mod->check != nullptr && !mod->check(a->logger, &res, mod->user_data))) {
			a->rejected[m]++;
			return nullptr;
		}
	}
	a->okcount++;
	a->overlapcount[res.overlap]++;
	if (a->longest_overlap < res.overlap)
		a->longest_overlap = res.overlap;
	return &res;
}

void ensure_staging(PandaAssembler a, size_t pairs) {
	if (a->h_flen.size() < pairs) {
		a->h_fwd.resize(pairs * ROW);
		a->h_rev.resize(pairs * ROW);
		a->h_flen.resize(pairs);
		a->h_rlen.resize(pairs);
		a->h_id.resize(pairs);
		a->h_pre.resize(pairs);
	}
}

}	// namespace

extern "C" {

#ifndef PAMD_WITH_REFERENCE_OBJECTS	// (nt.c, offset.c, seqid.c, misc.c, proxy.c stay in a link with the reference's objects)
// ---- per-base helpers (nt.c:120-150) ----
// Plugins compiled against the reference headers call these on the panda_result rows they are
// handed (plugin_min_phred.c:16).  The bisection keeps the reference's quirks: 1 at or below
// qual_score[0] (= -2), "mid == lower" returns lower.
double panda_quality_log_probability(const panda_qual *q) {
	int v = q->qual < 0 ? 0 : (q->qual > PAMD_NQ - 1 ? PAMD_NQ - 1 : q->qual);
	return host_tables().score[v];
}

char panda_result_phred(const panda_result *r) {
	// same walk as result_phred() in pamd_kernels.hip, on the host copy of qual_score
	const double *qs = host_tables().score;
	const double p = r->p;
	if (p <= qs[0])
		return 1;
	int lo = 0, hi = PAMD_NQ - 1;
	while (lo < hi) {
		const int mid = lo + (hi - lo) / 2;
		const double s = qs[mid];
		if (s == p)
			return (char) mid;
		if (mid == lo)
			break;
		if (s > p)
			hi = mid;
		else if (s < p)
			lo = mid + 1;
		else
			break;	// NaN: the reference would spin forever; stop at the current lower bound
	}
	return (char) lo;
}

// ---- the rest of what callers and plugins reach for around the assembler (standalone build only: in a build that links the
// reference's objects, nt.c, offset.c and misc.c bring their own) ----
double panda_quality_probability(const panda_qual *q) {	// nt.c:116-119
	return exp(panda_quality_log_probability(q));
}

double panda_log1mexp(double p) {	// offset.c:35-38, with the host's libm as the reference uses it
	return (p > M_LN2) ? log1p(-exp(-p)) : log(-expm1(-p));
}

// offset.c:103-112 / :124-133 for a caller's own rows (plugins such as the reference's other_primer and hang filters call
// these; the assembler's scans run in the kernels): the same circular-bucket walk the host path uses, `reverse` reading the
// sequence from its end
size_t panda_compute_offset_qual(double threshold, double penalty, bool reverse, const panda_qual *haystack, size_t haystack_length, const panda_nt *needle, size_t needle_length) {
	const PamdTables &T = host_tables();
	return host_primer_offset(threshold, penalty, haystack_length, [&](size_t i, panda_nt &nt, double &p, double &notp) {
		const panda_qual &b = haystack[reverse ? haystack_length - i - 1 : i];
		const int q = host_clampq(b.qual);
		nt = b.nt;
		p = T.score[q];
		notp = T.score_err[q];
	}, needle, needle_length);
}

size_t panda_compute_offset_result(double threshold, double penalty, bool reverse, const panda_result *haystack, size_t haystack_length, const panda_nt *needle, size_t needle_length) {
	return host_primer_offset(threshold, penalty, haystack_length, [&](size_t i, panda_nt &nt, double &p, double &notp) {
		const panda_result &b = haystack[reverse ? haystack_length - i - 1 : i];
		nt = b.nt;
		p = b.p;
		notp = panda_log1mexp(b.p);
	}, needle, needle_length);
}

// nt.c:46-114,152-173: IUPAC letters <-> base sets, indexed as the reference indexes them (c & 0x1F: either case; U reads as T,
// X as N, any other character as the empty set).  The device parser carries the same table (pamd_io.hip, nt_table)
static const unsigned char iupac_sets[32] = { 0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0, 0, 0, 5, 6, 8, 8, 7, 9, 15, 10, 0, 0, 0, 0, 0, 0 };

panda_nt panda_nt_from_ascii(char c) {
	return (panda_nt) iupac_sets[(int) c & 0x1F];
}

panda_nt panda_nt_from_ascii_complement(char c) {	// iupac_reverse, nt.c:81-114: every base of the set complemented = the four bits in reverse order
	const unsigned v = iupac_sets[(int) c & 0x1F];
	return (panda_nt) (((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3));
}

panda_nt panda_nt_complement(panda_nt nt) {	// complementary_nt, nt.c:27-44, entry for entry: V and H both come back as V there
	static const unsigned char table[16] = { 0, 8, 4, 12, 2, 10, 6, 7, 1, 9, 5, 7, 3, 11, 7, 15 };
	return (panda_nt) table[(unsigned) nt & 15u];
}

char panda_nt_to_ascii(panda_nt val) {
	return (unsigned) val > 15u ? 'N' : "NACMGRSVTWYHKDBN"[(unsigned) val];
}

// proxy.c:141-162: what plugins say while they are set up.  The log proxy itself (proxy.c, writer.c) is not part of this
// library -- a PandaLogProxy is passed through to callbacks untouched --, so in the standalone build these go to stderr
// whatever the proxy; a program that links the reference's proxy.c gets the reference's
void panda_log_proxy_write_str(PandaLogProxy proxy, const char *str) {
	(void) proxy;
	fprintf(stderr, "%s\n", str);
}

void panda_log_proxy_write_f(PandaLogProxy proxy, const char *format, ...) {
	(void) proxy;
	va_list va;
	va_start(va, format);
	vfprintf(stderr, format, va);
	va_end(va);
	fputc('\n', stderr);
}

void panda_log_proxy_perror(PandaLogProxy proxy, const char *prefix) {
	(void) proxy;
	perror(prefix);
}

// seqid.c:46-127: what a client does with the identifier of a result -- compare, copy, print.  (Parsing a header line is the
// device parser's job, pamd_io.hip; the per-string panda_seqid_parse stays the reference's.)
void panda_seqid_clear(panda_seq_identifier *id) {
	id->instrument[0] = id->run[0] = id->flowcell[0] = id->tag[0] = '\0';
	id->lane = id->tile = id->x = id->y = 0;
}

int panda_seqid_compare(const panda_seq_identifier *one, const panda_seq_identifier *two) {	// fields in the order they print
	int r = strncmp(one->instrument, two->instrument, sizeof one->instrument);
	if (r == 0)
		r = strncmp(one->run, two->run, sizeof one->run);
	if (r == 0)
		r = strncmp(one->flowcell, two->flowcell, sizeof one->flowcell);
	if (r == 0)
		r = one->lane - two->lane;
	if (r == 0)
		r = one->tile - two->tile;
	if (r == 0)
		r = one->x - two->x;
	if (r == 0)
		r = one->y - two->y;
	if (r == 0)
		r = strncmp(one->tag, two->tag, sizeof one->tag);
	return r;
}

bool panda_seqid_equal(const panda_seq_identifier *one, const panda_seq_identifier *two) {
	return one->lane == two->lane && one->tile == two->tile && one->x == two->x && one->y == two->y && strncmp(one->instrument, two->instrument, sizeof one->instrument) == 0
		&& strncmp(one->run, two->run, sizeof one->run) == 0 && strncmp(one->flowcell, two->flowcell, sizeof one->flowcell) == 0 && strncmp(one->tag, two->tag, sizeof one->tag) == 0;
}

void panda_seqid_copy(const panda_seq_identifier *src, panda_seq_identifier *dest) {
	dest->lane = src->lane;
	dest->tile = src->tile;
	dest->x = src->x;
	dest->y = src->y;
	strncpy(dest->instrument, src->instrument, sizeof dest->instrument);
	strncpy(dest->run, src->run, sizeof dest->run);
	strncpy(dest->flowcell, src->flowcell, sizeof dest->flowcell);
	strncpy(dest->tag, src->tag, sizeof dest->tag);
}

void panda_seqid_xprint(const panda_seq_identifier *id, PandaPrintf xprintf, void *x) {	// seqid.c:109-116
	if (id != nullptr)
		xprintf(x, "%s:%s:%s:%d:%d:%d:%d:%s", id->instrument, id->run, id->flowcell, id->lane, id->tile, id->x, id->y, id->tag);
}

void panda_seqid_print(const panda_seq_identifier *id, FILE *file) {
	if (id != nullptr)
		fprintf(file, "%s:%s:%s:%d:%d:%d:%d:%s", id->instrument, id->run, id->flowcell, id->lane, id->tile, id->x, id->y, id->tag);
}

const char *panda_seqid_str(const panda_seq_identifier *id) {	// a buffer per thread, as the reference's (buffer.c)
	static thread_local char text[sizeof(panda_seq_identifier) + 64];
	if (id == nullptr)
		return nullptr;
	snprintf(text, sizeof text, "%s:%s:%s:%d:%d:%d:%d:%s", id->instrument, id->run, id->flowcell, id->lane, id->tile, id->x, id->y, id->tag);
	return text;
}

// ---- seqid.c:127-285: one header string -> the identifier (the device parser's program, pamd_io.hip parse_header, for a
// caller's single strings).  A field is a run of characters up to '\0' ':' '#' '/' ' ' (PARSE_CHUNK_MAYBE, seqid.c:136).
extern "C++" {
namespace {
struct IdCursor {
	const char **at;
	bool at_end() const { return **at == '\0'; }
	static bool ends(char c) { return c == '\0' || c == ':' || c == '#' || c == '/' || c == ' '; }
	bool push() {	// PARSE_PUSH
		if (at_end())
			return false;
		(*at)++;
		return true;
	}
	bool integer(int &value) {	// PARSE_INT: digits only, the value wraps as a 32-bit integer
		unsigned v = 0;
		if (at_end())
			return false;
		for (; !ends(**at); (*at)++) {
			if (**at < '0' || **at > '9')
				return false;
			v = 10u * v + (unsigned) (**at - '0');
		}
		value = (int) v;
		return true;
	}
	bool text(char *target, size_t room) {	// PARSE_STR: a field that must not start at the terminator, shorter than its array
		size_t n = 0;
		if (at_end())
			return false;
		for (; !ends(**at); (*at)++) {
			if (n + 1 >= room)
				return false;
			target[n++] = **at;
		}
		target[n] = '\0';
		return true;
	}
	bool tag(panda_seq_identifier *id) {	// the tag chunk: may be empty, at most PANDA_TAG_LEN characters (seqid.c:221-226,269-274)
		size_t n = 0;
		id->tag[0] = '\0';
		for (; !ends(**at); (*at)++) {
			if (n >= PANDA_TAG_LEN)
				return false;
			id->tag[n++] = **at;
			// (a tag of exactly PANDA_TAG_LEN characters ends in the structure's padding byte, as the reference's does: seqid.c:224)
			static_assert(offsetof(panda_seq_identifier, tag) + PANDA_TAG_LEN < sizeof(panda_seq_identifier), "room for the terminator");
			reinterpret_cast<char *>(id)[offsetof(panda_seq_identifier, tag) + n] = '\0';
		}
		return true;
	}
};
}	// namespace
}	// extern "C++"

int panda_seqid_parse_fail(panda_seq_identifier *id, const char *input, PandaTagging policy, PandaIdFmt *detected_format, const char **end_ptr) {
	const char *temp;
	int value = 0;
	if (end_ptr == nullptr)
		end_ptr = &temp;
	*end_ptr = input;
	IdCursor c = { end_ptr };
	auto policy_ok = [&]() { return policy == PANDA_TAG_OPTIONAL || policy == (id->tag[0] == '\0' ? PANDA_TAG_ABSENT : PANDA_TAG_PRESENT); };
	auto four_ints = [&]() {	// lane, tile, x, y, each followed by its separator
		int *field[4] = { &id->lane, &id->tile, &id->x, &id->y };
		for (int *f : field) {
			if (!c.integer(value) || !c.push())
				return false;
			*f = value;
		}
		return true;
	};
	if (strlen(input) > 3 && (input[0] == 'E' || input[0] == 'S') && input[1] == 'R' && input[2] == 'R') {	// Short Read Archive, seqid.c:157-172
		if (detected_format != nullptr)
			*detected_format = input[0] == 'S' ? PANDA_IDFMT_SRA : PANDA_IDFMT_EBI_SRA;
		*end_ptr += 3;
		panda_seqid_clear(id);
		for (int field = 0; field < 2; field++) {	// PARSE_SRA_INT + PARSE_PUSH: <number>.<number>
			unsigned v = 0;
			for (; **end_ptr != '\0' && **end_ptr != '.' && **end_ptr != ' '; (*end_ptr)++) {
				if (**end_ptr < '0' || **end_ptr > '9')
					return 0;
				v = 10u * v + (unsigned) (**end_ptr - '0');
			}
			if (!c.push())
				return 0;
			if (field == 0)
				snprintf(id->instrument, sizeof id->instrument, "%cRR%d", (int) input[0], (int) v);
			else
				id->lane = (int) v;
		}
		return c.push() ? 1 : 0;
	}
	if (strchr(input, '/') != nullptr) {	// CASAVA 1.4-1.6, or 1.7 converted to it (six colons before any '#'), seqid.c:173-235
		size_t colons = 0;
		for (const char *p = input; *p != '\0' && *p != '#'; p++)
			colons += *p == ':';
		if (detected_format != nullptr)
			*detected_format = colons == 6 ? PANDA_IDFMT_CASAVA_CONVERTED : PANDA_IDFMT_CASAVA_1_4;
		if (colons != 6)
			id->run[0] = id->flowcell[0] = '\0';
		if (!c.text(id->instrument, sizeof id->instrument) || !c.push())
			return 0;
		if (colons == 6 && (!c.text(id->run, sizeof id->run) || !c.push() || !c.text(id->flowcell, sizeof id->flowcell) || !c.push()))
			return 0;
		if (!four_ints())
			return 0;
		id->tag[0] = '\0';
		if (*(*end_ptr - 1) == '#' && (!c.tag(id) || !c.push()))
			return 0;
		if (!policy_ok())
			return 0;
		return c.integer(value) ? value : 0;
	}
	// CASAVA 1.7+, seqid.c:236-284: instrument:run:flowcell:lane:tile:x:y mate:filtered:control:tag
	if (detected_format != nullptr)
		*detected_format = PANDA_IDFMT_CASAVA_1_7;
	if (!c.text(id->instrument, sizeof id->instrument) || !c.push() || !c.text(id->run, sizeof id->run) || !c.push() || !c.text(id->flowcell, sizeof id->flowcell) || !c.push())
		return 0;
	if (!four_ints())
		return 0;
	int mate = 0;
	if (!c.integer(mate) || !c.push())
		return 0;
	if (c.at_end())	// "filtered": any chunk
		return 0;
	while (!IdCursor::ends(**end_ptr))
		(*end_ptr)++;
	if (!c.push() || !c.integer(value) || !c.push())	// control bits
		return 0;
	if (!c.tag(id) || !policy_ok())
		return 0;
	return mate;
}

int panda_seqid_parse(panda_seq_identifier *id, const char *input, PandaTagging policy) {
	return panda_seqid_parse_fail(id, input, policy, nullptr, nullptr);
}

// ---- writer.c: the writer object.  What a thread appends is its own until it commits; a commit hands it to `write` whole ----
struct panda_writer {
	size_t refcnt = 1;
	PandaBufferWrite write = nullptr;
	void *write_data = nullptr;
	PandaDestroy write_destroy = nullptr;
	PandaWriter commit_slave = nullptr;
	std::mutex lock;
	unsigned long serial = 0;	// (a thread finds its pending text under this, not under the address: a new writer at an old one's address starts empty)
	// what each thread has appended and not committed (writer.c:40-52's buffer_list: the writer owns the buffers, a thread finds
	// its own through a thread-local key); the last unref writes what is left in them (writer.c:186-193)
	std::list<std::pair<std::thread::id, std::string>> buffers;
};
extern "C++" {
namespace {
std::atomic<unsigned long> g_writer_serial{1};
std::string &pending_of(PandaWriter w) {
	static thread_local std::vector<std::pair<unsigned long, std::string *>> mine;	// this thread's buffers, by writer (a cache: the text is the writer's)
	for (auto &e : mine)
		if (e.first == w->serial)
			return *e.second;
	std::string *text = nullptr;
	{
		std::lock_guard<std::mutex> g(w->lock);
		for (auto &b : w->buffers)
			if (b.first == std::this_thread::get_id())
				text = &b.second;
		if (text == nullptr) {
			w->buffers.emplace_back(std::this_thread::get_id(), std::string());
			text = &w->buffers.back().second;
		}
	}
	if (mine.size() >= 64)	// (writers come and go: the oldest cache entry makes room; its text stays with its writer)
		mine.erase(mine.begin());
	mine.emplace_back(w->serial, text);
	return *text;
}
void file_write(const char *buffer, size_t buffer_length, void *file) {
	if (fwrite(buffer, 1, buffer_length, static_cast<FILE *>(file)) != buffer_length)
		perror("writer");
}
void bzip_write(const char *buffer, size_t buffer_length, void *data) {	// writer.c:108-114
	int bzerror;
	pamd_bz2()->bzWrite(&bzerror, data, const_cast<char *>(buffer), (int) buffer_length);
}
void bzip_close(void *data) {
	pamd_bz2()->bzclose(data);
}

void null_write(const char *, size_t, void *) {
}
void close_file(void *file) {
	fclose(static_cast<FILE *>(file));
}
}	// namespace
}	// extern "C++"

PandaWriter panda_writer_new(PandaBufferWrite write, void *write_data, PandaDestroy write_destroy) {
	PandaWriter w = new panda_writer;
	w->write = write;
	w->write_data = write_data;
	w->write_destroy = write_destroy;
	w->serial = g_writer_serial.fetch_add(1);
	return w;
}

PandaWriter panda_writer_new_file(FILE *file) {
	return panda_writer_new(file_write, file, close_file);
}

PandaWriter panda_writer_new_null(void) {
	return panda_writer_new(null_write, nullptr, nullptr);
}

PandaWriter panda_writer_new_stderr(void) {
	return panda_writer_new(file_write, stderr, nullptr);
}

PandaWriter panda_writer_new_stdout(void) {
	return panda_writer_new(file_write, stdout, nullptr);
}

PandaWriter panda_writer_open_file(const char *filename, bool bzip) {
	if (bzip) {	// writer.c:108-151: BZ2_bzopen / BZ2_bzWrite / BZ2_bzclose (libbz2 bound at run time, pamd_bz2.h)
		const PamdBz2 *bz = pamd_bz2();
		if (bz == nullptr) {
			pamd_set_error("%s: bzip2 output needs libbz2.so.1, which cannot be loaded here", filename);
			return nullptr;
		}
		void *file = bz->bzopen(filename, "w");
		return file == nullptr ? nullptr : panda_writer_new(bzip_write, file, bzip_close);
	}
	FILE *file = fopen(filename, "w");
	return file == nullptr ? nullptr : panda_writer_new_file(file);
}

void panda_writer_append_v(PandaWriter writer, const char *format, va_list va) {
	char text[2048];	// (the reference's uncommitted buffer holds 2048 bytes: longer output is cut there too)
	const int n = vsnprintf(text, sizeof text, format, va);
	if (n > 0)
		pending_of(writer).append(text, (size_t) n < sizeof text ? (size_t) n : sizeof text - 1);
}

void panda_writer_append(PandaWriter writer, const char *format, ...) {
	va_list va;
	va_start(va, format);
	panda_writer_append_v(writer, format, va);
	va_end(va);
}

void panda_writer_append_c(PandaWriter writer, char c) {
	pending_of(writer).push_back(c);
}

void panda_writer_append_id(PandaWriter writer, const panda_seq_identifier *id) {
	panda_seqid_xprint(id, reinterpret_cast<PandaPrintf>(panda_writer_append), writer);
}

void panda_writer_commit(PandaWriter writer) {
	std::string &text = pending_of(writer);
	if (!text.empty()) {
		std::lock_guard<std::mutex> g(writer->lock);
		writer->write(text.data(), text.size(), writer->write_data);
	}
	text.clear();
	if (writer->commit_slave != nullptr)
		panda_writer_commit(writer->commit_slave);
}

void panda_writer_flush(PandaWriter writer) {	// (nothing is held back past a commit here)
	(void) writer;
}

PandaWriter panda_writer_get_slave(PandaWriter writer) {
	return writer->commit_slave;
}

void panda_writer_set_slave(PandaWriter writer, PandaWriter slave) {	// writer.c:288-306: no cycles
	for (PandaWriter check = slave; check != nullptr; check = check->commit_slave)
		if (check == writer || check->commit_slave == writer)
			return;
	if (writer->commit_slave != nullptr)
		panda_writer_unref(writer->commit_slave);
	writer->commit_slave = panda_writer_ref(slave);
}

PandaWriter panda_writer_ref(PandaWriter writer) {
	if (writer == nullptr)
		return nullptr;
	std::lock_guard<std::mutex> g(writer->lock);
	writer->refcnt++;
	return writer;
}

void panda_writer_unref(PandaWriter writer) {
	if (writer == nullptr)
		return;
	size_t left;
	{
		std::lock_guard<std::mutex> g(writer->lock);
		left = --writer->refcnt;
	}
	if (left == 0) {
		for (auto &b : writer->buffers)	// writer.c:186-193: what any thread appended and never committed is written now
			if (!b.second.empty())
				writer->write(b.second.data(), b.second.size(), writer->write_data);
		if (writer->commit_slave != nullptr)
			panda_writer_unref(writer->commit_slave);
		if (writer->write_destroy != nullptr && writer->write_data != nullptr)
			writer->write_destroy(writer->write_data);
		delete writer;
	}
}

// ---- output.c:85-126: one record as FASTA / FASTQ text ----
bool panda_output_fasta(const panda_result_seq *sequence, PandaWriter writer) {
	if (sequence->sequence_length == 0)
		return true;
	panda_writer_append_c(writer, '>');
	panda_writer_append_id(writer, &sequence->name);
	panda_writer_append(writer, ";%f", exp(sequence->quality));
	panda_writer_append_c(writer, '\n');
	std::string &text = pending_of(writer);
	for (size_t it = 0; it < sequence->sequence_length; it++)
		text.push_back(panda_nt_to_ascii(sequence->sequence[it].nt));
	text.push_back('\n');
	panda_writer_commit(writer);
	return true;
}

bool panda_output_fastq(const panda_result_seq *sequence, PandaWriter writer) {
	if (sequence->sequence_length == 0)
		return true;
	panda_writer_append_c(writer, '@');
	panda_writer_append_id(writer, &sequence->name);
	panda_writer_append(writer, ";%f", exp(sequence->quality));
	panda_writer_append_c(writer, '\n');
	std::string &text = pending_of(writer);
	for (size_t it = 0; it < sequence->sequence_length; it++)
		text.push_back(panda_nt_to_ascii(sequence->sequence[it].nt));
	text.append("\n+\n");
	for (size_t it = 0; it < sequence->sequence_length; it++)
		text.push_back((char) (33 + panda_result_phred(&sequence->sequence[it])));
	text.push_back('\n');
	panda_writer_commit(writer);
	return true;
}

int panda_api_version(void) {	// misc.c:33-36
	return PANDA_API;
}

char const *panda_version(void) {	// misc.c:28-31 (PACKAGE_STRING there)
	return "pandaseq_amd (assembler core of pandaseq 2.11 on MI355X)";
}
#endif

void pamd_compat_set_device(int device) {
	g_device.store(device);
}

void pamd_compat_set_thread_device(int device) {
	t_device = device;
}

int pamd_compat_device_for_new_assembler(void) {
	return pick_device();
}

void pamd_compat_set_batch_pairs(size_t pairs) {
	g_batch_pairs.store(pairs ? pairs : 1);
}

void pamd_compat_set_lazy_rows(int lazy) {
	g_lazy_rows.store(lazy ? 1 : 0);
}

void pamd_compat_set_per_pair_device(int device) {
	g_per_pair_device.store(device ? 1 : 0);
}

void pamd_compat_set_exact_sums(int exact) {
	g_exact_sums.store(exact ? 1 : 0);
	g_fast_sums_asked.store(!exact);
}

#ifndef PAMD_WITH_REFERENCE_OBJECTS
size_t panda_max_len(void) {	// misc.c:38-41
	return PAMD_MAX_LEN;
}
#endif

PandaAssembler panda_assembler_new(PandaNextSeq next, void *next_data, PandaDestroy next_destroy, PandaLogProxy logger) {
	return panda_assembler_new_kmer(next, next_data, next_destroy, logger, PANDA_DEFAULT_NUM_KMERS);
}

PandaAssembler panda_assembler_new_kmer(PandaNextSeq next, void *next_data, PandaDestroy next_destroy, PandaLogProxy logger, size_t num_kmers) {
	pamd_assembler *dev = nullptr;
	const int device = pick_device();
	// num_kmers != 2 (assembler_support.c:80): the reference's table indexing is consistent only for 2 (assembler.c:94 vs :99); what it
	// does for other values is reproduced pair by pair on the host path (host_align; host_routed() below), the kernels seed with 2
	if (num_kmers > PAMD_MAX_NUM_KMERS) {
		pamd_set_error("num_kmers = %zu: at most %d (the reference asks for \"no more than 10\", pandaseq-assembler.h:58)", num_kmers, PAMD_MAX_NUM_KMERS);
	} else {
		dev = pamd_assembler_create(device);
	}
	if (dev == nullptr) {
		if (next_destroy != nullptr)
			next_destroy(next_data);
		fprintf(stderr, "pandaseq_amd: %s\n", pamd_last_error());
		return nullptr;
	}
	PandaAssembler a = new panda_assembler;
	a->next = next;
	a->next_data = next_data;
	a->next_destroy = next_destroy;
	a->logger = logger;
	a->dev = dev;
	a->device = device;
	a->devs[0] = dev;
	for (int k = 1; k < panda_assembler::NSLOT && next != nullptr; k++)	// only a source-driven assembler keeps several batches in flight
		a->devs[k] = pamd_assembler_create(device);
	a->num_kmers = num_kmers;
	a->algo = panda_algorithm_simple_bayes_new();	// assembler_support.c:77
	memset(&a->result, 0, sizeof a->result);
	a->result.sequence = a->result_seq;
	return a;
}

// Test hook (tests/c/test_per_pair.c; not declared in the public headers): an assembler WITHOUT a device, on which only the
// per-pair host path works -- panda_assembler_assemble(), assembler.c:368-383 -- so that path can be checked against the oracle
// on a machine without a GPU.  Every batch entry point fails on it, loudly (push_config).  panda_assembler_new() itself still
// refuses to build an assembler without a GPU.
PandaAssembler pamd_compat_host_only_assembler_new(void) {
	PandaAssembler a = new panda_assembler;
	a->algo = panda_algorithm_simple_bayes_new();
	memset(&a->result, 0, sizeof a->result);
	a->result.sequence = a->result_seq;
	return a;
}
// ... and one with a given num_kmers (panda_assembler_new_kmer, assembler_support.c:36-100)
PandaAssembler pamd_compat_host_only_assembler_new_kmer(size_t num_kmers) {
	if (num_kmers > PAMD_MAX_NUM_KMERS)
		return nullptr;
	PandaAssembler a = pamd_compat_host_only_assembler_new();
	a->num_kmers = num_kmers;
	return a;
}

PandaAssembler panda_assembler_ref(PandaAssembler a) {
	std::lock_guard<std::mutex> g(g_ref_lock);
	a->refcnt++;
	return a;
}

void panda_assembler_unref(PandaAssembler a) {	// assembler_support.c:151-175
	if (a == nullptr)
		return;
	size_t c;
	{
		std::lock_guard<std::mutex> g(g_ref_lock);
		c = --a->refcnt;
	}
	if (c != 0)
		return;
	for (PandaModule m : a->modules)
		panda_module_unref(m);
	if (a->next_destroy != nullptr && a->next != nullptr)
		a->next_destroy(a->next_data);
	if (a->noalgn_destroy != nullptr && a->noalgn != nullptr)
		a->noalgn_destroy(a->noalgn_data);
	if (getenv("PAMD_COMPAT_STATS") != nullptr && a->n_batches > 0)
		fprintf(stderr, "pandaseq_amd: %ld batches; caller: %.3f s pulling and staging, %.3f s blocked on a batch; background: %.3f s in all (%.2f ms per batch)\n", a->n_batches, a->t_fill, a->t_wait, a->t_bg, 1e3 * a->t_bg / (double) a->n_batches);
	panda_algorithm_unref(a->algo);
	if (a->slots[0] != nullptr)
		g_streaming.fetch_sub(1);
	for (pamd_slot *sl : a->slots)
		slot_free(sl);
	for (pamd_assembler *d : a->devs)
		pamd_assembler_destroy(d);
	delete a;
}

void panda_assembler_copy_configuration(PandaAssembler dest, PandaAssembler src) {	// assembler_support.c:119-137
	for (PandaModule m : src->modules)
		panda_assembler_add_module(dest, m);
	panda_assembler_set_forward_primer(dest, src->forward_primer, src->forward_primer_length);
	panda_assembler_set_reverse_primer(dest, src->reverse_primer, src->reverse_primer_length);
	dest->forward_trim = src->forward_trim;
	dest->reverse_trim = src->reverse_trim;
	dest->threshold = src->threshold;
	dest->minoverlap = src->minoverlap;
	dest->maxoverlap = src->maxoverlap;
	dest->post_primers = src->post_primers;
	panda_algorithm_unref(dest->algo);
	dest->algo = panda_algorithm_ref(src->algo);
	dest->primer_penalty = src->primer_penalty;
	dest->dev_dirty = true;
}

const panda_result_seq *panda_assembler_assemble(PandaAssembler a, panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length) {
	if (forward_length > PAMD_MAX_LEN || reverse_length > PAMD_MAX_LEN) {	// the reference asserts (assembler.c:375-376)
		fprintf(stderr, "pandaseq_amd: read longer than PANDA_MAX_LEN\n");
		abort();
	}
	// a pending prefetched batch of panda_assembler_next() shares the staging area: refuse to mix
	bool pending = false;
	for (pamd_slot *sl : a->slots)
		pending |= sl != nullptr && (sl->pos < sl->n || sl->done.valid());
	if (pending) {
		fprintf(stderr, "pandaseq_amd: panda_assembler_assemble() called while panda_assembler_next() has prefetched pairs pending\n");
		return nullptr;
	}
	// One pair handed over at a time (what panda_diff does, diff.c:26-99) is assembled on the caller's thread through the
	// class's two functions, as assembler.c:368-383 does: an upload, two launches and a download per pair cost 40 us, the pair
	// itself 3.  Arrays (pamd_compat_assemble_many) and streams (panda_assembler_next) of pairs go to the kernels.
	// PAMD_COMPAT_PER_PAIR=device / pamd_compat_set_per_pair_device(1) sends single pairs there too (tests compare the two).
	if (host_routed(a) || !compat_per_pair_device() || a->dev == nullptr)
		return host_assemble(a, id, forward, forward_length, reverse, reverse_length);
	ensure_staging(a, 1);
	stage_pair(a, 0, id, forward, forward_length, reverse, reverse_length);
	a->batch_n = 1;
	a->batch_pos = 1;
	if (run_batch(a) != PAMD_OK) {
		fprintf(stderr, "pandaseq_amd: %s\n", pamd_last_error());
		abort();	// no CPU fallback: a device failure is fatal, never silently wrong
	}
	return replay(a, 0, forward, reverse);
}

namespace {
// the caller's arrays as a PandaNextSeq source (pamd_compat_assemble_many)
struct array_source {
	const panda_seq_identifier *ids;
	const panda_qual *const *forward, *const *reverse;
	const size_t *forward_length, *reverse_length;
	size_t n, pos;
};
static bool array_next(panda_seq_identifier *id, const panda_qual **forward, size_t *forward_length, const panda_qual **reverse, size_t *reverse_length, void *user_data) {
	array_source *src = static_cast<array_source *>(user_data);
	if (src->pos >= src->n)
		return false;
	const size_t i = src->pos++;
	*id = src->ids[i];
	*forward = src->forward[i];
	*forward_length = src->forward_length[i];
	*reverse = src->reverse[i];
	*reverse_length = src->reverse_length[i];
	return true;
}
}	// namespace

// slots 0 .. count-1 exist, each with a device assembler of its own (a launch owns its assembler's deferred list); the ones
// beyond stay unmade until a longer array or panda_assembler_next() wants them
static void ensure_slots(PandaAssembler a, int count) {
	for (int k = 0; k < count; k++) {
		if (a->devs[k] == nullptr) {
			a->devs[k] = pamd_assembler_create(a->device);
			a->dev_dirty = true;
			if (a->devs[k] == nullptr) {
				fprintf(stderr, "pandaseq_amd: %s\n", pamd_last_error());
				abort();
			}
		}
		if (a->slots[k] == nullptr) {
			if (k == 0)
				g_streaming.fetch_add(1);
			a->slots[k] = new pamd_slot;
		}
		a->slots[k]->dev = a->devs[k];
	}
}

size_t pamd_compat_assemble_many(PandaAssembler a, const panda_seq_identifier *ids, const panda_qual *const *forward, const size_t *forward_length, const panda_qual *const *reverse, const size_t *reverse_length, size_t n, PamdVisit visit, void *user_data) {
	// a pending prefetched batch of panda_assembler_next() shares the staging area: refuse to mix (as panda_assembler_assemble)
	for (pamd_slot *sl : a->slots)
		if (sl != nullptr && (sl->pos < sl->n || sl->done.valid())) {
			fprintf(stderr, "pandaseq_amd: pamd_compat_assemble_many() called while panda_assembler_next() has prefetched pairs pending\n");
			return 0;
		}
	size_t assembled = 0;
	if (host_routed(a)) {	// a class of the caller's, or num_kmers != 2: pair by pair on this thread
		for (size_t i = 0; i < n; i++) {
			if (forward_length[i] > PAMD_MAX_LEN || reverse_length[i] > PAMD_MAX_LEN) {	// the reference asserts (assembler.c:375-376)
				fprintf(stderr, "pandaseq_amd: read longer than PANDA_MAX_LEN\n");
				abort();
			}
			const panda_result_seq *r = host_assemble(a, &ids[i], forward[i], forward_length[i], reverse[i], reverse_length[i]);
			assembled += r != nullptr;
			if (visit != nullptr)
				visit(i, r, user_data);
		}
		return assembled;
	}
	if (a->dev == nullptr) {
		fprintf(stderr, "pandaseq_amd: this assembler has no device: only panda_assembler_assemble() runs on it\n");
		abort();
	}
	// The arrays are a source like any other: they go through panda_assembler_next()'s machinery -- 65 536 pairs per batch,
	// four batches in flight (upload, kernels, index-coded download and expansion of different batches overlap), records handed
	// back in order -- with the one difference that `visit` sees every pair, the ones that did not assemble as NULL.
	// (Round 3 ran one synchronous upload / launch / download of full strided rows per batch: 1.0 M pairs/s.)
	// an array no longer than a batch is cut in NSLOT pieces, so that its upload, kernels, download and expansion overlap as well;
	// a handful of pairs is one piece in one slot (round 4 made four slots and three more device assemblers for any n)
	size_t batch = g_batch_pairs.load();
	const size_t piece = (n + panda_assembler::NSLOT - 1) / panda_assembler::NSLOT;
	if (piece < batch)
		batch = piece < 4096 ? 4096 : piece;
	const int ns = (int) std::min<size_t>(panda_assembler::NSLOT, std::max<size_t>(1, (n + batch - 1) / batch));
	ensure_slots(a, ns);
	array_source src = { ids, forward, reverse, forward_length, reverse_length, n, 0 };
	const PandaNextSeq saved_next = a->next;
	void *const saved_data = a->next_data;
	const bool saved_eof = a->eof, saved_primed = a->primed;
	const int saved_cur = a->cur;
	a->next = array_next;
	a->next_data = &src;
	a->eof = false;
	for (int k = 0; k < ns; k++)
		slot_fill_and_submit(a, a->slots[k], batch);
	size_t index = 0;
	for (int cur = 0;; cur = (cur + 1) % ns) {
		pamd_slot *s = a->slots[cur];
		slot_wait(s);
		if (s->n == 0)
			break;	// the arrays are exhausted (slots are filled in order: an empty one is the end)
		a->n_batches++;
		for (size_t i = 0; i < s->n; i++) {
			const panda_result_seq *r = slot_replay(a, s, i);
			assembled += r != nullptr;
			if (visit != nullptr)
				visit(index, r, user_data);
			index++;
		}
		s->pos = s->n;
		slot_fill_and_submit(a, s, batch);
	}
	for (int k = 0; k < ns; k++) {	// nothing may be left in flight or half consumed
		slot_wait(a->slots[k]);
		a->slots[k]->n = a->slots[k]->pos = 0;
	}
	a->next = saved_next;
	a->next_data = saved_data;
	a->eof = saved_eof;
	a->primed = saved_primed;
	a->cur = saved_cur;
	return assembled;
}

const panda_result_seq *panda_assembler_next(PandaAssembler a) {
	if (a->next == nullptr)
		return nullptr;
	if (host_routed(a) && !a->primed) {	// a class of the caller's, or num_kmers != 2: pair by pair on this thread (assembler.c:350-371)
		for (;;) {
			const panda_qual *f = nullptr, *r = nullptr;
			size_t fl = 0, rl = 0;
			if (!a->next(&a->result.name, &f, &fl, &r, &rl, a->next_data))
				return nullptr;
			if (fl > PAMD_MAX_LEN || rl > PAMD_MAX_LEN) {	// the reference asserts (assembler.c:359-360)
				fprintf(stderr, "pandaseq_amd: read longer than PANDA_MAX_LEN\n");
				abort();
			}
			const panda_result_seq *res = host_assemble(a, &a->result.name, f, fl, r, rl);
			if (res != nullptr)
				return res;
		}
	}
	if (a->slots[panda_assembler::NSLOT - 1] == nullptr)
		ensure_slots(a, panda_assembler::NSLOT);
	while (true) {
		pamd_slot *s = a->slots[a->cur];
		if (s->pos == s->n) {
			if (!a->primed) {	// the first call: every slot gets a batch, the first one is waited for
				a->primed = true;
				const double t0 = now_s();
				for (int k = 0; k < panda_assembler::NSLOT; k++)
					slot_fill_and_submit(a, a->slots[k]);
				a->t_fill += now_s() - t0;
				a->cur = 0;
			} else {	// the batch just consumed makes room for a new one at the end of the line
				if (s->n == 0 && a->eof)
					return nullptr;
				const double t0 = now_s();
				slot_fill_and_submit(a, s);
				a->t_fill += now_s() - t0;
				a->cur = (a->cur + 1) % panda_assembler::NSLOT;
			}
			s = a->slots[a->cur];
			const double t1 = now_s();
			slot_wait(s);
			a->t_wait += now_s() - t1;
			a->t_bg += s->seconds;
			a->n_batches++;
			if (s->n == 0)
				return nullptr;	// the source is exhausted (assembler.c:352-353)
		}
		const panda_result_seq *res = slot_replay(a, s, s->pos++);
		if (res != nullptr)
			return res;
	}
}

bool panda_assembler_add_module(PandaAssembler a, PandaModule module) {	// module.c:156-180
	if (module == nullptr)
		return false;
	a->modules.push_back(panda_module_ref(module));
	a->rejected.push_back(0);
	a->dev_dirty = true;
	return true;
}

size_t panda_assembler_add_modules(PandaAssembler a, PandaModule *modules, size_t modules_length) {
	size_t it;
	for (it = 0; it < modules_length; it++)
		if (!panda_assembler_add_module(a, modules[it]))
			return it;
	return it;
}

bool panda_assembler_foreach_module(PandaAssembler a, PandaModuleCallback callback, void *data) {
	for (size_t it = 0; it < a->modules.size(); it++)
		if (!callback(a, a->modules[it], a->rejected[it], data))
			return false;
	return true;
}

void panda_assembler_module_stats(PandaAssembler a) {	// module.c:208-216 (STAT lines; here to stderr)
	for (size_t it = 0; it < a->modules.size(); it++)
		if (a->rejected[it] > 0)
			fprintf(stderr, "%s%sSTAT\t%s\t%zu\n", a->name, a->name[0] ? "\t" : "", a->modules[it]->name, a->rejected[it]);
}

PandaAlgorithm panda_assembler_get_algorithm(PandaAssembler a) {
	return a->algo;
}

void panda_assembler_set_algorithm(PandaAssembler a, PandaAlgorithm algorithm) {
	if (algorithm == nullptr)
		return;
	// one of this library's seven classes (scored by the kernels), or a class that brings its own two functions (scored on
	// the host through them, as the reference calls them: assembler.c:54-57,125,209); anything else has no implementation
	// here: say so now and keep the algorithm the assembler had, rather than fail at the first batch
	const struct panda_algorithm_class *cls = panda_algorithm_class(algorithm);
	if (device_algo_of(algorithm) < 0 && (cls == nullptr || cls->overlap_probability == nullptr || cls->match_probability == nullptr)) {
		fprintf(stderr, "pandaseq_amd: algorithm %s has neither a device implementation nor overlap_probability / match_probability functions; keeping %s\n", cls != nullptr && cls->name != nullptr ? cls->name : "(unnamed)", a->algo != nullptr ? panda_algorithm_class(a->algo)->name : "(none)");
		return;
	}
	panda_algorithm_unref(a->algo);
	a->algo = panda_algorithm_ref(algorithm);
	a->dev_dirty = true;
}

long panda_assembler_get_bad_read_count(PandaAssembler a) { return a->badreadcount; }
long panda_assembler_get_count(PandaAssembler a) { return a->count; }

void panda_assembler_set_fail_alignment(PandaAssembler a, PandaFailAlign handler, void *handler_data, PandaDestroy handler_destroy) {
	if (a->noalgn_destroy != nullptr && a->noalgn != nullptr)
		a->noalgn_destroy(a->noalgn_data);
	a->noalgn = handler;
	a->noalgn_data = handler_data;
	a->noalgn_destroy = handler_destroy;
}

long panda_assembler_get_failed_alignment_count(PandaAssembler a) { return a->noalgncount; }

panda_nt *panda_assembler_get_forward_primer(PandaAssembler a, size_t *length) {
	if (length != nullptr)
		*length = a->forward_primer_length;
	return a->forward_primer_length == 0 ? nullptr : a->forward_primer;
}

void panda_assembler_set_forward_primer(PandaAssembler a, panda_nt *sequence, size_t length) {
	if (length < PAMD_MAX_LEN) {
		for (size_t it = 0; it < length; it++)
			a->forward_primer[it] = sequence[it];
		a->forward_primer_length = length;
		a->forward_trim = 0;
		a->dev_dirty = true;
	}
}

size_t panda_assembler_get_forward_trim(PandaAssembler a) { return a->forward_trim; }

void panda_assembler_set_forward_trim(PandaAssembler a, size_t trim) {
	a->forward_trim = trim;
	a->forward_primer_length = 0;
	a->dev_dirty = true;
}

size_t panda_assembler_get_longest_overlap(PandaAssembler a) { return a->longest_overlap; }
long panda_assembler_get_low_quality_count(PandaAssembler a) { return a->lowqcount; }
int panda_assembler_get_minimum_overlap(PandaAssembler a) { return a->minoverlap; }

void panda_assembler_set_minimum_overlap(PandaAssembler a, int overlap) {
	if (overlap > 1 && (size_t) overlap < 2 * PAMD_MAX_LEN) {
		a->minoverlap = overlap;
		a->dev_dirty = true;
	}
}

int panda_assembler_get_maximum_overlap(PandaAssembler a) { return a->maxoverlap; }

void panda_assembler_set_maximum_overlap(PandaAssembler a, int overlap) {
	if (overlap >= 0 && (size_t) overlap < 2 * PAMD_MAX_LEN) {
		a->maxoverlap = overlap;
		a->dev_dirty = true;
	}
}

const char *panda_assembler_get_name(PandaAssembler a) {
	if (a == nullptr || a->name[0] == '\0')
		return nullptr;
	return a->name;
}

void panda_assembler_set_name(PandaAssembler a, const char *name) {
	if (name == nullptr) {
		a->name[0] = '\0';
		return;
	}
	strncpy(a->name, name, PAMD_MAX_LEN);
	a->name[PAMD_MAX_LEN - 1] = '\0';
}

long panda_assembler_get_no_forward_primer_count(PandaAssembler a) { return a->nofpcount; }
long panda_assembler_get_no_reverse_primer_count(PandaAssembler a) { return a->norpcount; }
size_t panda_assembler_get_num_kmer(PandaAssembler a) { return a->num_kmers; }
long panda_assembler_get_ok_count(PandaAssembler a) { return a->okcount; }

long panda_assembler_get_overlap_count(PandaAssembler a, size_t overlap) {
	return (overlap < 2 * PAMD_MAX_LEN) ? a->overlapcount[overlap] : -1;
}

bool panda_assembler_get_primers_after(PandaAssembler a) { return a->post_primers; }

void panda_assembler_set_primers_after(PandaAssembler a, bool after) {
	a->post_primers = after;
	a->dev_dirty = true;
}

panda_nt *panda_assembler_get_reverse_primer(PandaAssembler a, size_t *length) {
	if (length != nullptr)
		*length = a->reverse_primer_length;
	return a->reverse_primer_length == 0 ? nullptr : a->reverse_primer;
}

void panda_assembler_set_reverse_primer(PandaAssembler a, panda_nt *sequence, size_t length) {
	if (length < PAMD_MAX_LEN) {
		for (size_t it = 0; it < length; it++)
			a->reverse_primer[it] = sequence[it];
		a->reverse_primer_length = length;
		a->reverse_trim = 0;
		a->dev_dirty = true;
	}
}

size_t panda_assembler_get_reverse_trim(PandaAssembler a) { return a->reverse_trim; }

void panda_assembler_set_reverse_trim(PandaAssembler a, size_t trim) {
	a->reverse_trim = trim;
	a->reverse_primer_length = 0;
	a->dev_dirty = true;
}

long panda_assembler_get_slow_count(PandaAssembler a) { return a->slowcount; }
double panda_assembler_get_threshold(PandaAssembler a) { return std::exp(a->threshold); }

void panda_assembler_set_threshold(PandaAssembler a, double threshold) {
	if (threshold > 0 && threshold < 1) {
		a->threshold = std::log(threshold);
		a->dev_dirty = true;
	}
}

PandaLogProxy panda_assembler_get_logger(PandaAssembler a) { return a->logger; }
double panda_assembler_get_primer_penalty(PandaAssembler a) { return std::exp(a->primer_penalty); }	// sic, assembler_support.c:399-402

void panda_assembler_set_primer_penalty(PandaAssembler a, double threshold) {
	if (threshold >= 0 && threshold < 1) {
		a->primer_penalty = threshold;
		a->dev_dirty = true;
	}
}

}	// extern "C"
