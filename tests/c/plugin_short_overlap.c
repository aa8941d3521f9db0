/*
 * plugin_short_overlap.c -- a check plugin written for the tests, in the shape pandaseq-plugin.h:33-35 gives
 * the reference's own plugins: <module>_LTX_api, _LTX_opener, _LTX_desc, _LTX_usage, _LTX_version.
 * It rejects assembled sequences whose overlap is shorter than its argument and, as a precheck, pairs whose
 * forward read is shorter than 3 bases.  tests/test_modules_cpu.py loads it with panda_module_load() (no GPU
 * needed for that); tests/test_compat_gpu.py runs it as a host callback around the device batch.
 */
#include <stdlib.h>
#include "pandaseq_amd_compat.h"

int short_overlap_LTX_api = PANDA_API;
const char *short_overlap_LTX_desc = "Reject sequences whose overlap is shorter than a number of bases.";
const char *short_overlap_LTX_usage = "short_overlap:bases";
const char *short_overlap_LTX_version = "0.1";

static int live_instances = 0;
int short_overlap_live_instances(void) {
	return live_instances;
}

static bool check(PandaLogProxy logger, const panda_result_seq *sequence, void *user_data) {
	(void) logger;
	return sequence->overlap >= *(size_t *) user_data;
}

static bool precheck(PandaLogProxy logger, const panda_seq_identifier *id, const panda_qual *forward, size_t forward_length, const panda_qual *reverse, size_t reverse_length, void *user_data) {
	(void) logger; (void) id; (void) forward; (void) reverse; (void) reverse_length; (void) user_data;
	return forward_length >= 3;
}

static void destroy_instance(void *data) {
	free(data);
	live_instances--;
}

bool short_overlap_LTX_opener(PandaLogProxy logger, const char *args, PandaPreCheck *precheck_out, PandaCheck *check_out, void **user_data, PandaDestroy *destroy) {
	char *end;
	long v;
	(void) logger;
	if (args == NULL || *args == '\0')
		return false;
	v = strtol(args, &end, 10);
	if (*end != '\0' || v < 0)
		return false;
	size_t *limit = malloc(sizeof(size_t));
	*limit = (size_t) v;
	live_instances++;
	*user_data = limit;
	*destroy = destroy_instance;
	*check_out = check;
	*precheck_out = precheck;
	return true;
}
