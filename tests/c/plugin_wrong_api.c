/* a plugin built for another PANDA_API: panda_module_load() must refuse it (module.c:263-270) */
#include <stdbool.h>
#include <stddef.h>
int wrong_api_LTX_api = 2;
bool wrong_api_LTX_opener(void *logger, const char *args, void **precheck, void **check, void **user_data, void **destroy) {
	(void) logger; (void) args; (void) precheck; (void) check; (void) user_data; (void) destroy;
	return true;
}
