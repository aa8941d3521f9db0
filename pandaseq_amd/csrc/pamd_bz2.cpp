// pamd_bz2.cpp -- see pamd_bz2.h
#include "pamd_bz2.h"
#include <dlfcn.h>
#include <mutex>

namespace {
PamdBz2 g_bz2;
bool g_bz2_ok = false;
std::once_flag g_bz2_once;

void bind() {
	void *lib = nullptr;
	for (const char *name : { "libbz2.so.1.0", "libbz2.so.1", "libbz2.so" })
		if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr)
			break;
	if (lib == nullptr)
		return;
	g_bz2.bzdopen = reinterpret_cast<void *(*)(int, const char *)>(dlsym(lib, "BZ2_bzdopen"));
	g_bz2.bzopen = reinterpret_cast<void *(*)(const char *, const char *)>(dlsym(lib, "BZ2_bzopen"));
	g_bz2.bzRead = reinterpret_cast<int (*)(int *, void *, void *, int)>(dlsym(lib, "BZ2_bzRead"));
	g_bz2.bzWrite = reinterpret_cast<void (*)(int *, void *, void *, int)>(dlsym(lib, "BZ2_bzWrite"));
	g_bz2.bzclose = reinterpret_cast<void (*)(void *)>(dlsym(lib, "BZ2_bzclose"));
	g_bz2_ok = g_bz2.bzdopen && g_bz2.bzopen && g_bz2.bzRead && g_bz2.bzWrite && g_bz2.bzclose;
}
}	// namespace

const PamdBz2 *pamd_bz2() {
	std::call_once(g_bz2_once, bind);
	return g_bz2_ok ? &g_bz2 : nullptr;
}
