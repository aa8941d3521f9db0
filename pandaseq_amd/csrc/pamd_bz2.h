// pamd_bz2.h -- libbz2 bound at run time.
//
// The reference links libbz2 (fileio.c:48-58,73-85: BZ2_bzdopen / BZ2_bzRead for inputs that begin with "BZ"; writer.c:108-151:
// BZ2_bzopen / BZ2_bzWrite for `-j` output).  An image may hold the library without its header (this one does), so the five
// functions are looked up in libbz2.so.1 when first needed, with the signatures bzlib.h 1.0 documents; where the library is
// absent the callers report that bzip2 is not available, as before.
#ifndef PAMD_BZ2_H
#define PAMD_BZ2_H

struct PamdBz2 {
	void *(*bzdopen)(int fd, const char *mode);					// BZ2_bzdopen
	void *(*bzopen)(const char *path, const char *mode);				// BZ2_bzopen
	int (*bzRead)(int *bzerror, void *file, void *buffer, int length);		// BZ2_bzRead
	void (*bzWrite)(int *bzerror, void *file, void *buffer, int length);		// BZ2_bzWrite
	void (*bzclose)(void *file);							// BZ2_bzclose
};
enum { PAMD_BZ_OK = 0, PAMD_BZ_STREAM_END = 4 };	// BZ_OK, BZ_STREAM_END

// the bound functions, or NULL when libbz2 cannot be loaded (thread-safe; looked up once)
const PamdBz2 *pamd_bz2();

#endif
