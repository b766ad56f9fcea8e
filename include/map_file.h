/*
 * map_file.h -- read-only whole-file mapping helper.
 *
 * Same interface as the reference's helper (/root/reference/map_file.h:8-16), re-implemented:
 * the reference ships no licence, and north_star asks for the API to be retained so that
 * host code written against it keeps compiling.  Failure behaviour is the reference's:
 * perror(<filename>) and exit(EXIT_FAILURE) (/root/reference/map_file.c:22-40), which
 * includes the empty-file case (mmap of length 0 fails with EINVAL).
 */
#ifndef BWTS_MAP_FILE_H
#define BWTS_MAP_FILE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	void *sp, *ep;
} ptr_range;

ptr_range map_input_file(const char *filename);
void map_input_file2(const char *filename, void **start, long *len);
void unmap_file(ptr_range extent);

/* maps `path`, stores the base in ptr and the ELEMENT count (bytes / sizeof *ptr) in len */
#define map_in(ptr, len, path) map_input_file2(path, (void**)&ptr, &len), len /= sizeof(*ptr)

#ifdef __cplusplus
}
#endif

#endif
