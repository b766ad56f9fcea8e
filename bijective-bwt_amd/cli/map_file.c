/* map_file.c -- see include/map_file.h.  Behaviour follows /root/reference/map_file.c:16-59. */
#include "map_file.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static void die(const char *what)
{
	perror(what);
	exit(EXIT_FAILURE);
}

void map_input_file2(const char *filename, void **start, long *len)
{
	struct stat st;
	void *base;
	int fd = open(filename, O_RDONLY);

	if (fd < 0)
		die(filename);
	if (fstat(fd, &st) != 0)
		die(NULL);
	/* a zero-length mapping fails with EINVAL, as in the reference */
	base = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	if (base == MAP_FAILED)
		die(filename);
	close(fd);
	*start = base;
	*len = (long)st.st_size;
}

ptr_range map_input_file(const char *filename)
{
	ptr_range r;
	void *base;
	long bytes;

	map_input_file2(filename, &base, &bytes);
	r.sp = base;
	r.ep = (char *)base + bytes;
	return r;
}

void unmap_file(ptr_range extent)
{
	munmap(extent.sp, (size_t)((char *)extent.ep - (char *)extent.sp));
}
