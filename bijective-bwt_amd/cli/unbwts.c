/*
 * unbwts -- inverse bijective BWT of a file, on the GPU.
 *
 * Command-line contract of the reference program (/root/reference/unbwts.c:19-92, :136-176):
 *   unbwts <infile.bwts> [<outfile>]
 *   fewer than 2 args: two usage lines on stderr, exit 1                         (:21-25)
 *   outfile given: opened "wb"; failure prints a message + perror(name), exit 1  (:141-149)
 *   no outfile: "<infile>_XXXXXX" is created with mkstemps and
 *               "Writing to <name>" is printed on stdout                         (:152-158)
 * The transform itself (reference :31-86) is bwts_inverse_sink() from libbwts_hip.so; the text is written piece by
 * piece as it arrives from the GPU (reference: one fwrite of the whole buffer, :173).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "bwts.h"
#include "map_file.h"

static FILE *open_output(const char *explicit_name, const char *in_name)
{
	FILE *fp;

	if (explicit_name) {
		fp = fopen(explicit_name, "wb");
		if (!fp) {
			fprintf(stderr, "Couldn't open output file for writing\n");
			perror(explicit_name);
			exit(1);
		}
		return fp;
	}
	{
		char *name = NULL;
		int fd;

		if (asprintf(&name, "%s_XXXXXX", in_name) <= 0) {
			fprintf(stderr, "Allocating outfile name failed. Abort\n");
			exit(1);
		}
		fd = mkstemps(name, 0);
		printf("Writing to %s\n", name);
		fflush(stdout);
		fp = fd >= 0 ? fdopen(fd, "w") : NULL;
		if (!fp) {
			fprintf(stderr, "Couldn't open output file for writing\n");
			perror(name);
			exit(1);
		}
		free(name);
		return fp;
	}
}

static int write_piece(void *user, const uint8_t *data, uint64_t len)
{
	return fwrite(data, 1, (size_t)len, (FILE *)user) == (size_t)len ? 0 : 1;
}

int main(int argc, char **argv)
{
	unsigned char *bwts;
	long len;
	bwts_ctx *ctx;
	FILE *fp;
	int rc;
	const char *dev = getenv("BWTS_DEVICE");
	const char *show = getenv("BWTS_TIMINGS");

	if (argc < 2) {
		fprintf(stderr, "Usage: unbwts <infile.bwts> [<outfile>]\n");
		fprintf(stderr, "If output file name is unspecified, a name is generated\n");
		exit(1);
	}
	map_in(bwts, len, argv[1]);

	if ((rc = bwts_ctx_create(&ctx, dev ? atoi(dev) : 0)) != BWTS_OK) {
		fprintf(stderr, "unbwts: %s\n", bwts_strerror(rc));
		exit(1);
	}
	fp = open_output(argc < 3 ? NULL : argv[2], argv[1]);
	if ((rc = bwts_inverse_sink(ctx, bwts, (uint64_t)len, write_piece, fp)) != BWTS_OK) {
		fprintf(stderr, "unbwts: %s\n", bwts_strerror(rc));
		exit(1);
	}
	fclose(fp);
	if (show && show[0] == '1') {
		bwts_timings t;
		bwts_last_timings(ctx, &t);
		fprintf(stderr, "Transform (device) time %0.3f  H2D %0.3f  D2H+write %0.3f\n", 1e-3 * t.total_ms, 1e-3 * t.h2d_ms,
			1e-3 * t.d2h_ms);
	}
	bwts_ctx_destroy(ctx);
	return 0;
}
