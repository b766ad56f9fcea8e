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
#include "cli_report.h"

static double write_s;

/* The output is opened by the first piece that arrives (the reference computes first and calls write_out() afterwards,
 * unbwts.c:62-89): by then the whole input has been read, so `unbwts f f` works, and a failing transform leaves nothing behind. */
struct out_file {
	const char *explicit_name, *in_name;
	FILE *fp;
	char *auto_name;
	int failed;
};

static void open_output(struct out_file *o)
{
	int fd;

	if (o->explicit_name) {
		o->fp = fopen(o->explicit_name, "wb");
		if (!o->fp) {
			fprintf(stderr, "Couldn't open output file for writing\n");
			perror(o->explicit_name);
			o->failed = 1;
		}
		return;
	}
	if (asprintf(&o->auto_name, "%s_XXXXXX", o->in_name) <= 0) {
		fprintf(stderr, "Allocating outfile name failed. Abort\n");
		o->auto_name = NULL;
		o->failed = 1;
		return;
	}
	fd = mkstemps(o->auto_name, 0);
	printf("Writing to %s\n", o->auto_name);
	fflush(stdout);
	o->fp = fd >= 0 ? fdopen(fd, "w") : NULL;
	if (!o->fp) {
		fprintf(stderr, "Couldn't open output file for writing\n");
		perror(o->auto_name);
		o->failed = 1;
	}
}

static int write_piece(void *user, const uint8_t *data, uint64_t len)
{
	struct out_file *o = (struct out_file *)user;
	double t0;
	size_t done;

	if (!o->fp && !o->failed)
		open_output(o);
	if (o->failed)
		return 1;
	t0 = cli_now_s();
	done = fwrite(data, 1, (size_t)len, o->fp);
	write_s += cli_now_s() - t0;
	if (done != (size_t)len) {
		perror("write");
		o->failed = 1;
		return 1;
	}
	return 0;
}

int main(int argc, char **argv)
{
	unsigned char *bwts;
	long len;
	bwts_ctx *ctx;
	struct out_file o;
	int rc;
	const char *dev = getenv("BWTS_DEVICE");
	const char *show_env = getenv("BWTS_TIMINGS");
#ifdef SHOW_TIMINGS
	const int show = 1;
#else
	const int show = show_env && show_env[0] == '1';
#endif
	struct cli_marks marks;
	bwts_timings t;

	marks.main_start = cli_now_s();
	if (argc < 2) {
		fprintf(stderr, "Usage: unbwts <infile.bwts> [<outfile>]\n");
		fprintf(stderr, "If output file name is unspecified, a name is generated\n");
		exit(1);
	}
	map_in(bwts, len, argv[1]);
	marks.mapped = cli_now_s();

	if ((rc = bwts_ctx_create(&ctx, dev ? atoi(dev) : 0)) != BWTS_OK) {
		fprintf(stderr, "unbwts: %s\n", bwts_strerror(rc));
		exit(1);
	}
	marks.ctx_ready = cli_now_s();
	memset(&o, 0, sizeof o);
	o.explicit_name = argc < 3 ? NULL : argv[2];
	o.in_name = argv[1];
	rc = bwts_inverse_sink(ctx, bwts, (uint64_t)len, write_piece, &o);
	if (rc == BWTS_OK && o.fp && fclose(o.fp) != 0) {
		perror("write");
		o.failed = 1;
	}
	if (rc != BWTS_OK || o.failed) {
		if (o.auto_name)
			unlink(o.auto_name);        /* nothing half-written stays behind under a name this run made up */
		if (rc != BWTS_OK && !o.failed)
			fprintf(stderr, "unbwts: %s\n", bwts_strerror(rc));
		exit(1);
	}
	marks.done = cli_now_s();
	marks.write_s = write_s;
	/* The reference's unbwts has no timers; these are the forward program's extra lines, same format. */
	if (show) {
		bwts_last_timings(ctx, &t);
		fprintf(stderr, "Transform (device) time %0.3f  H2D %0.3f  D2H+write %0.3f (fwrite %0.3f)  wall %0.3f\n", 1e-3 * t.total_ms,
			1e-3 * t.h2d_ms, 1e-3 * t.d2h_ms, write_s, marks.done - marks.ctx_ready);
		cli_report_host_costs(stderr, &t);
	}
	bwts_ctx_destroy(ctx);
	marks.destroyed = cli_now_s();
	if (show)
		cli_report_process(stderr, &marks);
	return 0;
}
