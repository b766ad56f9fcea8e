/*
 * mk_bwts -- forward bijective BWT of a file, on the GPU.
 *
 * Command-line contract of the reference program (/root/reference/mk_bwts_sa.c:33-65):
 *   mk_bwts <infile> [<outfile.bwts>]
 *   fewer than 2 args: two usage lines on stderr, exit 1                  (:34-38)
 *   no outfile: the raw transform goes to standard output                 (:54)
 *   outfile cannot be opened: message + perror(name), exit 1              (:55-59)
 *   output is exactly the input's length: no header, no index             (:60)
 * Built with -DBWTS_AUTONAME this source is mk_bwts_new_algo, the program the reference's `make test` drives
 * (/root/reference/mk_bwts_new_algo.c:37-65, :192-234): same transform, but without an outfile the output goes to
 * "<infile>_XXXXXX.bwts" (mkstemps) and "Writing to <name>" is printed on stdout (:210-216).
 *
 * The transform itself (reference :47-52) is bwts_forward_sink() from libbwts_hip.so: the output is written piece by
 * piece as it arrives from the GPU, so no second n-byte buffer exists on the host.
 * Phase lines of the reference's MARK_TIME (:13-22, :50,:124,:168,:190,:62) go to stderr when built with
 * -DSHOW_TIMINGS (as in the reference, Makefile:2-3) or run with BWTS_TIMINGS=1; device phases come from HIP events.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "bwts.h"
#include "map_file.h"
#include "cli_report.h"

static void fail(const char *what, int code)
{
	fprintf(stderr, "mk_bwts: %s: %s\n", what, bwts_strerror(code));
	exit(1);
}

#define now_s cli_now_s
static double write_s;

/* The output is opened by the first piece that arrives (the reference computes first and opens afterwards, mk_bwts_sa.c:47-60):
 * by then the whole input has been read, so `mk_bwts f f` works, and a transform that fails leaves no truncated destination. */
struct out_file {
	const char *explicit_name, *in_name;
	FILE *fp;
	char *auto_name;        /* name made by mkstemps (removed again if the transform fails) */
	int failed;
};

static void open_output(struct out_file *o)
{
	if (o->explicit_name) {
#ifdef BWTS_AUTONAME
		o->fp = fopen(o->explicit_name, "wb");
#else
		o->fp = fopen(o->explicit_name, "w");
#endif
		if (!o->fp) {
			fprintf(stderr, "Couldn't open BWTS file for writing\n");
			perror(o->explicit_name);
			o->failed = 1;
		}
		return;
	}
#ifdef BWTS_AUTONAME
	{
		int fd;

		if (asprintf(&o->auto_name, "%s_XXXXXX.bwts", o->in_name) <= 0) {
			fprintf(stderr, "Allocating outfile name failed. Abort\n");
			o->auto_name = NULL;
			o->failed = 1;
			return;
		}
		fd = mkstemps(o->auto_name, 5);
		printf("Writing to %s\n", o->auto_name);
		fflush(stdout);
		o->fp = fd >= 0 ? fdopen(fd, "w") : NULL;
		if (!o->fp) {
			fprintf(stderr, "Couldn't open BWTS file for writing\n");
			perror(o->auto_name);
			o->failed = 1;
		}
	}
#else
	o->fp = stdout;
#endif
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
	t0 = now_s();
	done = fwrite(data, 1, (size_t)len, o->fp);
	write_s += now_s() - t0;
	if (done != (size_t)len) {
		perror("write");
		o->failed = 1;
		return 1;
	}
	return 0;
}

int main(int argc, char **argv)
{
	unsigned char *text;
	long len;
	bwts_ctx *ctx;
	struct out_file o;
	bwts_timings t;
	int rc;
	const char *dev = getenv("BWTS_DEVICE");
	const char *show_env = getenv("BWTS_TIMINGS");
#ifdef SHOW_TIMINGS
	const int show = 1;
#else
	const int show = show_env && show_env[0] == '1';
#endif
	double t_start, t_wall;
	struct cli_marks marks;

	marks.main_start = now_s();
	if (argc < 2) {
		fprintf(stderr, "Usage: mk_bwts_sa <infile> [<outfile.bwts>]\n");
#ifdef BWTS_AUTONAME
		fprintf(stderr, "If outfile is not supplied, a unique file name is generated\n");
#else
		fprintf(stderr, "If unspecified, output is written to standard output\n");
#endif
		exit(1);
	}
	map_in(text, len, argv[1]);
	marks.mapped = now_s();

	if ((rc = bwts_ctx_create(&ctx, dev ? atoi(dev) : 0)) != BWTS_OK)
		fail("cannot open GPU context", rc);
	marks.ctx_ready = now_s();
	if (show)
		bwts_set_timing(ctx, 2);
	memset(&o, 0, sizeof o);
	o.explicit_name = argc < 3 ? NULL : argv[2];
	o.in_name = argv[1];

	t_start = now_s();
	rc = bwts_forward_sink(ctx, text, (uint64_t)len, write_piece, &o);
	if (rc == BWTS_OK && o.fp && (o.fp != stdout ? fclose(o.fp) : fflush(o.fp)) != 0) {
		perror("write");
		o.failed = 1;
	}
	if (rc != BWTS_OK || o.failed) {
		/* nothing half-written stays behind under a name this run made up */
		if (o.auto_name)
			unlink(o.auto_name);
		if (rc != BWTS_OK && !o.failed)
			fail("transform failed", rc);
		exit(1);
	}
	marks.done = now_s();
	marks.write_s = write_s;
	t_wall = marks.done - t_start;

	if (show) {
		bwts_last_timings(ctx, &t);
		/* the reference's five labels (mk_bwts_sa.c:50,124,168,190,62); seconds.  The engine has no separate ISA or
		 * fix-up pass: "Compute ISA" is the regrouping / rank work of the doubling sort, "Fix sort order" the factor
		 * search (cyclic ranking needs no fix-up), "Write BWTS" the copy out of the GPU overlapped with fwrite. */
		fprintf(stderr, "Suffix sort time %0.3f\n",
			1e-3 * (t.k[BWTS_K_KEYBUILD].ms + t.k[BWTS_K_RADIX_HIST].ms + t.k[BWTS_K_RADIX_SCAN].ms +
				t.k[BWTS_K_RADIX_SCATTER].ms + t.k[BWTS_K_HISTOGRAM].ms));
		fprintf(stderr, "Compute ISA time %0.3f\n", 1e-3 * t.k[BWTS_K_RERANK].ms);
		fprintf(stderr, "Fix sort order time %0.3f\n", 1e-3 * t.k[BWTS_K_LYNDON].ms);
		fprintf(stderr, "Generate BWTS time %0.3f\n", 1e-3 * (t.k[BWTS_K_EMIT].ms + t.k[BWTS_K_OTHER].ms));
		fprintf(stderr, "Write BWTS time %0.3f\n", 1e-3 * t.d2h_ms);
		fprintf(stderr, "Transform (device) time %0.3f  H2D %0.3f  D2H+write %0.3f (fwrite %0.3f)  wall %0.3f\n", 1e-3 * t.total_ms,
			1e-3 * t.h2d_ms, 1e-3 * t.d2h_ms, write_s, t_wall);
		/* what a one-shot run pays outside kernels and copies */
		cli_report_host_costs(stderr, &t);
	}
	/* (the reference frees nothing and lets exit() do it, mk_bwts_sa.c:64; a library context is handed back properly, and the
	 * time that takes is reported) */
	bwts_ctx_destroy(ctx);
	marks.destroyed = now_s();
	if (show)
		cli_report_process(stderr, &marks);
	return 0;
}
