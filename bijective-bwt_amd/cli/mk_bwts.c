/*
 * mk_bwts -- forward bijective BWT of a file, on the GPU.
 *
 * Command-line contract of the reference program (/root/reference/mk_bwts_sa.c:33-65):
 *   mk_bwts <infile> [<outfile.bwts>]
 *   fewer than 2 args: two usage lines on stderr, exit 1                  (:34-38)
 *   no outfile: the raw transform goes to standard output                 (:54)
 *   outfile cannot be opened: message + perror(name), exit 1              (:55-59)
 *   output is exactly the input's length: no header, no index             (:60)
 * The transform itself (reference :47-52) is bwts_forward() from libbwts_hip.so.
 * BWTS_TIMINGS=1 prints the reference's MARK_TIME phase lines (:13-22) on stderr,
 * taken from HIP events.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bwts.h"
#include "map_file.h"

static void fail(const char *what, int code)
{
	fprintf(stderr, "mk_bwts: %s: %s\n", what, bwts_strerror(code));
	exit(1);
}

int main(int argc, char **argv)
{
	unsigned char *text;
	long len;
	unsigned char *bwts;
	bwts_ctx *ctx;
	FILE *fp;
	int rc;
	const char *dev = getenv("BWTS_DEVICE");
	const char *show = getenv("BWTS_TIMINGS");

	if (argc < 2) {
		fprintf(stderr, "Usage: mk_bwts_sa <infile> [<outfile.bwts>]\n");
		fprintf(stderr, "If unspecified, output is written to standard output\n");
		exit(1);
	}
	map_in(text, len, argv[1]);

	bwts = (unsigned char *)malloc((size_t)len);
	if (!bwts) {
		perror("malloc");
		exit(1);
	}
	if ((rc = bwts_ctx_create(&ctx, dev ? atoi(dev) : 0)) != BWTS_OK)
		fail("cannot open GPU context", rc);
	if ((rc = bwts_forward(ctx, text, (uint64_t)len, bwts)) != BWTS_OK)
		fail("transform failed", rc);

	if (show && show[0] == '1') {
		bwts_timings t;
		bwts_last_timings(ctx, &t);
		/* same labels as MARK_TIME (mk_bwts_sa.c:50,124,168,190); seconds of device time */
		fprintf(stderr, "Suffix sort time %0.3f\n",
			1e-3 * (t.k[BWTS_K_KEYBUILD].ms + t.k[BWTS_K_RADIX_HIST].ms + t.k[BWTS_K_RADIX_SCAN].ms +
				t.k[BWTS_K_RADIX_SCATTER].ms + t.k[BWTS_K_HISTOGRAM].ms));
		fprintf(stderr, "Compute ISA time %0.3f\n", 1e-3 * t.k[BWTS_K_RERANK].ms);
		fprintf(stderr, "Fix sort order time %0.3f\n", 1e-3 * t.k[BWTS_K_LYNDON].ms);
		fprintf(stderr, "Generate BWTS time %0.3f\n", 1e-3 * (t.k[BWTS_K_EMIT].ms + t.k[BWTS_K_OTHER].ms));
		fprintf(stderr, "Transform (device) time %0.3f  H2D %0.3f  D2H %0.3f\n", 1e-3 * t.total_ms, 1e-3 * t.h2d_ms,
			1e-3 * t.d2h_ms);
	}
	bwts_ctx_destroy(ctx);

	fp = argc < 3 ? stdout : fopen(argv[2], "w");
	if (!fp) {
		fprintf(stderr, "Couldn't open BWTS file for writing\n");
		perror(argv[2]);
		exit(1);
	}
	fwrite(bwts, 1, (size_t)len, fp);
	if (fp != stdout)
		fclose(fp);
	return 0;
}
