/*
 * cli_report.h -- the timing lines both CLIs print under BWTS_TIMINGS=1 (or -DSHOW_TIMINGS) beyond the reference's own MARK_TIME
 * labels (/root/reference/mk_bwts_sa.c:13-22): where a one-shot process spends its wall time outside kernels and copies.
 */
#ifndef BWTS_CLI_REPORT_H
#define BWTS_CLI_REPORT_H
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "bwts.h"

static double cli_now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* marks of one run, seconds on CLOCK_MONOTONIC (system-wide: a parent's clock reading compares with ours) */
struct cli_marks {
	double main_start;      /* first statement of main()                          */
	double mapped;          /* input mapped                                        */
	double ctx_ready;       /* bwts_ctx_create returned                            */
	double done;            /* transform finished, output closed                   */
	double destroyed;       /* bwts_ctx_destroy returned                           */
	double write_s;         /* summed time inside fwrite                           */
};

/* what the process paid outside kernels and copies (bwts_timings.host_ms) */
static void cli_report_host_costs(FILE *f, const bwts_timings *t)
{
	fprintf(f, "Start-up time %0.3f  (HIP runtime + context %0.3f, code object load %0.3f)  allocation time %0.3f  (device in/out %0.3f, "
		"pinned staging %0.3f, arenas %0.3f; %0.1f GiB on the device)\n",
		1e-3 * (t->host_ms[BWTS_H_INIT] + t->host_ms[BWTS_H_MODULE]), 1e-3 * t->host_ms[BWTS_H_INIT], 1e-3 * t->host_ms[BWTS_H_MODULE],
		1e-3 * (t->host_ms[BWTS_H_IO_ALLOC] + t->host_ms[BWTS_H_STAGING_ALLOC] + t->host_ms[BWTS_H_ARENA_ALLOC]),
		1e-3 * t->host_ms[BWTS_H_IO_ALLOC], 1e-3 * t->host_ms[BWTS_H_STAGING_ALLOC], 1e-3 * t->host_ms[BWTS_H_ARENA_ALLOC],
		(double)t->device_bytes / (double)(1ull << 30));
}

/* Where the wall time of the whole process went, mark by mark.  BWTS_T0_NS (set by a timing harness to its CLOCK_MONOTONIC reading
 * just before it started us, in nanoseconds) adds what passed before main(): exec, the dynamic loader, the HIP library's own
 * initialisers.  What follows the last line -- exit handlers, the driver taking the process's device memory back -- is the
 * harness's wall time minus "since launch". */
static void cli_report_process(FILE *f, const struct cli_marks *m)
{
	const char *t0 = getenv("BWTS_T0_NS");
	const double end = cli_now_s();

	fprintf(f, "Process time: map input %0.3f  context %0.3f  transform+output %0.3f  (fwrite %0.3f)  teardown %0.3f  main() %0.3f",
		m->mapped - m->main_start, m->ctx_ready - m->mapped, m->done - m->ctx_ready, m->write_s, m->destroyed - m->done,
		end - m->main_start);
	if (t0 && t0[0]) {
		const double launch = 1e-9 * strtod(t0, NULL);
		fprintf(f, "  before main() %0.3f  since launch %0.3f", m->main_start - launch, end - launch);
	}
	fprintf(f, "\n");
}
#endif
