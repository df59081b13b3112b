/*
 * examples/host_mfcc_pipeline.c -- several INDEPENDENT MFCC batches (the rows of the reference's batch_mfcc,
 * audio/edison/mfcc/mfcc_utils.py:75-131, or batches from different producers: any address, own output each) from a plain
 * C host, three ways, with wall times and a bit-for-bit comparison of the results:
 *
 *   (a) one edison_mfcc_batch_dev call per batch on the context's stream            -- the serial launch sequence
 *   (b) ONE edison_mfcc_batches_dev call for the whole list                         -- one launch keeps the chip busy across batches
 *   (c) one call per batch on the context's two queues (edison_queues_calibrate once, then edison_queues_fork /
 *       edison_mfcc_batch_queue_dev / edison_queues_join)                           -- the next launch in flight while this one drains
 *
 * (b) is the fast one (0.41-0.42 of 8 TB/s against 0.37-0.39 for (a) at 65 536 frames per batch). (c) gains +1 ... +5 % over (a) when the
 * calibration finds a pair of streams whose hardware queues lie well, and IS (a) when it does not -- which two HIP streams overlap
 * profitably is decided by where runtime and driver put their hardware queues, so the library measures it
 * (profiles/r05_mfcc_two_queues_notes.txt).
 *
 *   cc examples/host_mfcc_pipeline.c -Iinclude -Ledison_amd/csrc -ledison_hip -Wl,-rpath,$PWD/edison_amd/csrc -o host_mfcc_pipeline
 *   ./host_mfcc_pipeline [n_batches = 8] [frames_per_batch = 65536] [repetitions = 50]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "edison_hip.h"

#define CHECK(x) do { int r_ = (x); if (r_ != EDISON_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, edison_last_error(ctx)); return 1; } } while (0)

static double now_ms(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static double median5(double *v)
{
	for (int i = 0; i < 5; i++) for (int j = i + 1; j < 5; j++) if (v[j] < v[i]) { double t = v[i]; v[i] = v[j]; v[j] = t; }
	return v[2];
}

int main(int argc, char **argv)
{
	const int nb = argc > 1 ? atoi(argv[1]) : 8;
	const int64_t nf = argc > 2 ? atoll(argv[2]) : 65536;
	const int reps = argc > 3 ? atoi(argv[3]) : 50;
	if (nb < 1 || nb > 64 || nf < 1 || nf > (1 << 20) || reps < 1) { fprintf(stderr, "usage: %s [n_batches 1..64] [frames 1..2^20] [repetitions]\n", argv[0]); return 2; }
	edison_ctx *ctx = NULL;
	if (edison_init(0, &ctx) != EDISON_OK) { fprintf(stderr, "edison_init: %s\n", edison_last_error(NULL)); return 1; }

	/* the batches: separate allocations, own outputs (a, b, c each get their own set of outputs) */
	const size_t in_bytes = (size_t)nf * EDISON_FRAME_LEN * sizeof(int16_t), out_bytes = (size_t)nf * EDISON_NUM_MFCC * sizeof(float);
	int16_t *host = (int16_t *)malloc(in_bytes);
	const int16_t *audio[64];
	float *out_a[64], *out_b[64], *out_c[64];
	unsigned s = 12345u;
	for (int b = 0; b < nb; b++)
	{
		for (size_t i = 0; i < (size_t)nf * EDISON_FRAME_LEN; i++) { s = s * 1664525u + 1013904223u; host[i] = (int16_t)((int)(s >> 18) - 8192); }
		void *p;
		CHECK(edison_dev_alloc(ctx, in_bytes, &p)); audio[b] = (const int16_t *)p;
		CHECK(edison_dev_upload(ctx, p, host, in_bytes));
		CHECK(edison_dev_alloc(ctx, out_bytes, &p)); out_a[b] = (float *)p;
		CHECK(edison_dev_alloc(ctx, out_bytes, &p)); out_b[b] = (float *)p;
		CHECK(edison_dev_alloc(ctx, out_bytes, &p)); out_c[b] = (float *)p;
	}
	free(host);

	/* (c): which of the context's candidate streams make a pair worth using, measured on batch 0 (~0.17 s) */
	double cal_serial_us = 0, cal_best_us = 0;
	int cal_pair = 0;
	CHECK(edison_queues_calibrate(ctx, audio[0], nf, EDISON_FRAME_LEN, EDISON_MFCC_B, &cal_serial_us, &cal_best_us, &cal_pair));

	enum { PASSES = 6 };             /* pass 0 warms up (clocks, code objects); the medians of passes 1..5 are reported: a, b, c interleaved */
	double ta[PASSES], tb[PASSES], tc[PASSES];
	for (int pass = 0; pass < PASSES; pass++)
	{
		double t_a, t_b, t_c;
		/* (a) serial */
		double t0 = now_ms();
		for (int r = 0; r < reps; r++)
			for (int b = 0; b < nb; b++)
				CHECK(edison_mfcc_batch_dev(ctx, audio[b], nf, EDISON_FRAME_LEN, EDISON_MFCC_B, EDISON_NUM_MFCC, out_a[b], NULL, 1.0f));
		CHECK(edison_sync(ctx));
		t_a = now_ms() - t0;
		/* (b) one launch per list */
		t0 = now_ms();
		for (int r = 0; r < reps; r++)
			CHECK(edison_mfcc_batches_dev(ctx, nb, audio, nf, EDISON_FRAME_LEN, EDISON_MFCC_B, EDISON_NUM_MFCC, out_b, NULL, 1.0f));
		CHECK(edison_sync(ctx));
		t_b = now_ms() - t0;
		/* (c) two launches in flight: fork the context's two queues from its stream, alternate, join */
		t0 = now_ms();
		CHECK(edison_queues_fork(ctx));
		for (int r = 0; r < reps; r++)
			for (int b = 0; b < nb; b++)
				CHECK(edison_mfcc_batch_queue_dev(ctx, (r * nb + b) & 1, audio[b], nf, EDISON_FRAME_LEN, EDISON_MFCC_B, EDISON_NUM_MFCC, out_c[b], NULL, 1.0f));
		CHECK(edison_queues_join(ctx));
		CHECK(edison_sync(ctx));
		t_c = now_ms() - t0;
		ta[pass] = t_a; tb[pass] = t_b; tc[pass] = t_c;
	}
	const double t_a = median5(ta + 1), t_b = median5(tb + 1), t_c = median5(tc + 1);

	/* the three ways must agree bit for bit */
	float *ha = (float *)malloc(out_bytes), *hb = (float *)malloc(out_bytes);
	int same_b = 1, same_c = 1;
	for (int b = 0; b < nb; b++)
	{
		CHECK(edison_dev_download(ctx, ha, out_a[b], out_bytes));
		CHECK(edison_dev_download(ctx, hb, out_b[b], out_bytes));
		same_b = same_b && memcmp(ha, hb, out_bytes) == 0;
		CHECK(edison_dev_download(ctx, hb, out_c[b], out_bytes));
		same_c = same_c && memcmp(ha, hb, out_bytes) == 0;
	}
	const double n = (double)reps * nb, bytes = 2100.0 * (double)nf;
	printf("{\"batches\": %d, \"frames_per_batch\": %lld, \"repetitions\": %d, "
	       "\"serial_us_per_batch\": %.2f, \"list_us_per_batch\": %.2f, \"two_queues_us_per_batch\": %.2f, "
	       "\"serial_frac_of_8TBs\": %.4f, \"list_frac_of_8TBs\": %.4f, \"two_queues_frac_of_8TBs\": %.4f, "
	       "\"calibration\": {\"serial_us\": %.2f, \"kept_us\": %.2f, \"pair\": %d}, "
	       "\"list_equals_serial\": %s, \"two_queues_equals_serial\": %s}\n",
	       nb, (long long)nf, reps, t_a * 1e3 / n, t_b * 1e3 / n, t_c * 1e3 / n, bytes / (t_a * 1e-3 / n) / 8e12, bytes / (t_b * 1e-3 / n) / 8e12,
	       bytes / (t_c * 1e-3 / n) / 8e12, cal_serial_us, cal_best_us, cal_pair, same_b ? "true" : "false", same_c ? "true" : "false");
	free(ha); free(hb);
	edison_shutdown(ctx);
	return same_b && same_c ? 0 : 3;
}
