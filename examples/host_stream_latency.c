/*
 * examples/host_stream_latency.c -- the firmware's continuous-microphone loop (firmware/src/app.c:288-371: one new frame
 * per audio event -> audioCalcMFCCs -> mfccToNetInputPush -> aiRunInference) as a plain C program on the streaming
 * entry points of libedison_hip.so, timing every push on the host: what a C host pays per frame, with no Python between
 * the microphone and the library. bench.py runs it and puts the numbers beside the ones taken through the ctypes mirror.
 *
 *   cc -O2 examples/host_stream_latency.c -Iinclude -Ledison_amd/csrc -ledison_hip -Wl,-rpath,$PWD/edison_amd/csrc -o host_stream_latency
 *   ./host_stream_latency [pushes=2000] [hop=512] [graph=0|1]       -> one JSON line
 *
 * Input: seeded Gaussian-like noise (sum of four LCG draws), sigma ~3000, the level of SURVEY.md 8(d)'s generator.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "edison_hip.h"

static double now_us(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp_double(const void *a, const void *b)
{
	const double x = *(const double *)a, y = *(const double *)b;
	return x < y ? -1 : x > y;
}

int main(int argc, char **argv)
{
	const int pushes = argc > 1 ? atoi(argv[1]) : 2000;
	const int hop = argc > 2 ? atoi(argv[2]) : 512;
	const int graph = argc > 3 ? atoi(argv[3]) : 0;
	const int warm = 100;
	if (pushes < 1 || pushes > 1000000 || hop < 2 || hop > EDISON_FRAME_LEN) { fprintf(stderr, "bad arguments\n"); return 2; }

	/* the firmware's own bring-up call (ai.c:112): context on GPU 0 (EDISON_DEVICE) + the shipped model (EDISON_MODEL) */
	if (aiInitialize() != 0) { fprintf(stderr, "aiInitialize failed (no gfx950 device / model?)\n"); return 1; }
	edison_ctx *ctx = edison_global_ctx();
	edison_stream_opts o;
	edison_stream_default_opts(&o);
	o.hop = hop;
	o.chunk_frames = 1;
	o.launch_mode = graph ? EDISON_STREAM_LAUNCH_GRAPH : EDISON_STREAM_LAUNCH_DIRECT;
	edison_stream *s = NULL;
	if (edison_stream_create_ex(ctx, &o, &s) != EDISON_OK) { fprintf(stderr, "stream: %s\n", edison_last_error(ctx)); return 1; }

	const size_t total = (size_t)(pushes + warm) * hop;
	int16_t *audio = (int16_t *)malloc(total * sizeof(int16_t));
	double *lat = (double *)malloc(sizeof(double) * pushes);
	if (!audio || !lat) return 1;
	unsigned lcg = 23u;
	for (size_t i = 0; i < total; i++)
	{
		int acc = 0;
		for (int k = 0; k < 4; k++) { lcg = lcg * 1664525u + 1013904223u; acc += (int)(lcg >> 16) - 32768; }
		int v = acc / 25; /* sum of 4 uniforms on +-32768: sigma 37 837 -> / 25 ~ 1500..3000 */
		audio[i] = (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v);
	}
	int8_t soft[EDISON_NET_OUT];
	int32_t am = -1;
	long hist[EDISON_NET_OUT] = {0};
	for (int i = 0; i < pushes + warm; i++)
	{
		const double t0 = now_us();
		if (edison_stream_push(s, audio + (size_t)i * hop, NULL, soft, &am) != EDISON_OK)
		{
			fprintf(stderr, "push: %s\n", edison_last_error(ctx));
			return 1;
		}
		const double t1 = now_us();
		if (i >= warm) { lat[i - warm] = t1 - t0; hist[am < 0 || am >= EDISON_NET_OUT ? 0 : am]++; }
	}
	qsort(lat, pushes, sizeof(double), cmp_double);
	printf("{\"p50\": %.1f, \"p90\": %.1f, \"p99\": %.1f, \"pushes\": %d, \"hop\": %d, \"graph\": %d, \"last_class\": \"%s\"}\n",
	       lat[pushes / 2], lat[(size_t)(pushes * 0.9)], lat[(size_t)(pushes * 0.99)], pushes, hop, graph,
	       aiGetKeywordFromIndex((uint32_t)(am < 0 ? 0 : am)));
	edison_stream_destroy(s);
	free(audio);
	free(lat);
	return 0;
}
