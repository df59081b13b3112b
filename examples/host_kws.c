/*
 * examples/host_kws.c -- the firmware's hardware-in-the-loop flow (firmware/src/app.c:167-216) written against the
 * reference's own C names, linked against libedison_hip.so instead of the STM32 firmware:
 *
 *     per frame:  audioCalcMFCCs(frame, &mfcc)  ->  mfccToNetInput(mfcc, 13, 31, f)        (app.c:190-193)
 *     then:       aiRunInference(netInput, netOutput)  ->  aiGetKeywordFromIndex(argmax)    (app.c:203)
 *
 * and, next to it, the batched entry point doing the same work in one call. Reads raw 16-bit little-endian PCM
 * (32000 samples) from the file given as argv[1]; prints the class of both paths.
 *
 *   cc examples/host_kws.c -Iinclude -Ledison_amd/csrc -ledison_hip -Wl,-rpath,$PWD/edison_amd/csrc -o host_kws
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edison_hip.h"

int main(int argc, char **argv)
{
	static int16_t audio[32000];
	if (argc < 2) { fprintf(stderr, "usage: %s utterance.pcm\n", argv[0]); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	size_t n = fread(audio, sizeof(int16_t), 32000, f);
	fclose(f);
	if (n < 31 * 1024) memset(audio + n, 0, (32000 - n) * sizeof(int16_t)); /* zero pad like kws_on_mcu.py:287-290 */

	/* ---- the firmware's call sequence */
	if (aiInitialize() != 0) return 1;
	audioInit();
	uint16_t in_x, in_y;
	aiGetInputShape(&in_x, &in_y); /* 13, 31 */
	for (uint16_t fr = 0; fr < in_y; fr++)
	{
		int16_t *mfcc;
		audioCalcMFCCs(audio + (size_t)fr * EDISON_FRAME_LEN, &mfcc);
		mfccToNetInput(mfcc, in_x, in_y, fr);
	}
	int8_t net_in[EDISON_NET_IN], net_out[EDISON_NET_OUT];
	memcpy(net_in, aiNnomGetInputBuffer(), sizeof(net_in));
	if (aiRunInference(net_in, net_out) != 0) return 1;
	int best = 0;
	for (int i = 1; i < EDISON_NET_OUT; i++) if (net_out[i] > net_out[best]) best = i;
	printf("firmware-style: %s (%d/127)\n", aiGetKeywordFromIndex((uint32_t)best), net_out[best]);

	/* ---- the batched entry point: one call, features never leave the GPU */
	int32_t am = -1;
	int8_t soft[EDISON_NET_OUT];
	if (edison_kws_batch(edison_global_ctx(), audio, 1, 32000, NULL, NULL, soft, &am) != EDISON_OK)
	{
		fprintf(stderr, "edison_kws_batch: %s\n", edison_last_error(edison_global_ctx()));
		return 1;
	}
	printf("batched:        %s (%d/127)\n", aiGetKeywordFromIndex((uint32_t)am), soft[am]);
	aiPrintInfo(); /* the net's description and the last inference time, as the firmware prints them (ai.c:189-196) */
	return best == am ? 0 : 3;
}
