/*
 * legacy.c -- the reference firmware's own call surface (plain C), kept so that code written against
 * edison's C API links against libedison_hip.so unchanged. Every function here is a batch = 1 wrapper
 * around the batched GPU entry points of include/edison_hip.h; none of them computes on the CPU.
 *
 *   aiInitialize / aiGetInputShape / aiRunInference / aiGetKeywordFromIndex / aiGetKeywordCount
 *                                   firmware/src/ai/ai.c:112,205,227,243,248   (prototypes ai.h:74-80)
 *   aiNnomInit / aiNnomRunInference / aiNnomPredict / aiNnomGet{Input,Output}Buffer
 *                                   firmware/src/ai/ai_nnom.c:64,69,96,124,128
 *   mfccToNetInput / mfccToNetInputPush   firmware/src/app.c:675-695, 706-719
 *
 * Ownership mirrors the reference: a process-global model with static 403-byte input and 10-byte output
 * buffers (weights.h:136-137), not re-entrant; in_data / out_data are caller-owned.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"

/* firmware/src/ai/nnom/keywords.txt */
static const char *const g_keywords[EDISON_NET_OUT] = {"edison", "cinema", "bedroom", "office", "livingroom",
                                                       "kitchen", "on", "off", "_cold", "_noise"};

static edison_ctx *g_ctx = NULL;
static int8_t g_net_in[EDISON_NET_IN];   /* nnom_input_data[403] (weights.h:136) and app.c's netInput */
static int8_t g_net_out[EDISON_NET_OUT]; /* nnom_output_data[10] (weights.h:137)                      */

#define NNOM_INPUT_SCALE 1 /* weights.h:162-164 */
#define NNOM_INPUT_MIN (-128)
#define NNOM_INPUT_MAX 127

static void default_model_path(char *out, size_t cap)
{
	const char *env = getenv("EDISON_MODEL");
	if (env && env[0]) { snprintf(out, cap, "%s", env); return; }
	Dl_info info;
	out[0] = 0;
	if (dladdr((void *)&default_model_path, &info) && info.dli_fname)
	{
		snprintf(out, cap, "%s", info.dli_fname);
		char *slash = strrchr(out, '/');
		if (slash) *slash = 0; else snprintf(out, cap, ".");
		size_t n = strlen(out);
		snprintf(out + n, cap - n, "/../data/kws_nnom.ednn");
	}
}

edison_ctx *edison_global_ctx(void) { return g_ctx; }

int aiInitialize(void)
{
	if (g_ctx) return EDISON_OK;
	const char *dev = getenv("EDISON_DEVICE");
	edison_ctx *ctx = NULL;
	int r = edison_init(dev ? atoi(dev) : 0, &ctx);
	if (r != EDISON_OK)
	{
		fprintf(stderr, "aiInitialize: %s\n", edison_last_error(NULL));
		return r;
	}
	char path[1024];
	default_model_path(path, sizeof(path));
	r = edison_model_load(ctx, path);
	if (r != EDISON_OK)
	{
		fprintf(stderr, "aiInitialize: %s\n", edison_last_error(ctx));
		edison_shutdown(ctx);
		return r;
	}
	g_ctx = ctx;
	return EDISON_OK;
}

void aiNnomInit(void) { (void)aiInitialize(); }

void aiGetInputShape(uint16_t *x, uint16_t *y)
{
	*x = EDISON_NUM_MFCC;   /* ai.c:213 */
	*y = EDISON_UTT_FRAMES; /* ai.c:214 */
}

int aiNnomRunInference(void *in_data, void *out_data)
{
	if (!g_ctx)
	{
		int r = aiInitialize();
		if (r != EDISON_OK) return r;
	}
	memcpy(g_net_in, in_data, sizeof(g_net_in)); /* ai_nnom.c:74 */
	int r = edison_cnn_batch(g_ctx, g_net_in, 1, NULL, g_net_out, NULL);
	if (r != EDISON_OK) return r;
	memcpy(out_data, g_net_out, sizeof(g_net_out)); /* ai_nnom.c:80 */
	return EDISON_OK;
}

int aiRunInference(void *in_data, void *out_data) { return aiNnomRunInference(in_data, out_data); }

int aiNnomPredict(uint32_t *label, float *prob)
{
	/* nnom_predict (nnom_utils.c:258-305) on the static input buffer: run, first-max label, prob = max/sum */
	if (!g_ctx)
	{
		int r = aiInitialize();
		if (r != EDISON_OK) return r;
	}
	int32_t am = 0;
	int r = edison_cnn_batch(g_ctx, g_net_in, 1, NULL, g_net_out, &am);
	if (r != EDISON_OK) return r;
	int sum = 0;
	for (int i = 0; i < EDISON_NET_OUT; i++) sum += g_net_out[i];
	*label = (uint32_t)am;
	*prob = sum != 0 ? (float)g_net_out[am] / (float)sum : 0.0f;
	return EDISON_OK;
}

int8_t *aiNnomGetInputBuffer(void) { return g_net_in; }
int8_t *aiNnomGetOutputBuffer(void) { return g_net_out; }

const char *aiGetKeywordFromIndex(uint32_t idx) { return idx < EDISON_NET_OUT ? g_keywords[idx] : ""; }
uint32_t aiGetKeywordCount(void) { return EDISON_NET_OUT; }

void mfccToNetInput(int16_t *mfcc, uint16_t in_x, uint16_t in_y, uint32_t xoffset)
{
	(void)in_y;
	for (int c = 0; c < in_x; c++)
	{
		if ((size_t)xoffset * in_x + (size_t)c >= sizeof(g_net_in)) break;
		int16_t t = (int16_t)(mfcc[c] / NNOM_INPUT_SCALE); /* app.c:689 */
		t = (t > NNOM_INPUT_MAX) ? NNOM_INPUT_MAX : t;
		t = (t < NNOM_INPUT_MIN) ? NNOM_INPUT_MIN : t;
		g_net_in[xoffset * in_x + c] = (int8_t)t;
	}
}

void mfccToNetInputPush(int16_t *mfcc, uint16_t in_x, uint16_t in_y)
{
	if ((size_t)in_x * in_y > sizeof(g_net_in) || in_y == 0) return;
	memmove(g_net_in, g_net_in + in_x, (size_t)(in_y - 1) * in_x); /* app.c:711-715: drop the oldest row */
	mfccToNetInput(mfcc, in_x, in_y, (uint32_t)(in_y - 1));        /* app.c:718: append the newest     */
}

/*
 * firmware/src/audioprocessing.h:21-22. Caller-owned 1024-sample frame in, pointer to a callee-owned static buffer
 * of 32 int16 out, valid until the next call (bufDctInline, audioprocessing.c:80,210). The numbers are the
 * firmware's own: MFCC variant C, the Q15/Q31 integer pipeline of audioprocessing.c:116-215 on the GPU
 * (mfcc_q15_kernels.hip).
 */
static int16_t g_mfcc_q15_out[EDISON_NUM_MEL];

void audioInit(void) { (void)aiInitialize(); }

void audioCalcMFCCs(int16_t *inp, int16_t **oup)
{
	*oup = g_mfcc_q15_out;
	int r = g_ctx ? EDISON_OK : aiInitialize();
	if (r == EDISON_OK) r = edison_mfcc_q15_batch(g_ctx, inp, 1, EDISON_FRAME_LEN, EDISON_NUM_MEL, g_mfcc_q15_out, NULL);
	if (r != EDISON_OK)
	{
		/* the firmware calls Error_Handler() on failure (audioprocessing.c:105,200); here: report and zero */
		fprintf(stderr, "audioCalcMFCCs: GPU MFCC failed: %s\n", edison_last_error(g_ctx));
		memset(g_mfcc_q15_out, 0, sizeof(g_mfcc_q15_out));
	}
}

int edison_mfcc_frame(const int16_t *frame1024, int variant, float *out32)
{
	if (!g_ctx)
	{
		int r = aiInitialize();
		if (r != EDISON_OK) return r;
	}
	return edison_mfcc_batch(g_ctx, frame1024, 1, EDISON_FRAME_LEN, variant, EDISON_NUM_MEL, out32, NULL, 1.0f);
}
