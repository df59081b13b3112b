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
#include <time.h>

#include "../../include/edison_hip.h"
#include "edison_fsm_core.h"

/* firmware/src/ai/nnom/keywords.txt */
static const char *const g_keywords[EDISON_NET_OUT] = {"edison", "cinema", "bedroom", "office", "livingroom",
                                                       "kitchen", "on", "off", "_cold", "_noise"};

static edison_ctx *g_ctx = NULL;
static int8_t g_net_in[EDISON_NET_IN];   /* nnom_input_data[403] (weights.h:136) and app.c's netInput */
static int8_t g_net_out[EDISON_NET_OUT]; /* nnom_output_data[10] (weights.h:137)                      */
static uint32_t g_last_inference_us;     /* lastInferenceTimeUs (ai.c:86)                             */

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
	struct timespec t0, t1; /* utilTic / utilToc around the inference (ai.c:233-239) */
	clock_gettime(CLOCK_MONOTONIC, &t0);
	int r = edison_cnn_batch(g_ctx, g_net_in, 1, NULL, g_net_out, NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	g_last_inference_us = (uint32_t)((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L);
	if (r != EDISON_OK) return r;
	memcpy(out_data, g_net_out, sizeof(g_net_out)); /* ai_nnom.c:80 */
	return EDISON_OK;
}

/* ai_nnom.h:7 (the firmware prints NNoM's model_stat table): the loaded graph, layer by layer */
void aiNnomPrintInfo(void)
{
	edison_net_info info;
	if (g_ctx && edison_net_get_info(g_ctx, &info) == EDISON_OK)
	{
		printf(" NNoM int8 graph on the GPU: input (%d, %d, %d), %d compute layers, %d outputs%s\n", info.in_h, info.in_w,
		       info.in_c, info.n_layers, info.n_out, info.accelerated ? ", kws_conv on the matrix cores" : ", general kernel");
		for (int i = 0; i < info.n_layers; i++)
		{
			edison_net_layer_info_t li;
			static const char *const names[5] = {"?", "Conv2D", "MaxPool", "Dense", "Softmax"};
			if (edison_net_layer_info(g_ctx, i, &li) != EDISON_OK) break;
			printf("  #%d %-8s -> (%d, %d, %d)%s\n", i + 1, names[li.type >= 1 && li.type <= 4 ? li.type : 0], li.out_h, li.out_w, li.out_c,
			       li.relu ? " ReLU" : "");
		}
	}
	else
		printf(" no model loaded (aiInitialize)\n");
}

/* ai.c:189-196,261-267: the net's description and the duration of the last inference, on stdout */
void aiPrintInfo(void)
{
	printf("-------------------------------------------------------------\n");
	printf("AI net information\n");
	aiNnomPrintInfo();
	printf("\n\n last inference time: %.2fms\n", (float)g_last_inference_us / 1000.0);
}

/* ai_nnom.c:50-56 (NNOM_VERIFICATION): create the model and run it once on whatever the input buffer holds */
void aiNnomTest(void)
{
	int8_t out[EDISON_NET_OUT];
	if (!g_ctx && aiInitialize() != EDISON_OK) return;
	(void)aiNnomRunInference(g_net_in, out);
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

/* ---------------------------------------------------------------------------------------------------------------
 * edisonFSM (firmware/src/app.c:727-928) without the LED strip: the keyword roles are the firmware's tables
 * (ediLocations / ediValues, app.c:135-147; EDI_WAKEWORD, app.c:50), resolved by name against the keyword list
 * exactly like the EDI_RESET state does (app.c:770-784).
 */
#define EDI_LOC_TIMEOUT_MS 5000u /* app.c:48 */
static const char *const g_fsm_locations[] = {"cinema", "bedroom", "office", "livingroom", "kitchen", NULL};
static const char *const g_fsm_values[] = {"off", "on", NULL};
static const char *const g_fsm_wakeword = "edison";

static int fsm_role(const char *const *names, uint32_t keyword_idx)
{
	if (keyword_idx >= EDISON_NET_OUT) return 0;
	for (int i = 0; names[i]; i++)
		if (strcmp(names[i], g_keywords[keyword_idx]) == 0) return 1;
	return 0;
}

void edison_fsm_init(edison_fsm *f)
{
	if (!f) return;
	memset(f, 0, sizeof(*f));
	f->state = EDISON_FSM_RESET;
	f->wake_idx = f->loc_idx = f->val_idx = f->last_loc = f->last_val = -1;
}

/* the roles of the ten classes for ed_fsm_step_core (edison_fsm_core.h) */
void edison_fsm_roles(int32_t *wake_idx, uint32_t *loc_mask, uint32_t *val_mask)
{
	int32_t w = -1;
	uint32_t lm = 0, vm = 0;
	for (uint32_t i = 0; i < EDISON_NET_OUT; i++)
	{
		if (strcmp(g_keywords[i], g_fsm_wakeword) == 0) w = (int32_t)i;
		if (fsm_role(g_fsm_locations, i)) lm |= 1u << i;
		if (fsm_role(g_fsm_values, i)) vm |= 1u << i;
	}
	if (wake_idx) *wake_idx = w;
	if (loc_mask) *loc_mask = lm;
	if (val_mask) *val_mask = vm;
}

int edison_fsm_step(edison_fsm *f, float pred_max, uint32_t pred_idx, uint32_t dt_us, double true_threshold)
{
	if (!f) return EDISON_E_ARGUMENT;
	ed_fsm_roles_t roles;
	edison_fsm_roles(&roles.wake_idx, &roles.loc_mask, &roles.val_mask);
	const int r = ed_fsm_step_core(f, (double)pred_max > true_threshold, pred_idx, dt_us, &roles);
	return r < 0 ? EDISON_E_ARGUMENT : r;
}
