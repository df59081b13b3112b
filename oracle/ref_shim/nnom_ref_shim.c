/*
 * oracle/ref_shim/nnom_ref_shim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A thin C-ABI around the *unmodified* reference int8 CNN: NNoM 0.3.0 + CMSIS-NN
 * (portable "Cortex-M0/M3" branches) + the committed model `kws_nnom/weights.h`,
 * compiled from where those sources lie under /root/reference by oracle/Makefile
 * into oracle/_ref/libnnom_ref.so. Nothing of the reference is copied here; this
 * file only includes the reference headers and calls the reference entry points:
 *
 *   nnom_model_create()   firmware/src/ai/nnom/kws_nnom/weights.h:138
 *   model_run()           firmware/src/ai/nnom/src/core/nnom.c:1037
 *   model_set_callback()  firmware/src/ai/nnom/src/core/nnom.c:1043
 *
 * The call sequence mirrors aiNnomInit()/aiNnomRunInference()
 * (firmware/src/ai/ai_nnom.c:64-85): memcpy 403 B in, model_run, memcpy 10 B out.
 *
 * Used by: tests/ (as the checker), tests/golden/gen_fixtures.py (to produce the
 * committed golden vectors) and bench.py's cpu_baseline leg (kind "reference").
 */
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#include "nnom.h"
#include "kws_nnom/weights.h"

#define REF_MAX_LAYERS 16

static nnom_model_t *g_model = NULL;

/* per-layer capture (filled by the layer callback when g_dump != NULL) */
static int8_t *g_dump = NULL;       /* caller buffer                          */
static size_t  g_dump_cap = 0;      /* its capacity in bytes                  */
static size_t  g_dump_used = 0;
static int32_t g_layer_sizes[REF_MAX_LAYERS];
static int32_t g_layer_types[REF_MAX_LAYERS];
static int     g_layer_count = 0;

static nnom_status_t capture_cb(nnom_model_t *m, nnom_layer_t *layer)
{
	(void)m;
	if (g_dump == NULL || layer->out == NULL || layer->out->tensor == NULL)
		return NN_SUCCESS;
	size_t n = tensor_size(layer->out->tensor);
	if (g_layer_count < REF_MAX_LAYERS && g_dump_used + n <= g_dump_cap)
	{
		memcpy(g_dump + g_dump_used, layer->out->tensor->p_data, n);
		g_dump_used += n;
		g_layer_sizes[g_layer_count] = (int32_t)n;
		g_layer_types[g_layer_count] = (int32_t)layer->type;
		g_layer_count++;
	}
	return NN_SUCCESS;
}

/* NNoM prints its compile log through printf; keep stdout clean for JSON lines. */
int nnom_ref_init(void)
{
	if (g_model != NULL)
		return 0;
	fflush(stdout);
	FILE *saved = stdout;
	FILE *devnull = fopen("/dev/null", "w");
	if (devnull) stdout = devnull;
	g_model = nnom_model_create();
	if (devnull) { fflush(devnull); stdout = saved; fclose(devnull); }
	if (g_model == NULL)
		return -1;
	model_set_callback(g_model, capture_cb);
	return 0;
}

/* One inference: 403 int8 in (HWC [31][13][1]) -> 10 int8 out (softmax, Q0.7). */
int nnom_ref_run(const int8_t *in403, int8_t *out10)
{
	if (g_model == NULL && nnom_ref_init() != 0)
		return -1;
	g_dump = NULL;
	memcpy(nnom_input_data, in403, sizeof(nnom_input_data));
	int ret = (int)model_run(g_model);
	memcpy(out10, nnom_output_data, sizeof(nnom_output_data));
	return ret;
}

/*
 * One inference with every layer's output tensor captured back to back into
 * `dump` (order = execution order: input, conv1(+relu), pool1, conv2, pool2,
 * conv3, conv4, dense, softmax, output). sizes/types receive one entry per
 * layer; returns the number of layers captured, or a negative NNoM status.
 */
int nnom_ref_run_layers(const int8_t *in403, int8_t *dump, int32_t dump_cap,
                        int32_t *sizes, int32_t *types, int32_t max_layers)
{
	if (g_model == NULL && nnom_ref_init() != 0)
		return -1;
	g_dump = dump; g_dump_cap = (size_t)dump_cap; g_dump_used = 0; g_layer_count = 0;
	memcpy(nnom_input_data, in403, sizeof(nnom_input_data));
	int ret = (int)model_run(g_model);
	g_dump = NULL;
	if (ret != 0)
		return ret < 0 ? ret : -ret;
	for (int i = 0; i < g_layer_count && i < max_layers; i++)
	{
		sizes[i] = g_layer_sizes[i];
		types[i] = g_layer_types[i];
	}
	return g_layer_count;
}

/*
 * Batch loop for the CPU baseline and for bulk parity: n utterances, each 403
 * int8 in; writes the dense logits (10 int8, pre-softmax), the softmax output
 * (10 int8) and the first-max argmax over the softmax output
 * (nnom_predict's rule, firmware/src/ai/nnom/src/core/nnom_utils.c:275-284).
 * logits/softmax/argmax may each be NULL.
 */
int nnom_ref_run_batch(const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	if (g_model == NULL && nnom_ref_init() != 0)
		return -1;
	int8_t lay[16384];
	int32_t sizes[REF_MAX_LAYERS], types[REF_MAX_LAYERS];
	for (int64_t u = 0; u < n; u++)
	{
		int8_t out[10];
		if (logits != NULL)
		{
			int nl = nnom_ref_run_layers(in + u * 403, lay, (int32_t)sizeof(lay), sizes, types, REF_MAX_LAYERS);
			if (nl < 3) return -2;
			/* dense output = third tensor from the end (dense, softmax, output) */
			size_t off = 0;
			for (int i = 0; i < nl - 3; i++) off += (size_t)sizes[i];
			memcpy(logits + u * 10, lay + off, 10);
			memcpy(out, nnom_output_data, 10);
		}
		else
		{
			int r = nnom_ref_run(in + u * 403, out);
			if (r != 0) return r < 0 ? r : -r;
		}
		if (softmax != NULL) memcpy(softmax + u * 10, out, 10);
		if (argmax != NULL)
		{
			int best = 0; int8_t mx = out[0];
			for (int i = 1; i < 10; i++) if (out[i] > mx) { mx = out[i]; best = i; }
			argmax[u] = best;
		}
	}
	return 0;
}

size_t nnom_ref_mem_stat(void) { return nnom_mem_stat(); }

/* sizes of the model header this build was compiled around (403 / 10 for the shipped one; `make alt` builds others) */
int nnom_ref_in_bytes(void) { return (int)sizeof(nnom_input_data); }
int nnom_ref_out_bytes(void) { return (int)sizeof(nnom_output_data); }
