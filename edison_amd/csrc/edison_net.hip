/*
 * edison_net.hip -- C-ABI of the general int8 network path (include/edison_hip.h, "any NNoM graph"): what
 * model_run() / nnom_predict() / the layer callback of model_set_callback() give a caller of the reference
 * (nnom.c:975-1043, nnom_utils.c:258-305), batched. The loaded model (edison_model_load*) decides the shapes;
 * edison_net_get_info / edison_net_layer_info report them.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "edison_ctx.h"

extern "C" int edison_net_get_info(edison_ctx *ctx, edison_net_info *out)
{
	if (!ctx || !out) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	const ed_net_plan_t *p = &ctx->net;
	out->in_h = p->in_h; out->in_w = p->in_w; out->in_c = p->in_c;
	out->n_out = p->out_n; out->n_layers = p->n_layers; out->acts_bytes = p->acts_bytes;
	out->has_softmax = p->has_softmax; out->accelerated = ctx->fast_model ? 1 : (ctx->mm_ok ? 2 : 0);
	return EDISON_OK;
}

extern "C" int edison_net_layer_info(edison_ctx *ctx, int layer, edison_net_layer_info_t *out)
{
	if (!ctx || !out) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (layer < 0 || layer >= ctx->net.n_layers) return ed_set_err(ctx, EDISON_E_ARGUMENT, "layer index out of range");
	const ed_net_layer_t *L = &ctx->net.L[layer];
	out->type = L->type; out->out_h = L->out_h; out->out_w = L->out_w; out->out_c = L->out_c;
	out->acts_offset = L->acts_off; out->relu = L->relu;
	return EDISON_OK;
}

static int net_launch(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax, int8_t *acts,
                      int allow_fast)
{
	if (!ctx || n < 0 || (!in && n > 0)) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (n == 0) return EDISON_OK;
	if (softmax && !ctx->net.has_softmax) return ed_set_err(ctx, EDISON_E_ARGUMENT, "the loaded model has no Softmax layer; pass softmax = NULL");
	if (n >= ((int64_t)1 << 31)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "too many inputs per call");
	/* the shipped graph keeps its matrix-core kernel; per-layer dumps always come from the general kernel */
	const char *fg_ = getenv("EDISON_NET_FORCE_GENERAL"); /* A/B knob, read per call: keep the shipped graph off its specialised kernel */
	const int force_general = fg_ ? atoi(fg_) : 0;
	if (allow_fast && ctx->fast_model && !acts && !force_general)
		return ed_ctx_kws_cnn_launch(ctx, in, n, ctx->net.in_n, logits, softmax, argmax);
	int e = acts ? ed_launch_net(ctx->d_net_plan, ctx->d_net_w, ctx->d_net_seeds, ctx->net.lds_bytes, in, n, ctx->net.in_n, logits, softmax,
	                             argmax, acts, ctx->n_cu, ctx->stream)
	             : ed_ctx_net_launch(ctx, in, n, ctx->net.in_n, logits, softmax, argmax);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "network kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

extern "C" int edison_net_batch_dev(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return net_launch(ctx, in, n, logits, softmax, argmax, NULL, 1);
}

extern "C" int edison_net_layers_dev(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *acts)
{
	if (!acts && n > 0) return EDISON_E_ARGUMENT;
	return net_launch(ctx, in, n, NULL, NULL, NULL, acts, 0);
}

namespace
{
struct net_buf
{
	void *p;
	net_buf() : p(NULL) {}
	~net_buf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
} // namespace

static int net_host(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax, int8_t *acts)
{
	if (!ctx || n < 0 || (!in && n > 0)) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (n == 0) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	const size_t cnt = (size_t)n, in_n = (size_t)ctx->net.in_n, out_n = (size_t)ctx->net.out_n, acts_n = (size_t)ctx->net.acts_bytes;
	net_buf f, l, s, a, t;
	ED_HIP(ctx, f.alloc(cnt * in_n));
	if (logits) ED_HIP(ctx, l.alloc(cnt * out_n));
	if (softmax) ED_HIP(ctx, s.alloc(cnt * out_n));
	if (argmax) ED_HIP(ctx, a.alloc(cnt * sizeof(int32_t)));
	if (acts) ED_HIP(ctx, t.alloc(cnt * acts_n));
	ED_HIP(ctx, hipMemcpyAsync(f.p, in, cnt * in_n, hipMemcpyHostToDevice, ctx->stream));
	int r = net_launch(ctx, (const int8_t *)f.p, n, (int8_t *)l.p, (int8_t *)s.p, (int32_t *)a.p, (int8_t *)t.p, acts == NULL);
	if (r != EDISON_OK) return r;
	if (logits) ED_HIP(ctx, hipMemcpyAsync(logits, l.p, cnt * out_n, hipMemcpyDeviceToHost, ctx->stream));
	if (softmax) ED_HIP(ctx, hipMemcpyAsync(softmax, s.p, cnt * out_n, hipMemcpyDeviceToHost, ctx->stream));
	if (argmax) ED_HIP(ctx, hipMemcpyAsync(argmax, a.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
	if (acts) ED_HIP(ctx, hipMemcpyAsync(acts, t.p, cnt * acts_n, hipMemcpyDeviceToHost, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_net_batch(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return net_host(ctx, in, n, logits, softmax, argmax, NULL);
}

extern "C" int edison_net_layers(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *acts)
{
	if (!acts && n > 0) return EDISON_E_ARGUMENT;
	return net_host(ctx, in, n, NULL, NULL, NULL, acts);
}
