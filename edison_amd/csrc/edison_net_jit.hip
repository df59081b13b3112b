/*
 * edison_net_jit.hip -- edison_net_specialize(): the loaded graph gets its OWN matrix-core kernel, compiled at run time.
 *
 * NNoM fixes a model's shapes, buffers and per-layer kernels once, in model_compile() (nnom.c:758-900), and model_run()
 * (nnom.c:975-1040) then only walks the list. The general GPU kernel (cnn_net_mfma_kernels.hip) walks a list too -- run
 * records read by scalar loads, tile shapes and pooling windows chosen by branches, pitches and shifts in registers -- and
 * that bookkeeping is a fifth of its time (98 spilled scalars, a dependent scalar-load chain per layer). Here the same
 * source text is compiled by hipRTC with the plans' scalars and layer records in front of it as C++ constants (net_spec.c):
 * the layer loop unrolls, every choice is made by the compiler, nothing spills. Same arithmetic, bit-identical outputs
 * (tests/test_gpu_net_jit.py); kws_conv graph: 198 -> 236 M inputs/s on one box.
 *
 * The kernel text and the two headers it includes are compiled into the library (build.py: net_jit_sources.c), hipRTC is
 * found with dlopen at the first call, code objects are cached on disk by (graph hash, source hash):
 * $EDISON_JIT_CACHE, else $XDG_CACHE_HOME/edison_amd, else $HOME/.cache/edison_amd; EDISON_JIT_CACHE=off disables the cache.
 * Without hipRTC, or when the compilation fails, the call reports it and the graph stays on the general kernel -- which is
 * the same algorithm on the same device, not a fallback to other code.
 */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "edison_ctx.h"

extern "C" const unsigned char ed_jit_src_kernel[], ed_jit_src_edison_hip_h[], ed_jit_src_edison_internal_h[];
extern "C" const size_t ed_jit_src_kernel_len, ed_jit_src_edison_hip_h_len, ed_jit_src_edison_internal_h_len;

/* what <stdint.h> / <stddef.h> give the two headers: hipRTC has no system include path */
static const char k_stdint_h[] =
	"#pragma once\n"
	"typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
	"typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n"
	"typedef unsigned long uintptr_t;\n";
static const char k_stddef_h[] = "#pragma once\n";

typedef struct _hiprtcProgram *rtc_program;
struct rtc_api
{
	void *lib;
	int (*CreateProgram)(rtc_program *, const char *, const char *, int, const char *const *, const char *const *);
	int (*CompileProgram)(rtc_program, int, const char *const *);
	int (*GetProgramLogSize)(rtc_program, size_t *);
	int (*GetProgramLog)(rtc_program, char *);
	int (*GetCodeSize)(rtc_program, size_t *);
	int (*GetCode)(rtc_program, char *);
	int (*DestroyProgram)(rtc_program *);
};

static int rtc_load(rtc_api *a)
{
	static const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
	memset(a, 0, sizeof(*a));
	for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !a->lib; i++) a->lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
	if (!a->lib) return 0;
#define SYM(f) (*(void **)&a->f = dlsym(a->lib, "hiprtc" #f))
	SYM(CreateProgram); SYM(CompileProgram); SYM(GetProgramLogSize); SYM(GetProgramLog); SYM(GetCodeSize); SYM(GetCode); SYM(DestroyProgram);
#undef SYM
	return a->CreateProgram && a->CompileProgram && a->GetProgramLogSize && a->GetProgramLog && a->GetCodeSize && a->GetCode && a->DestroyProgram;
}

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
	const unsigned char *b = (const unsigned char *)p;
	for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}

/* "" = no cache */
static void cache_path(char *out, size_t cap, uint64_t graph, uint64_t source)
{
	out[0] = 0;
	const char *dir = getenv("EDISON_JIT_CACHE");
	char base[400];
	if (dir && (!strcmp(dir, "off") || !strcmp(dir, "0"))) return;
	if (dir && dir[0]) snprintf(base, sizeof(base), "%s", dir);
	else if (getenv("XDG_CACHE_HOME") && getenv("XDG_CACHE_HOME")[0]) snprintf(base, sizeof(base), "%s/edison_amd", getenv("XDG_CACHE_HOME"));
	else if (getenv("HOME") && getenv("HOME")[0])
	{
		snprintf(base, sizeof(base), "%s/.cache", getenv("HOME"));
		(void)mkdir(base, 0700);
		snprintf(base, sizeof(base), "%s/.cache/edison_amd", getenv("HOME"));
	}
	else return;
	(void)mkdir(base, 0700);
	snprintf(out, cap, "%s/net_gfx950_%016llx_%016llx.hsaco", base, (unsigned long long)graph, (unsigned long long)source);
}

static char *read_file(const char *path, size_t *n)
{
	FILE *f = fopen(path, "rb");
	if (!f) return NULL;
	fseek(f, 0, SEEK_END);
	const long len = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = len > 0 && len < (64L << 20) ? (char *)malloc((size_t)len) : NULL;
	if (buf && fread(buf, 1, (size_t)len, f) != (size_t)len) { free(buf); buf = NULL; }
	fclose(f);
	if (buf) *n = (size_t)len;
	return buf;
}

static void write_file_atomic(const char *path, const char *data, size_t n)
{
	char tmp[600];
	snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", path, (long)getpid());
	FILE *f = fopen(tmp, "wb");
	if (!f) return;
	const int ok = fwrite(data, 1, n, f) == n;
	if (fclose(f) != 0 || !ok || rename(tmp, path) != 0) (void)remove(tmp);
}

void ed_ctx_net_spec_drop(edison_ctx *ctx)
{
	if (ctx->spec_mod) (void)hipModuleUnload((hipModule_t)ctx->spec_mod);
	ctx->spec_mod = NULL;
	ctx->spec_fn = NULL;
	ctx->spec_state = 0;
}

extern "C" int edison_net_specialize(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_net_specialize: no model loaded");
	if (!ctx->mm_ok || !ctx->h_mm_plan)
		return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_net_specialize: this graph has no matrix-core plan (it runs on the layer-by-layer kernel)");
	if (ctx->spec_fn && ctx->spec_epoch == ctx->model_epoch) return EDISON_OK; /* already done for this load */
	ED_HIP(ctx, hipSetDevice(ctx->device));
	ED_HIP(ctx, hipDeviceSynchronize()); /* launches of the previous load's kernel may still be in flight */
	ed_ctx_net_spec_drop(ctx);

	/* the specialisation header */
	const size_t need = ed_emit_net_spec(&ctx->net, ctx->h_mm_plan, NULL, 0) + 1;
	char *spec = (char *)malloc(need);
	if (!spec) return ed_set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
	(void)ed_emit_net_spec(&ctx->net, ctx->h_mm_plan, spec, need);
	const uint64_t graph = ed_net_spec_hash(&ctx->net, ctx->h_mm_plan);
	uint64_t source = fnv(1469598103934665603ull, ed_jit_src_kernel, ed_jit_src_kernel_len);
	source = fnv(source, ed_jit_src_edison_hip_h, ed_jit_src_edison_hip_h_len);
	source = fnv(source, ed_jit_src_edison_internal_h, ed_jit_src_edison_internal_h_len);
	source = fnv(source, k_stdint_h, sizeof(k_stdint_h));

	char path[600];
	cache_path(path, sizeof(path), graph, source);
	size_t code_bytes = 0;
	char *code = path[0] ? read_file(path, &code_bytes) : NULL;
	int from_cache = code != NULL;
	if (!code)
	{
		rtc_api rtc;
		if (!rtc_load(&rtc))
		{
			free(spec);
			return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_net_specialize: libhiprtc.so not found (the graph stays on the general kernel)");
		}
		const char *headers[] = {k_stdint_h, k_stddef_h, (const char *)ed_jit_src_edison_hip_h, (const char *)ed_jit_src_edison_internal_h, spec};
		const char *names[] = {"stdint.h", "stddef.h", "edison_hip.h", "edison_internal.h", "emm_spec.h"};
		const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DEMM_JIT=1", "-DEMM_SPEC=1", "-DEMM_SPEC_HEADER=\"emm_spec.h\""};
		rtc_program prog = NULL;
		int r = rtc.CreateProgram(&prog, (const char *)ed_jit_src_kernel, "cnn_net_mfma_kernels.hip", 5, headers, names);
		if (r == 0) r = rtc.CompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
		if (r != 0)
		{
			size_t ln = 0;
			char *log = NULL;
			if (prog && rtc.GetProgramLogSize(prog, &ln) == 0 && ln > 1 && (log = (char *)malloc(ln + 1)) != NULL && rtc.GetProgramLog(prog, log) == 0) log[ln] = 0;
			snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: hipRTC error %d: %.400s", r, log ? log : "(no log)");
			free(log);
			if (prog) (void)rtc.DestroyProgram(&prog);
			free(spec);
			return EDISON_E_RUNTIME;
		}
		if (rtc.GetCodeSize(prog, &code_bytes) == 0 && code_bytes > 0 && (code = (char *)malloc(code_bytes)) != NULL && rtc.GetCode(prog, code) != 0) { free(code); code = NULL; }
		(void)rtc.DestroyProgram(&prog);
		if (!code) { free(spec); return ed_set_err(ctx, EDISON_E_RUNTIME, "edison_net_specialize: hipRTC returned no code object"); }
		if (path[0]) write_file_atomic(path, code, code_bytes);
	}
	free(spec);

	hipModule_t mod = NULL;
	hipFunction_t fn = NULL;
	hipError_t e = hipModuleLoadData(&mod, code);
	if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, "ed_net_mfma_spec");
	free(code);
	if (e != hipSuccess)
	{
		if (mod) (void)hipModuleUnload(mod);
		if (from_cache && path[0]) (void)remove(path); /* a damaged cache entry: the next call compiles again */
		snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: loading the code object failed: %s", hipGetErrorString(e));
		(void)hipGetLastError();
		return EDISON_E_RUNTIME;
	}
	/* more than 64 KB of dynamic LDS has to be asked for (a module function takes the same call) */
	if (ctx->mm_lds > 64 * 1024 && hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->mm_lds) != hipSuccess) (void)hipGetLastError();
	ctx->spec_mod = (void *)mod;
	ctx->spec_fn = (void *)fn;
	ctx->spec_epoch = ctx->model_epoch;
	ctx->spec_state = from_cache ? 2 : 1;
	return EDISON_OK;
}

/* 0: the loaded graph runs on the general kernel, 1: on its own kernel compiled by this process, 2: ... loaded from the cache */
extern "C" int edison_net_specialized(edison_ctx *ctx)
{
	return ctx && ctx->spec_fn && ctx->spec_epoch == ctx->model_epoch ? ctx->spec_state : 0;
}

/* the launch of the graph's own kernel: grid and LDS exactly as ed_launch_net_mfma's */
int ed_ctx_net_spec_launch(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax,
                           int32_t *argmax)
{
	if (n <= 0) return 0;
	const int waves = ctx->mm_waves;
	int per_cu = (160 * 1024) / (ctx->mm_lds + 256);
	if (per_cu > 32 / waves) per_cu = 32 / waves;
	if (per_cu < 1) per_cu = 1;
	const int64_t per_block = (int64_t)ctx->mm_batch * waves;
	int64_t blocks = (n + per_block - 1) / per_block;
	if (blocks > (int64_t)ctx->n_cu * per_cu) blocks = (int64_t)ctx->n_cu * per_cu;
	const ed_net_plan_t *dev_plan = ctx->d_net_plan;
	const ed_mm_plan_t *dev_mm = ctx->d_mm_plan;
	const int8_t *dev_frag = ctx->d_mm_frag;
	const int32_t *dev_seeds = ctx->d_mm_seeds;
	void *kargs[] = {(void *)&dev_plan, (void *)&dev_mm, (void *)&dev_frag, (void *)&dev_seeds, (void *)&in, (void *)&n, (void *)&in_stride,
	                 (void *)&logits, (void *)&softmax, (void *)&argmax};
	return (int)hipModuleLaunchKernel((hipFunction_t)ctx->spec_fn, (unsigned)blocks, 1, 1, (unsigned)(64 * waves), 1, 1, (unsigned)ctx->mm_lds, stream, kargs, NULL);
}
