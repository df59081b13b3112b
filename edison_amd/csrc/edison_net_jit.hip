/*
 * edison_net_jit.hip -- edison_net_specialize(): the loaded graph gets its OWN matrix-core kernel, compiled at run time.
 *
 * NNoM fixes a model's shapes, buffers and per-layer kernels once, in model_compile() (nnom.c:758-900), and model_run()
 * (nnom.c:975-1040) then only walks the list. The general GPU kernel (cnn_net_mfma_kernels.hip) walks a list too -- run
 * records read by scalar loads, tile shapes and pooling windows chosen by branches, pitches and shifts in registers -- and
 * that bookkeeping is a fifth of its time (98 spilled scalars, a dependent scalar-load chain per layer). Here the same
 * source text is compiled by hipRTC with the plans' scalars and layer records in front of it as C++ constants (net_spec.c):
 * the layer loop unrolls, every choice is made by the compiler, nothing spills. Same arithmetic, bit-identical outputs
 * (tests/test_gpu_net_jit.py); kws_conv graph: 262-276 M inputs/s on the general kernel, 505-642 M on its own.
 *
 * The kernel text and the two headers it includes are compiled into the library (build.py: net_jit_sources.c). Two compilers,
 * EDISON_JIT_COMPILER=hipcc|hiprtc picks one, the default tries them in this order:
 *   hipcc   a child process ($EDISON_HIPCC, $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc, hipcc on PATH) on a temporary copy of
 *           the text: the installed ROCm's compiler, the one that built the library itself;
 *   hiprtc  in-process (dlopen). A process that already carries ANOTHER ROCm's compiler library gets that one -- a PyTorch
 *           wheel bundles libamd_comgr.so of ROCm 7.0, and the same text comes out 30 % longer with 14 spilled scalars
 *           (211 M inputs/s instead of 236 M on kws_conv) -- which is why the child process goes first.
 * Code objects are cached on disk by (graph hash, source hash, compiler): $EDISON_JIT_CACHE, else $XDG_CACHE_HOME/edison_amd,
 * else $HOME/.cache/edison_amd; EDISON_JIT_CACHE=off disables the cache. (EDISON_JIT_DEFINE=NAME=VALUE adds one -D to the hipcc
 * command line and to the cache key: A/B work on the kernel's compile-time knobs, tools/lab/ab_net_own.py.) Without a compiler, or when the compilation fails,
 * the call reports it and the graph stays on the general kernel -- the same algorithm on the same device, not other code.
 */
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char **environ;

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

/* The cache holds code objects that this process will EXECUTE: a directory of this user only. It is created 0700 (with its
 * parent under $HOME); a directory that exists already is used only if it belongs to this uid and is writable by nobody else --
 * else there is no cache (every edison_net_specialize() compiles). "" = no cache. `compiler` = which compiler made the entry AND
 * its identity (hipcc: path, size and time stamp of the binary; hipRTC: its version), so that an upgraded ROCm never meets the
 * objects of the old one (the same text gives 30 % more instructions under ROCm 7.0's hipRTC than under 7.2's hipcc). */
static void cache_path(char *out, size_t cap, uint64_t graph, uint64_t source, const char *compiler)
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
	struct stat st;
	if (stat(base, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return;
	snprintf(out, cap, "%s/net_gfx950_%016llx_%016llx_%s.hsaco", base, (unsigned long long)graph, (unsigned long long)source, compiler);
}

static char *read_fd(int fd, off_t len, size_t *n)
{
	char *buf = len > 0 && len < (off_t)(64L << 20) ? (char *)malloc((size_t)len) : NULL;
	size_t got = 0;
	while (buf && got < (size_t)len)
	{
		const ssize_t k = read(fd, buf + got, (size_t)len - got);
		if (k < 0 && errno == EINTR) continue;
		if (k <= 0) { free(buf); buf = NULL; break; }
		got += (size_t)k;
	}
	if (buf) *n = (size_t)len;
	return buf;
}

/* A cache entry is read only if it is a regular file of this uid that nobody else can write -- checked on the descriptor the bytes
 * are then read from (one open with O_NOFOLLOW, fstat, read: no window between the check and the use, no symbolic link followed). */
static char *read_cache_entry(const char *path, size_t *n)
{
	const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
	if (fd < 0) return NULL;
	struct stat st;
	char *buf = NULL;
	if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH)))
		buf = read_fd(fd, st.st_size, n);
	close(fd);
	return buf;
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

static int write_text(const char *dir, const char *name, const void *data, size_t n)
{
	char p[700];
	snprintf(p, sizeof(p), "%s/%s", dir, name);
	FILE *f = fopen(p, "wb");
	if (!f) return 0;
	const int ok = fwrite(data, 1, n, f) == n;
	return fclose(f) == 0 && ok;
}

/* the compiler this library will start: $EDISON_HIPCC, else $ROCM_PATH/bin/hipcc, else /opt/rocm/bin/hipcc -- never whatever
 * "hipcc" $PATH happens to hold (a library that runs a program by itself names it) */
static int find_hipcc(char *out, size_t cap)
{
	const char *env = getenv("EDISON_HIPCC");
	if (env && env[0]) { snprintf(out, cap, "%s", env); return access(out, X_OK) == 0; }
	if (getenv("ROCM_PATH") && getenv("ROCM_PATH")[0])
	{
		snprintf(out, cap, "%s/bin/hipcc", getenv("ROCM_PATH"));
		if (access(out, X_OK) == 0) return 1;
	}
	snprintf(out, cap, "/opt/rocm/bin/hipcc");
	return access(out, X_OK) == 0;
}

/* identity of the in-process compiler for the cache key: the HIP runtime this process runs on (libhiprtc comes with it; loading
 * the compiler library just to ask would cost more than the cache saves) */
static int hiprtc_version(void)
{
	int v = 0;
	return hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
}

/* identity of that compiler for the cache key: path, size and modification time of the binary */
static uint64_t hipcc_identity(void)
{
	char p[512];
	struct stat st;
	if (!find_hipcc(p, sizeof(p)) || stat(p, &st) != 0) return 0;
	uint64_t h = fnv(1469598103934665603ull, p, strlen(p));
	const long long v[2] = {(long long)st.st_size, (long long)st.st_mtime};
	return fnv(h, v, sizeof(v));
}

/* The installed compiler as a child process on a temporary copy of the text. 1: *code / *code_bytes hold the code object
 * (malloc'd), 0: no hipcc on this machine, -1: it ran and failed (text in ctx->err). */
static int compile_with_hipcc(edison_ctx *ctx, const char *spec, char **code, size_t *code_bytes)
{
	char hipcc[512];
	if (!find_hipcc(hipcc, sizeof(hipcc))) return 0;
	const char *tmp = getenv("TMPDIR") && getenv("TMPDIR")[0] ? getenv("TMPDIR") : "/tmp";
	char dir[600];
	snprintf(dir, sizeof(dir), "%s/edison_jit_XXXXXX", tmp);
	if (!mkdtemp(dir)) { snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: cannot create a directory under %s", tmp); return -1; }
	char src[700], out[700], log[700], inc[700];
	snprintf(src, sizeof(src), "%s/cnn_net_mfma_kernels.hip", dir);
	snprintf(out, sizeof(out), "%s/own.hsaco", dir);
	snprintf(log, sizeof(log), "%s/hipcc.log", dir);
	snprintf(inc, sizeof(inc), "-I%s", dir);
	int r = -1;
	if (write_text(dir, "cnn_net_mfma_kernels.hip", ed_jit_src_kernel, ed_jit_src_kernel_len) &&
	    write_text(dir, "edison_hip.h", ed_jit_src_edison_hip_h, ed_jit_src_edison_hip_h_len) &&
	    write_text(dir, "edison_internal.h", ed_jit_src_edison_internal_h, ed_jit_src_edison_internal_h_len) &&
	    write_text(dir, "emm_spec.h", spec, strlen(spec)))
	{
		/* device code only, a plain ELF code object (no offload bundle). -pragma-unroll-threshold: the layer loop's body holds every
		 * tile shape until it is unrolled and the layer records become constants; LLVM's default cap on a "#pragma unroll" refuses a
		 * body that size, and everything the specialisation is for hangs on that unroll (335 -> 589 M inputs/s on kws_conv) */
		/* the text the compiler gets is the product text: no knob is defined. (A LAB build of this file -- tools/lab/mkvariant.py
		 * x=@edison_net_jit.hip -- passes EDISON_JIT_DEFINE=NAME=VALUE on as one more -D behind -DED_LAB, for A/B work on the
		 * kernel's knobs; part of the cache key.) */
		char extra[128];
		const char *lab = "-DEMM_NO_LAB_DEFINE=1";
		snprintf(extra, sizeof(extra), "-DEMM_NO_EXTRA_DEFINE=1");
#ifdef ED_LAB
		const char *xd = getenv("EDISON_JIT_DEFINE");
		if (xd && xd[0]) { snprintf(extra, sizeof(extra), "-D%s", xd); lab = "-DED_LAB=1"; }
#endif
		const char *argv[] = {hipcc, "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17", "-fno-slp-vectorize",
		                      "-DEMM_JIT=1", "-DEMM_SPEC=1", "-DEMM_SPEC_HEADER=\"emm_spec.h\"", lab, extra, "-mllvm", "-pragma-unroll-threshold=1000000", "-include", "hip/hip_runtime.h", inc, "-x", "hip", "-c", src, "-o", out, NULL};
		posix_spawn_file_actions_t fa;
		posix_spawn_file_actions_init(&fa);
		posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
		posix_spawn_file_actions_addopen(&fa, 1, log, O_WRONLY | O_CREAT | O_TRUNC, 0600);
		posix_spawn_file_actions_adddup2(&fa, 1, 2);
		pid_t pid = 0;
		int status = 0;
		/* the compiler is a plain host program: it must not inherit what makes THIS process's children touch the GPU -- a
		 * profiler's LD_PRELOAD / HSA_TOOLS_LIB / ROCP* settings would load the tool (and initialise the device) in hipcc and in
		 * every program hipcc starts in turn */
		size_t n_env = 0;
		while (environ && environ[n_env]) n_env++;
		char **envp = (char **)calloc(n_env + 1, sizeof(char *));
		size_t kept = 0;
		static const char *drop[] = {"LD_PRELOAD=", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP", "ROCTRACER", "ROCPROFILER", "HIP_TOOLS", "OMPT_"};
		for (size_t k = 0; envp && k < n_env; k++)
		{
			int skip = 0;
			for (size_t d = 0; d < sizeof(drop) / sizeof(drop[0]); d++) skip |= strncmp(environ[k], drop[d], strlen(drop[d])) == 0;
			if (!skip) envp[kept++] = environ[k];
		}
		/* its own process group: hipcc is a driver that starts clang / lld in turn, and a compiler that has to be stopped is stopped
		 * with everything it started */
		posix_spawnattr_t at;
		posix_spawnattr_init(&at);
		posix_spawnattr_setflags(&at, POSIX_SPAWN_SETPGROUP);
		posix_spawnattr_setpgroup(&at, 0);
		const int sp = envp ? posix_spawn(&pid, hipcc, &fa, &at, (char *const *)argv, envp) : ENOMEM;
		posix_spawnattr_destroy(&at);
		free(envp);
		posix_spawn_file_actions_destroy(&fa);
		if (sp != 0) snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: cannot start %s: %s", hipcc, strerror(sp));
		else
		{
			/* a compiler that does not come back must not hang a model load: two minutes (it needs ~1 s), then it is stopped by PID */
			int done = 0, confirmed = 0; /* confirmed: waitpid handed us the child's own exit status */
			for (int waited_ms = 0; !done; waited_ms += 5)
			{
				const pid_t w = waitpid(pid, &status, WNOHANG);
				if (w == pid) { done = 1; confirmed = 1; }
				else if (w < 0 && errno != EINTR)
				{
					/* ECHILD: the host program ignores SIGCHLD or reaps children itself -- the child has ALREADY been reaped (that is
					 * what the error says), so its pid and process group may belong to somebody else by now: signal nothing. Nobody
					 * can tell us how the compiler ended, so nothing is taken from it either (confirmed stays 0). */
					done = 1;
				}
				else if (waited_ms > 120000)
				{
					(void)kill(-pid, SIGKILL);
					while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
					done = 1;
				}
				else usleep(5000);
			}
			if (!confirmed) status = -1;
			*code = confirmed && WIFEXITED(status) && WEXITSTATUS(status) == 0 ? read_file(out, code_bytes) : NULL;
			if (*code) r = 1;
			else
			{
				size_t ln = 0;
				char *text = read_file(log, &ln);
				snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: %s failed (status %d): %.*s", hipcc, status, (int)(ln < 300 ? ln : 300), text ? text : "");
				free(text);
			}
		}
	}
	else snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: cannot write the kernel text under %s", dir);
	static const char *files[] = {"cnn_net_mfma_kernels.hip", "edison_hip.h", "edison_internal.h", "emm_spec.h", "own.hsaco", "hipcc.log"};
	for (size_t k = 0; k < sizeof(files) / sizeof(files[0]); k++)
	{
		char f[700];
		snprintf(f, sizeof(f), "%s/%s", dir, files[k]);
		(void)remove(f);
	}
	(void)rmdir(dir);
	return r;
}

/* hipRTC in this process. 1 / 0 (no libhiprtc.so) / -1 as above. */
static int compile_with_hiprtc(edison_ctx *ctx, const char *spec, char **code, size_t *code_bytes)
{
	rtc_api rtc;
	if (!rtc_load(&rtc)) return 0;
	const char *headers[] = {k_stdint_h, k_stddef_h, (const char *)ed_jit_src_edison_hip_h, (const char *)ed_jit_src_edison_internal_h, spec};
	const char *names[] = {"stdint.h", "stddef.h", "edison_hip.h", "edison_internal.h", "emm_spec.h"};
	const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DEMM_JIT=1", "-DEMM_SPEC=1", "-DEMM_SPEC_HEADER=\"emm_spec.h\"",
	                      "-mllvm", "-pragma-unroll-threshold=1000000"};
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
		return -1;
	}
	*code = NULL;
	if (rtc.GetCodeSize(prog, code_bytes) == 0 && *code_bytes > 0 && (*code = (char *)malloc(*code_bytes)) != NULL && rtc.GetCode(prog, *code) != 0) { free(*code); *code = NULL; }
	(void)rtc.DestroyProgram(&prog);
	if (!*code) { (void)ed_set_err(ctx, EDISON_E_RUNTIME, "edison_net_specialize: hipRTC returned no code object"); return -1; }
	return 1;
}

void ed_ctx_net_spec_drop(edison_ctx *ctx)
{
	if (ctx->spec_mod) (void)hipModuleUnload((hipModule_t)ctx->spec_mod);
	ctx->spec_mod = NULL;
	ctx->spec_fn = NULL;
	ctx->spec_state = 0;
}

/* cache_only: take the code object from the on-disk cache or leave the graph on the general kernel (EDISON_E_NO_IMPL), never
 * start a compiler -- what a model load does by itself */
static int specialize(edison_ctx *ctx, int cache_only)
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
	source = fnv(source, "pragma-unroll-threshold=1000000", 31); /* the compiler options are part of what a cache entry was made from */
#ifdef ED_LAB
	if (getenv("EDISON_JIT_DEFINE") && getenv("EDISON_JIT_DEFINE")[0]) source = fnv(source, getenv("EDISON_JIT_DEFINE"), strlen(getenv("EDISON_JIT_DEFINE")));
#endif

	const char *want = getenv("EDISON_JIT_COMPILER");
	const int try_hipcc = !want || !want[0] || !strcmp(want, "hipcc"), try_rtc = !want || !want[0] || !strcmp(want, "hiprtc");
	if (!try_hipcc && !try_rtc) { free(spec); return ed_set_err(ctx, EDISON_E_ARGUMENT, "EDISON_JIT_COMPILER: hipcc or hiprtc"); }
	char path[600];
	size_t code_bytes = 0;
	char *code = NULL;
	int state = 0; /* 1: hipcc, 2: cache, 3: hipRTC */
	/* the cache: an entry made by the compiler that would be tried first, then the other's */
	char id_hipcc[40], id_rtc[40];
	snprintf(id_hipcc, sizeof(id_hipcc), "hipcc-%016llx", (unsigned long long)hipcc_identity());
	snprintf(id_rtc, sizeof(id_rtc), "hiprtc-%d", hiprtc_version());
	for (int k = 0; k < 2 && !code; k++)
	{
		const int is_hipcc = k == 0;
		if (is_hipcc ? !try_hipcc : !try_rtc) continue;
		cache_path(path, sizeof(path), graph, source, is_hipcc ? id_hipcc : id_rtc);
		if (path[0] && (code = read_cache_entry(path, &code_bytes)) != NULL) state = 2;
	}
	if (!code && cache_only)
	{
		free(spec);
		return ed_set_err(ctx, EDISON_E_NO_IMPL, "no cached code object for this graph");
	}
	if (!code)
	{
		int r = try_hipcc ? compile_with_hipcc(ctx, spec, &code, &code_bytes) : 0;
		if (r == 1) state = 1;
		if (r != 1 && try_rtc)
		{
			/* no hipcc here, or it ran and failed (a read-only TMPDIR, a broken installation): the in-process compiler is the second
			 * chance; if that fails too, the FIRST compiler's message is the one worth keeping */
			char first[sizeof(ctx->err)];
			memcpy(first, ctx->err, sizeof(first));
			const int r1 = r;
			r = compile_with_hiprtc(ctx, spec, &code, &code_bytes);
			if (r == 1) state = 3;
			else if (r1 == -1) { memcpy(ctx->err, first, sizeof(first)); r = -1; }
		}
		if (r != 1)
		{
			free(spec);
			if (r == 0) return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_net_specialize: neither hipcc nor libhiprtc.so found (the graph stays on the general kernel)");
			return EDISON_E_RUNTIME;
		}
		/* (several processes that load the same new graph at once each compile it: the entry is written under a name of its own
		 * and renamed into place, the last one wins, all are the same) */
		cache_path(path, sizeof(path), graph, source, state == 1 ? id_hipcc : id_rtc);
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
		if (state == 2 && path[0]) (void)remove(path); /* a damaged cache entry: the next call compiles again */
		snprintf(ctx->err, sizeof(ctx->err), "edison_net_specialize: loading the code object failed: %s", hipGetErrorString(e));
		(void)hipGetLastError();
		return EDISON_E_RUNTIME;
	}
	/* more than 64 KB of dynamic LDS has to be asked for (a module function takes the same call) */
	if (ctx->mm_lds > 64 * 1024 && hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->mm_lds) != hipSuccess) (void)hipGetLastError();
	ctx->spec_mod = (void *)mod;
	ctx->spec_fn = (void *)fn;
	ctx->spec_epoch = ctx->model_epoch;
	ctx->spec_state = state;
	return EDISON_OK;
}

extern "C" int edison_net_specialize(edison_ctx *ctx) { return specialize(ctx, 0); }

/* a model load: the graph's own kernel if an earlier edison_net_specialize() left it in the cache; errors are not the load's */
void ed_ctx_net_spec_from_cache(edison_ctx *ctx)
{
	char keep[sizeof(ctx->err)];
	memcpy(keep, ctx->err, sizeof(keep));
	if (specialize(ctx, 1) != EDISON_OK) memcpy(ctx->err, keep, sizeof(keep));
}

/* 0: the loaded graph runs on the general kernel; on its own kernel: 1 compiled just now by the hipcc child process, 2 taken
 * from the cache, 3 compiled just now by hipRTC in this process */
extern "C" int edison_net_specialized(edison_ctx *ctx)
{
	return ctx && ctx->spec_fn && ctx->spec_epoch == ctx->model_epoch ? ctx->spec_state : 0;
}

/* the launch of the graph's own kernel: grid and LDS exactly as ed_launch_net_mfma's */
int ed_ctx_net_spec_launch(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax,
                           int32_t *argmax, unsigned *done_flag, unsigned done_seq, int *flag_written)
{
	if (flag_written) *flag_written = 0;
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
	unsigned *flag = (done_flag && n <= ctx->mm_batch) ? done_flag : NULL; /* one workgroup whose first wave takes every input */
	void *kargs[] = {(void *)&dev_plan, (void *)&dev_mm, (void *)&dev_frag, (void *)&dev_seeds, (void *)&in, (void *)&n, (void *)&in_stride,
	                 (void *)&logits, (void *)&softmax, (void *)&argmax, (void *)&flag, (void *)&done_seq};
	const hipError_t e = hipModuleLaunchKernel((hipFunction_t)ctx->spec_fn, (unsigned)blocks, 1, 1, (unsigned)(64 * waves), 1, 1, (unsigned)ctx->mm_lds, stream, kargs, NULL);
	if (e == hipSuccess && flag && flag_written) *flag_written = 1;
	return (int)e;
}
