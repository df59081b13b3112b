/*
 * edison_dist.hip -- the multi-GPU step of the batched scoring path behind the C-ABI: one process per GPU, utterances
 * sharded contiguously (no data-path collective: frames and utterances are independent), and ONE RCCL all-gather of the
 * per-class int8 logits over xGMI (10 B per utterance). The reference has no counterpart (its only transport is a
 * 115200-baud UART, firmware/src/hostinterface.c:96-112); the entry points exist so that
 * a C host built around aiRunInference-style code reaches the 8-GPU configuration without Python or torch.
 *
 * RCCL is bound at run time (dlopen), not at link time: libedison_hip.so then loads on machines without RCCL, and in a
 * process that already holds an RCCL (PyTorch bundles one with the soname librccl.so.1) the same instance is reused
 * instead of a second copy being mapped.
 */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "edison_ctx.h"

typedef struct { char internal[EDISON_DIST_ID_BYTES]; } ed_nccl_id; /* = ncclUniqueId (rccl.h:43) */
typedef void *ed_nccl_comm;
typedef int (*fn_get_unique_id)(ed_nccl_id *);
typedef int (*fn_comm_init_rank)(ed_nccl_comm *, int, ed_nccl_id, int);
typedef int (*fn_comm_destroy)(ed_nccl_comm);
typedef int (*fn_all_gather)(const void *, void *, size_t, int /* ncclDataType_t */, ed_nccl_comm, hipStream_t);
typedef const char *(*fn_error_string)(int);

static struct
{
	void *lib;
	fn_get_unique_id get_unique_id;
	fn_comm_init_rank comm_init_rank;
	fn_comm_destroy comm_destroy;
	fn_all_gather all_gather;
	fn_error_string error_string;
	char err[256];
} g_rccl;

static int rccl_bind(void)
{
	if (g_rccl.lib) return EDISON_OK;
	const char *env = getenv("EDISON_RCCL_LIB");
	const char *cand[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *h = NULL;
	/* an RCCL that is already mapped into the process (PyTorch's) wins: one RCCL per process */
	for (int i = 1; i < 3 && !h; i++) h = dlopen(cand[i], RTLD_NOW | RTLD_NOLOAD);
	for (int i = 0; i < 4 && !h; i++)
		if (cand[i]) h = dlopen(cand[i], RTLD_NOW | RTLD_LOCAL);
	if (!h)
	{
		snprintf(g_rccl.err, sizeof(g_rccl.err), "RCCL not found (librccl.so.1; set EDISON_RCCL_LIB): %s", dlerror());
		return EDISON_E_NO_IMPL;
	}
	g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
	g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
	g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
	g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
	g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
	if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_gather)
	{
		snprintf(g_rccl.err, sizeof(g_rccl.err), "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather");
		dlclose(h);
		return EDISON_E_NO_IMPL;
	}
	g_rccl.lib = h;
	return EDISON_OK;
}

static int rccl_fail(edison_ctx *ctx, const char *what, int rc)
{
	if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
	return EDISON_E_RUNTIME;
}

/* Can this process reach RCCL at all? Binds the library and nothing else: no bootstrap thread, no socket (a probe with
 * edison_dist_unique_id would leave one of each behind on every rank that asks). */
extern "C" int edison_dist_available(void)
{
	return rccl_bind();
}

extern "C" int edison_dist_unique_id(void *id128)
{
	if (!id128) return EDISON_E_ARGUMENT;
	int r = rccl_bind();
	if (r != EDISON_OK) return r;
	ed_nccl_id id;
	int rc = g_rccl.get_unique_id(&id);
	if (rc != 0) return EDISON_E_RUNTIME;
	memcpy(id128, &id, sizeof(id));
	return EDISON_OK;
}

extern "C" int edison_dist_init(edison_ctx *ctx, const void *id128, int rank, int world_size)
{
	if (!ctx || !id128 || world_size < 1 || rank < 0 || rank >= world_size) return EDISON_E_ARGUMENT;
	if (ctx->dist_comm) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_dist_init: this context already belongs to a communicator");
	int r = rccl_bind();
	if (r != EDISON_OK) return ed_set_err(ctx, r, g_rccl.err);
	ED_HIP(ctx, hipSetDevice(ctx->device));
	ed_nccl_id id;
	memcpy(&id, id128, sizeof(id));
	ed_nccl_comm comm = NULL;
	int rc = g_rccl.comm_init_rank(&comm, world_size, id, rank);
	if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
	ctx->dist_comm = comm;
	ctx->dist_rank = rank;
	ctx->dist_world = world_size;
	return EDISON_OK;
}

extern "C" int edison_dist_info(const edison_ctx *ctx, int *rank, int *world_size)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (rank) *rank = ctx->dist_comm ? ctx->dist_rank : 0;
	if (world_size) *world_size = ctx->dist_comm ? ctx->dist_world : 1;
	return EDISON_OK;
}

extern "C" int edison_dist_shutdown(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (ctx->dist_comm)
	{
		(void)hipStreamSynchronize(ctx->stream);
		(void)g_rccl.comm_destroy((ed_nccl_comm)ctx->dist_comm);
		ctx->dist_comm = NULL;
	}
	if (ctx->dist_scratch)
	{
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipFree(ctx->dist_scratch);
		ctx->dist_scratch = NULL;
		ctx->dist_scratch_bytes = 0;
	}
	return EDISON_OK;
}

/* all ranks: local [n_local_utt][10] int8 -> all [world * n_local_utt][10], rank r's rows at r * n_local_utt (every rank
 * passes the same n_local_utt: equal shards, the shape ncclAllGather has). Enqueued on the context's stream. */
extern "C" int edison_dist_allgather_logits(edison_ctx *ctx, const int8_t *local_logits, int64_t n_local_utt, int8_t *all_logits)
{
	if (!ctx || n_local_utt < 0 || (n_local_utt > 0 && (!local_logits || !all_logits))) return EDISON_E_ARGUMENT;
	if (n_local_utt == 0) return EDISON_OK;
	const size_t bytes = (size_t)n_local_utt * EDISON_NET_OUT;
	if (!ctx->dist_comm)
	{
		/* a context outside any communicator is a world of one: the gather is a copy */
		if (all_logits != local_logits) ED_HIP(ctx, hipMemcpyAsync(all_logits, local_logits, bytes, hipMemcpyDeviceToDevice, ctx->stream));
		return EDISON_OK;
	}
	int rc = g_rccl.all_gather(local_logits, all_logits, bytes, 0 /* ncclInt8 */, (ed_nccl_comm)ctx->dist_comm, ctx->stream);
	if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
	return EDISON_OK;
}

static void shard_of(int64_t n_items, int rank, int world, int64_t *lo, int64_t *hi)
{
	const int64_t base = n_items / world, rem = n_items % world;
	*lo = rank * base + (rank < rem ? rank : rem);
	*hi = *lo + base + (rank < rem ? 1 : 0);
}

/* The same gather for a batch that the world size does not divide: every rank passes the TOTAL number of utterances, its
 * own shard is edison_dist_shard_range(n_total_utt, rank, world) (the first n_total % world ranks hold one utterance
 * more), and all_logits receives the n_total_utt rows in rank order without gaps. Still one ncclAllGather: the shards are
 * padded to the largest one in a scratch block and the W received blocks are compacted by device copies on the same
 * stream. Equal shards take the direct path. */
extern "C" int edison_dist_allgather_logits_total(edison_ctx *ctx, const int8_t *local_logits, int64_t n_total_utt, int8_t *all_logits)
{
	if (!ctx || n_total_utt < 0) return EDISON_E_ARGUMENT;
	const int world = ctx->dist_comm ? ctx->dist_world : 1, rank = ctx->dist_comm ? ctx->dist_rank : 0;
	int64_t lo, hi;
	shard_of(n_total_utt, rank, world, &lo, &hi);
	if (n_total_utt == 0) return EDISON_OK;
	if (!all_logits || (hi > lo && !local_logits)) return EDISON_E_ARGUMENT;
	/* EDISON_DIST_FORCE_PADDED=1: take the padded road even where the shards are equal (the tests' way to run the pad /
	 * gather / compact sequence through a real communicator on a box with one GPU) */
	const char *force = getenv("EDISON_DIST_FORCE_PADDED");
	if ((n_total_utt % world == 0 && !(force && force[0] == '1')) || !ctx->dist_comm)
		return edison_dist_allgather_logits(ctx, local_logits, hi - lo, all_logits);
	const int64_t m = n_total_utt / world + 1;                      /* the largest shard */
	const size_t blk = (size_t)m * EDISON_NET_OUT, need = blk * (size_t)(world + 1);
	if (need > ctx->dist_scratch_bytes)
	{
		ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->dist_scratch) ED_HIP(ctx, hipFree(ctx->dist_scratch));
		ctx->dist_scratch = NULL;
		ctx->dist_scratch_bytes = 0;
		hipError_t e = hipMalloc(&ctx->dist_scratch, need);
		if (e == hipErrorOutOfMemory) return ed_set_err(ctx, EDISON_E_NO_MEMORY, "edison_dist_allgather_logits_total: out of HBM");
		ED_HIP(ctx, e);
		ctx->dist_scratch_bytes = need;
	}
	int8_t *send = (int8_t *)ctx->dist_scratch, *recv = send + blk;
	ED_HIP(ctx, hipMemsetAsync(send, 0, blk, ctx->stream));
	if (hi > lo) ED_HIP(ctx, hipMemcpyAsync(send, local_logits, (size_t)(hi - lo) * EDISON_NET_OUT, hipMemcpyDeviceToDevice, ctx->stream));
	int rc = g_rccl.all_gather(send, recv, blk, 0 /* ncclInt8 */, (ed_nccl_comm)ctx->dist_comm, ctx->stream);
	if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
	for (int r = 0; r < world; r++)
	{
		int64_t a, b;
		shard_of(n_total_utt, r, world, &a, &b);
		if (b > a)
			ED_HIP(ctx, hipMemcpyAsync(all_logits + (size_t)a * EDISON_NET_OUT, recv + (size_t)r * blk, (size_t)(b - a) * EDISON_NET_OUT,
			                           hipMemcpyDeviceToDevice, ctx->stream));
	}
	return EDISON_OK;
}

/* This rank's shard of the batched scoring path in one call: MFCC (variant B) -> int8 features -> CNN on n_local_utt
 * utterances, then the all-gather of the logits. softmax / argmax stay local (they follow from the logits). */
extern "C" int edison_kws_batch_sharded_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_local_utt, int64_t utt_stride,
                                            int8_t *feat, int8_t *logits_local, int8_t *softmax, int32_t *argmax,
                                            int8_t *logits_all)
{
	if (!ctx || !logits_local || !logits_all) return EDISON_E_ARGUMENT;
	int r = edison_kws_batch_dev(ctx, audio, n_local_utt, utt_stride, feat, logits_local, softmax, argmax);
	if (r != EDISON_OK) return r;
	return edison_dist_allgather_logits(ctx, logits_local, n_local_utt, logits_all);
}

/* The same for a batch of n_total_utt utterances that the world size need not divide: `audio` holds THIS rank's shard
 * (edison_dist_shard_range(n_total_utt, rank, world) utterances), logits_all receives all n_total_utt rows. */
extern "C" int edison_kws_batch_sharded_total_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_total_utt, int64_t utt_stride,
                                                  int8_t *feat, int8_t *logits_local, int8_t *softmax, int32_t *argmax,
                                                  int8_t *logits_all)
{
	if (!ctx || !logits_all || n_total_utt < 0) return EDISON_E_ARGUMENT;
	int64_t lo, hi;
	shard_of(n_total_utt, ctx->dist_comm ? ctx->dist_rank : 0, ctx->dist_comm ? ctx->dist_world : 1, &lo, &hi);
	if (hi > lo)
	{
		if (!logits_local) return EDISON_E_ARGUMENT;
		int r = edison_kws_batch_dev(ctx, audio, hi - lo, utt_stride, feat, logits_local, softmax, argmax);
		if (r != EDISON_OK) return r;
	}
	return edison_dist_allgather_logits_total(ctx, logits_local, n_total_utt, logits_all);
}

/* Contiguous shard [lo, hi) of `rank` out of n_items: the first n_items % world_size ranks take one item more. */
extern "C" int edison_dist_shard_range(int64_t n_items, int rank, int world_size, int64_t *lo, int64_t *hi)
{
	if (n_items < 0 || world_size < 1 || rank < 0 || rank >= world_size || !lo || !hi) return EDISON_E_ARGUMENT;
	shard_of(n_items, rank, world_size, lo, hi);
	return EDISON_OK;
}
