/*
 * cnn_net_mfma_kernels.hip -- ANY sequential NNoM int8 graph the planner accepts, with every Conv2D / Dense layer on the
 * gfx950 matrix cores (v_mfma_i32_32x32x32_i8). The GPU's model_run() (nnom.c:975-1040) for batch scoring; the
 * layer-by-layer kernel of cnn_net_kernels.hip stays for per-layer dumps and for graphs whose plan does not fit here.
 *
 * Arithmetic: exactly the reference's -- out = sat8((sum x*w + (bias << BL) + NN_ROUND(RS)) >> RS), ReLU as a tail
 * activation, max-pool over the part of the window inside the image, arm_softmax_q7's portable branch, first-maximum
 * argmax (citations in cnn_net_kernels.hip); integer sums are exact in any order, so the results are bit-identical.
 *
 * Scheme (ed_mm_plan_t, model_net_mm.c): implicit GEMM D[out_channel][pixel] with k = (kernel row, 16-byte chunk of the
 * row's contiguous kw * C_in input bytes). The consumer layer dictates how its input lies in LDS: zero-padded so that no
 * tap test is needed; the producing layer's epilogue writes straight into that layout. Layers whose C_in is not a
 * multiple of 16 read from an expanded copy with one aligned record per (input row, output x). A WAVEFRONT takes `batch`
 * inputs through the whole layer list by itself in its own slice of LDS -- no workgroup barrier in the loop (a wave's DS
 * instructions are serviced in order); the first version ran the workgroup in lockstep phases and spent a third of its
 * time in barriers. The weight fragments stay in LDS for the whole launch, shared by the waves, when they fit beside the
 * activation slices (mode 2), else they stream from L2 (0).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

/* diagnostic build only (-DEMM_STAMP=1, tools/lab): workgroup-level cycle stamps per phase into a debug buffer */
#ifndef EMM_STAMP
#define EMM_STAMP 0
#endif
#if EMM_STAMP
__device__ unsigned long long *g_emm_dbg = nullptr;
extern "C" void ed_set_net_debug_buffer(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_emm_dbg), &p, sizeof(p)); }
#define EMM_ST(i) { unsigned long long n_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_) :: "memory"); stamp_[i] += n_ - tl_; tl_ = n_; }
#else
#define EMM_ST(i)
#endif
#define EMM_MAX_THREADS 768 /* 12 waves: 168 VGPRs each (four accumulator tiles + the pipeline's operands need ~150) */

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

/* Everything a wave touches in its layer loop lives in LDS, and the pointers say so: a generic pointer makes the
 * compiler emit flat_load / flat_store (64-bit addresses, both wait counters, a full drain before every MFMA). */
#define EMM_LDS __attribute__((address_space(3)))
typedef EMM_LDS int8_t lds8;
typedef EMM_LDS int lds32;
#define EMM_LD128(p) (*reinterpret_cast<const EMM_LDS v4i *>(p))
#define EMM_ST128(p, v) (*reinterpret_cast<EMM_LDS v4i *>(p) = (v))
#define EMM_LD32(p) (*reinterpret_cast<const EMM_LDS int *>(p))
#define EMM_ST32(p, v) (*reinterpret_cast<EMM_LDS uint32_t *>(p) = (v))

__device__ __forceinline__ int emm_med3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* i / d and i % d for 0 <= i < 2^24, 0 < d: one float multiply and a one-step correction instead of the ~20-instruction
 * integer division sequence (gfx950 has no integer divide); inv = 1.0f / d is computed once per loop. */
__device__ __forceinline__ void emm_divmod(int i, int d, float inv, int &q, int &r)
{
	q = (int)((float)i * inv);
	r = i - q * d;
	if (r < 0) { q--; r += d; }
	else if (r >= d) { q++; r -= d; }
}

/* a layer record out of LDS (the implicit struct copy does not take an address-space-3 source) */
template <typename T>
__device__ __forceinline__ void emm_copy_record(T *dst, const EMM_LDS T *src)
{
	const lds32 *s = reinterpret_cast<const lds32 *>(src);
	int *d = reinterpret_cast<int *>(dst);
#pragma unroll
	for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = s[i];
}

struct emm_layout { int hp, wp, py, px, img; }; /* how an activation tensor lies in LDS: padded dims, origin, bytes per image */

__device__ __forceinline__ emm_layout emm_in_layout(const EMM_LDS ed_mm_layer_t *ML, const EMM_LDS ed_net_layer_t *PL, int n_layers, int li)
{
	emm_layout l;
	if (li < n_layers)
	{
		l.hp = ML[li].in_hp; l.wp = ML[li].in_wp; l.py = ML[li].in_py; l.px = ML[li].in_px; l.img = ML[li].in_img;
	}
	else
	{
		const EMM_LDS ed_net_layer_t &L = PL[n_layers - 1];
		l.hp = L.out_h; l.wp = L.out_w; l.py = 0; l.px = 0; l.img = ((L.out_n + 15) & ~15) + 16;
	}
	return l;
}

__device__ __forceinline__ void emm_zero(lds8 *buf, int bytes, int lane)
{
	for (int i = lane * 16; i < bytes; i += 64 * 16) EMM_ST128(buf + i, ((v4i){0, 0, 0, 0}));
}

/* Order this wave's LDS writes before its following LDS reads: DS instructions of a wave are issued and serviced in
 * order; the (code-less) wave barrier keeps the compiler from moving memory operations across. */
__device__ __forceinline__ void emm_sync() { __builtin_amdgcn_wave_barrier(); }

template <bool FRAG_LDS>
__device__ __forceinline__ v4i emm_load_a(const lds8 *fl, const int8_t *fg, int s)
{
	if (FRAG_LDS) return EMM_LD128(fl + s * 1024);
	return *reinterpret_cast<const v4i *>(fg + (size_t)s * 1024);
}

/*
 * The k-loop of one 32-column tile for NW accumulator tiles at once (the positions of a fused pooling window, 1 when
 * nothing is fused): the A fragment of a k-step serves all of them, and the NW MFMA chains are independent. Software
 * pipeline in program order, so that every wait covers loads issued a whole step earlier (LDS returns in order):
 *   step s:  chunk offset of step s+2  |  A(s+1), B_w(s+1) at the offset read one step ago  |  MFMAs of step s
 * The first step is peeled: its MFMAs take the accumulator seeds as C directly (no copies). Returns the element-wise
 * maximum over the windows (max before the one requantisation is exact: the requantisation is monotone).
 */
/* The operands carried into the next k-step pass through an empty asm: without it the optimiser notices that "load for
 * step s+1, use one iteration later" equals "load at the top of step s+1", rotates the loop back and the pipeline is gone
 * (every MFMA then waits for two dependent LDS round trips). */
#define EMM_KEEP(a, k, b)                                                \
	do {                                                                 \
		asm volatile("" : "+v"(a), "+v"(k));                             \
		_Pragma("unroll") for (int w_ = 0; w_ < NW; w_++) asm volatile("" : "+v"(b[w_])); \
	} while (0)

template <int NW, bool FRAG_LDS>
__device__ __forceinline__ v16i emm_chain(const lds8 *fl, const int8_t *fg, const lds8 *kp /* &koff[h] */, const lds8 *(&bw)[NW], int n_ks,
                                          const v16i &seedv)
{
	v16i aw[NW];
	const int last = n_ks - 1;
	int k_cur = EMM_LD32(kp), k_nxt = EMM_LD32(kp + 8 * (last < 1 ? last : 1));
	v4i a = emm_load_a<FRAG_LDS>(fl, fg, 0), b[NW];
#pragma unroll
	for (int w = 0; w < NW; w++) b[w] = EMM_LD128(bw[w] + k_cur);
	{
		const int k3 = EMM_LD32(kp + 8 * (last < 2 ? last : 2));
		const v4i an = emm_load_a<FRAG_LDS>(fl, fg, last < 1 ? last : 1);
		v4i bn[NW];
#pragma unroll
		for (int w = 0; w < NW; w++) bn[w] = EMM_LD128(bw[w] + k_nxt);
#pragma unroll
		for (int w = 0; w < NW; w++) aw[w] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[w], seedv, 0, 0, 0);
		a = an; k_nxt = k3;
#pragma unroll
		for (int w = 0; w < NW; w++) b[w] = bn[w];
		EMM_KEEP(a, k_nxt, b);
	}
	for (int s = 1; s < n_ks; s++)
	{
		const int s1 = s + 1 < n_ks ? s + 1 : last, s2 = s + 2 < n_ks ? s + 2 : last;
		const int k3 = EMM_LD32(kp + 8 * s2);
		const v4i an = emm_load_a<FRAG_LDS>(fl, fg, s1);
		v4i bn[NW];
#pragma unroll
		for (int w = 0; w < NW; w++) bn[w] = EMM_LD128(bw[w] + k_nxt);
#pragma unroll
		for (int w = 0; w < NW; w++) aw[w] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[w], aw[w], 0, 0, 0);
		a = an; k_nxt = k3;
#pragma unroll
		for (int w = 0; w < NW; w++) b[w] = bn[w];
		EMM_KEEP(a, k_nxt, b);
	}
	v16i acc = aw[0];
#pragma unroll
	for (int w = 1; w < NW; w++)
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = aw[w][i] > acc[i] ? aw[w][i] : acc[i];
	return acc;
}

/* what the tile loop of a layer needs, gathered once per layer (all wave-uniform) */
struct emm_mm_args
{
	const lds8 *bsrc;       /* B source: the layer's input image(s) or their expanded copy               */
	lds8 *o;                /* where the epilogue stores                                                  */
	const lds8 *fragl; const int8_t *fragg; /* this layer's fragments (LDS resident / global)             */
	const lds8 *koff;       /* chunk offsets [2 s + h]                                                    */
	const lds8 *seeds;      /* 32 * n_rt accumulator seeds                                                */
	const lds8 *coltab;     /* (B offset, output offset) per stored pixel, or null                        */
	int img, o_img;         /* bytes per image of the B source / of the output layout                     */
	int n_ks, n_rt, n_cols, pix_per_img, col_w;
	int pitch_x, pitch_y, sh, ph, pw;
	int o_origin, o_row, oc_pitch, out_c, rs, lo_clamp;
};

template <int NW, bool FRAG_LDS>
__device__ __forceinline__ void emm_layer_tiles(const emm_mm_args &A, int lane)
{
	const int col = lane & 31, h = lane >> 5;
	const int n_ct = (A.n_cols + 31) >> 5;
	const float inv_ppi = 1.0f / (float)A.pix_per_img, inv_ow = 1.0f / (float)A.col_w;
	/* the windows of a column start (wy * sh) input rows / wx output columns after its first one */
	int wdelta[NW];
#pragma unroll
	for (int w = 0; w < NW; w++)
	{
		const int wy = w / A.pw, wx = w - wy * A.pw;
		wdelta[w] = (wy * A.sh) * A.pitch_y + wx * A.pitch_x;
	}
	for (int rt = 0; rt < A.n_rt; rt++)
	{
		v16i seedv;
		{
			const lds8 *sp = A.seeds + 4 * (32 * rt + 4 * h);
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const v4i s4 = EMM_LD128(sp + 32 * g);
				seedv[4 * g] = s4.x; seedv[4 * g + 1] = s4.y; seedv[4 * g + 2] = s4.z; seedv[4 * g + 3] = s4.w;
			}
		}
		const lds8 *fl = A.fragl + rt * A.n_ks * 1024 + lane * 16;
		const int8_t *fg = A.fragg + (size_t)rt * A.n_ks * 1024 + lane * 16;
		for (int ct = 0; ct < n_ct; ct++)
		{
			const int q = ct * 32 + col;
			const bool live = q < A.n_cols;
			const int qq = live ? q : A.n_cols - 1;
			int b = 0, pp = qq;
			if (A.n_cols > A.pix_per_img) emm_divmod(qq, A.pix_per_img, inv_ppi, b, pp); /* uniform: more than one image per wave */
			int boff, ooff;
			if (A.coltab)
			{
				boff = EMM_LD32(A.coltab + 8 * pp); ooff = EMM_LD32(A.coltab + 8 * pp + 4);
			}
			else
			{
				int y, x;
				emm_divmod(pp, A.col_w, inv_ow, y, x);
				boff = (y * A.ph * A.sh) * A.pitch_y + (x * A.pw) * A.pitch_x;
				ooff = A.o_origin + y * A.o_row + x * A.oc_pitch;
			}
			const lds8 *bw[NW];
#pragma unroll
			for (int w = 0; w < NW; w++) bw[w] = A.bsrc + b * A.img + boff + wdelta[w];
			const v16i acc = emm_chain<NW, FRAG_LDS>(fl, fg, A.koff + 4 * h, bw, A.n_ks, seedv);
			/* lane (column, h) holds rows 32 rt + 8 g + 4 h .. +3 in registers 4g..4g+3 */
			lds8 *op = A.o + b * A.o_img + ooff;
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const int r0 = 32 * rt + 8 * g + 4 * h;
				if (!live || r0 >= A.out_c) continue;
				const int v0 = emm_med3(acc[4 * g] >> A.rs, A.lo_clamp, 127), v1 = emm_med3(acc[4 * g + 1] >> A.rs, A.lo_clamp, 127);
				const int v2 = emm_med3(acc[4 * g + 2] >> A.rs, A.lo_clamp, 127), v3 = emm_med3(acc[4 * g + 3] >> A.rs, A.lo_clamp, 127);
				if ((A.out_c & 3) == 0)
					EMM_ST32(op + r0, (uint32_t)(uint8_t)v0 | ((uint32_t)(uint8_t)v1 << 8) | ((uint32_t)(uint8_t)v2 << 16) | ((uint32_t)(uint8_t)v3 << 24));
				else
				{
					op[r0] = (int8_t)v0;
					if (r0 + 1 < A.out_c) op[r0 + 1] = (int8_t)v1;
					if (r0 + 2 < A.out_c) op[r0 + 2] = (int8_t)v2;
					if (r0 + 3 < A.out_c) op[r0 + 3] = (int8_t)v3;
				}
			}
		}
	}
}

template <bool FRAG_LDS>
__global__ __launch_bounds__(EMM_MAX_THREADS) void ed_net_mfma_kernel(const ed_net_plan_t *__restrict__ P, const ed_mm_plan_t *__restrict__ M,
                                                                 const int8_t *__restrict__ frag, const int32_t *__restrict__ seeds,
                                                                 const int8_t *__restrict__ in, int64_t n, int64_t in_stride,
                                                                 int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                                                 int32_t *__restrict__ argmax)
{
	extern __shared__ __attribute__((aligned(16))) int8_t emm_lds_generic[];
	lds8 *emm_lds = (lds8 *)emm_lds_generic;
	const int n_layers = P->n_layers, batch = M->batch, buf_bytes = M->buf_bytes;
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n_threads = blockDim.x, n_waves = n_threads >> 6;
	/* small tables, copied once per workgroup: chunk offsets of every layer | seeds | layer records | column tables; then
	 * the weight fragments (when resident); then one slice per wave: two ping-pong activation buffers and the expansion
	 * buffer */
	lds8 *tbl = emm_lds;
	const int n_koff = M->n_koff, n_seeds = M->n_seeds, n_coltab = M->n_cols;
	lds32 *koff_all = reinterpret_cast<lds32 *>(tbl);
	lds32 *seeds_l = reinterpret_cast<lds32 *>(tbl + ((4 * n_koff + 15) & ~15));
	EMM_LDS ed_net_layer_t *PL = reinterpret_cast<EMM_LDS ed_net_layer_t *>(reinterpret_cast<lds8 *>(seeds_l) + ((4 * n_seeds + 15) & ~15));
	EMM_LDS ed_mm_layer_t *MLs = reinterpret_cast<EMM_LDS ed_mm_layer_t *>(reinterpret_cast<lds8 *>(PL) + ((n_layers * (int)sizeof(ed_net_layer_t) + 15) & ~15));
	lds32 *coltab_l = reinterpret_cast<lds32 *>(reinterpret_cast<lds8 *>(MLs) + ((n_layers * (int)sizeof(ed_mm_layer_t) + 15) & ~15));
	lds8 *fragl = tbl + M->tbl_bytes;
	lds8 *slice = fragl + M->frag_lds + wave * (2 * buf_bytes + M->x_bytes);
	lds8 *bufs[2] = {slice, slice + buf_bytes};
	lds8 *xbuf = slice + 2 * buf_bytes;
	{
		for (int i = threadIdx.x; i < n_koff; i += n_threads) koff_all[i] = M->koff[i];
		for (int i = threadIdx.x; i < n_seeds; i += n_threads) seeds_l[i] = seeds[i];
		for (int i = threadIdx.x; i < 2 * n_coltab; i += n_threads) coltab_l[i] = M->coltab[i];
		const int *s1 = reinterpret_cast<const int *>(&P->L[0]);
		lds32 *d1 = reinterpret_cast<lds32 *>(PL);
		for (int i = threadIdx.x; i < n_layers * (int)(sizeof(ed_net_layer_t) / 4); i += n_threads) d1[i] = s1[i];
		const int *s2 = reinterpret_cast<const int *>(&M->L[0]);
		lds32 *d2 = reinterpret_cast<lds32 *>(MLs);
		for (int i = threadIdx.x; i < n_layers * (int)(sizeof(ed_mm_layer_t) / 4); i += n_threads) d2[i] = s2[i];
		if (FRAG_LDS)
		{
			const v4i *src = reinterpret_cast<const v4i *>(frag);
			for (int i = threadIdx.x; i < M->frag_bytes / 16; i += n_threads) EMM_ST128(fragl + 16 * i, src[i]);
		}
	}
	__syncthreads(); /* the only workgroup barrier: from here on every wave is on its own */
	const int out_n = P->out_n, logits_layer = P->logits_layer, has_softmax = P->has_softmax;

#if EMM_STAMP
	unsigned long long stamp_[48], tl_;
	for (int i = 0; i < 48; i++) stamp_[i] = 0;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_) :: "memory");
#endif
	for (int64_t u0 = ((int64_t)blockIdx.x * n_waves + wave) * batch; u0 < n; u0 += (int64_t)gridDim.x * n_waves * batch)
	{
		const int nb = (int)((n - u0) < batch ? (n - u0) : batch);
		EMM_ST(47)
		/* ---- the inputs into layer 0's layout */
		{
			const emm_layout l0 = emm_in_layout(MLs, PL, n_layers, 0);
			const int in_h = P->in_h, in_w = P->in_w, in_c = P->in_c, in_n = P->in_n;
			const float inv_n = 1.0f / (float)in_n, inv_c = 1.0f / (float)in_c, inv_w = 1.0f / (float)in_w;
			emm_zero(bufs[0], batch * l0.img, lane);
			emm_sync();
			for (int i0 = 0; i0 < nb * in_n; i0 += 8 * 64)
			{
				int8_t v[8];
#pragma unroll
				for (int k = 0; k < 8; k++)
				{
					const int i = i0 + k * 64 + lane;
					int b, e;
					emm_divmod(i, in_n, inv_n, b, e);
					v[k] = i < nb * in_n ? in[(u0 + b) * in_stride + e] : 0;
				}
#pragma unroll
				for (int k = 0; k < 8; k++)
				{
					const int i = i0 + k * 64 + lane;
					if (i >= nb * in_n) continue;
					int b, e, pix, c, y, x;
					emm_divmod(i, in_n, inv_n, b, e); emm_divmod(e, in_c, inv_c, pix, c); emm_divmod(pix, in_w, inv_w, y, x);
					bufs[0][b * l0.img + ((y + l0.py) * l0.wp + x + l0.px) * in_c + c] = v[k];
				}
			}
			(void)in_h;
			emm_sync();
		}
		EMM_ST(0)
		int cur = 0;
		for (int li = 0; li < n_layers; li++)
		{
			ed_net_layer_t L;
			ed_mm_layer_t ML;
			emm_copy_record(&L, &PL[li]);
			emm_copy_record(&ML, &MLs[li]);
			if (ML.skip) continue; /* a MaxPool taken in the epilogue of the layer in front of it */
			const int fused = ML.pool_h > 0, lnx = fused ? li + 2 : li + 1; /* the layer that consumes what this one stores */
			const emm_layout lin = emm_in_layout(MLs, PL, n_layers, li), lo = emm_in_layout(MLs, PL, n_layers, lnx);
			const int st_h = fused ? PL[li + 1].out_h : L.out_h, st_w = fused ? PL[li + 1].out_w : L.out_w; /* stored tensor */
			const lds8 *a = bufs[cur];
			lds8 *o = bufs[cur ^ 1];
			cur ^= 1;
			const int oc_pitch = L.out_c;                           /* bytes per output pixel */
			const int o_origin = (lo.py * lo.wp + lo.px) * oc_pitch; /* where pixel (0, 0) goes */
			const int o_row = lo.wp * oc_pitch;
			if (lo.hp != st_h || lo.wp != st_w) /* uniform: the consumer wants a zero border */
			{
				emm_zero(o, batch * lo.img, lane);
				emm_sync();
			}
			if (ML.mm)
			{
				const int dense = L.type == ED_NET_DENSE;
				const int out_w = dense ? 1 : L.out_w;
				emm_mm_args A;
				A.bsrc = a; A.img = lin.img;
				if (ML.expand)
				{
					/* one aligned record of 16 * cpr bytes per (input row, output x): the kw * C_in bytes under a kernel row */
					const int in_c = dense ? L.in_n : L.in_c, seg = (dense ? 1 : L.kw) * in_c, sw = dense ? 1 : L.sw;
					const int rec_per_img = (dense ? 1 : lin.hp) * out_w * ML.cpr;
					const float inv_rec = 1.0f / (float)rec_per_img, inv_row = 1.0f / (float)(out_w * ML.cpr), inv_cpr = 1.0f / (float)ML.cpr;
					for (int i = lane; i < nb * rec_per_img; i += 64)
					{
						int b, e, r, e2, xo, j;
						emm_divmod(i, rec_per_img, inv_rec, b, e); emm_divmod(e, out_w * ML.cpr, inv_row, r, e2); emm_divmod(e2, ML.cpr, inv_cpr, xo, j);
						/* 16 bytes from an arbitrary byte offset: five aligned dwords around them, funnel-shifted (v_alignbit), the
						 * bytes past the end of the kernel-row segment zeroed (the image buffers carry 16 bytes of slack) */
						const int soff = b * lin.img + (r * lin.wp + xo * sw) * in_c + 16 * j;
						const lds8 *s4 = a + (soff & ~3);
						const uint32_t sh = (uint32_t)(soff & 3) * 8;
						const uint32_t w0 = (uint32_t)EMM_LD32(s4), w1 = (uint32_t)EMM_LD32(s4 + 4), w2 = (uint32_t)EMM_LD32(s4 + 8), w3 = (uint32_t)EMM_LD32(s4 + 12), w4 = (uint32_t)EMM_LD32(s4 + 16);
						uint32_t d[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
						                 __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
						const int keep = seg - 16 * j; /* bytes of this chunk that belong to the segment (>= 1) */
#pragma unroll
						for (int t = 0; t < 4; t++)
						{
							const int kb = keep - 4 * t;
							d[t] = kb >= 4 ? d[t] : (kb <= 0 ? 0u : d[t] & (0xffffffffu >> (8 * (4 - kb))));
						}
						EMM_ST128(xbuf + b * ML.x_img + r * ML.pitch_y + xo * ML.pitch_x + 16 * j, ((v4i){(int)d[0], (int)d[1], (int)d[2], (int)d[3]}));
					}
					A.bsrc = xbuf;
					A.img = ML.x_img;
				}
				EMM_ST(1 + 5 * li)
				emm_sync();
				A.o = o; A.o_img = lo.img;
				A.fragl = fragl + ML.frag_off; A.fragg = frag + ML.frag_off;
				A.koff = reinterpret_cast<const lds8 *>(koff_all + ML.koff_off);
				A.seeds = reinterpret_cast<const lds8 *>(seeds_l + ML.seed_off);
				A.coltab = ML.col_off >= 0 ? reinterpret_cast<const lds8 *>(coltab_l + 2 * ML.col_off) : nullptr;
				A.n_ks = ML.n_ks; A.n_rt = ML.n_rt;
				A.col_w = dense ? 1 : st_w;
				A.pix_per_img = dense ? 1 : st_h * st_w;
				A.n_cols = nb * A.pix_per_img;
				A.pitch_x = ML.pitch_x; A.pitch_y = ML.pitch_y; A.sh = dense ? 1 : L.sh;
				A.ph = fused ? ML.pool_h : 1; A.pw = fused ? ML.pool_w : 1;
				A.o_origin = o_origin; A.o_row = o_row; A.oc_pitch = oc_pitch; A.out_c = L.out_c; A.rs = L.rs; A.lo_clamp = L.relu ? 0 : -128;
				const int nwin = A.ph * A.pw; /* 1, 2 or 4 (model_net_mm.c fuses nothing else) */
				if (nwin == 1) emm_layer_tiles<1, FRAG_LDS>(A, lane);
				else if (nwin == 2) emm_layer_tiles<2, FRAG_LDS>(A, lane);
				else emm_layer_tiles<4, FRAG_LDS>(A, lane);
			}
			else if (L.type == ED_NET_POOL && (L.in_c & 3) == 0)
			{
				/* four channels per thread: byte-wise signed maximum of dwords */
				const int c4n = L.in_c >> 2, per_img = L.out_h * L.out_w * c4n;
				const float inv_img = 1.0f / (float)per_img, inv_c4 = 1.0f / (float)c4n, inv_ow = 1.0f / (float)L.out_w;
				for (int i = lane; i < nb * per_img; i += 64)
				{
					int b, e, pix, c4, y, x;
					emm_divmod(i, per_img, inv_img, b, e); emm_divmod(e, c4n, inv_c4, pix, c4); emm_divmod(pix, L.out_w, inv_ow, y, x);
					int m0 = -129, m1 = -129, m2 = -129, m3 = -129;
					for (int ky = 0; ky < L.kh; ky++)
					{
						const int iy = y * L.sh - L.pad_h + ky;
						if ((unsigned)iy >= (unsigned)L.in_h) continue;
						for (int kx = 0; kx < L.kw; kx++)
						{
							const int ix = x * L.sw - L.pad_w + kx;
							if ((unsigned)ix >= (unsigned)L.in_w) continue;
							const int v = EMM_LD32(a + b * lin.img + (iy * L.in_w + ix) * L.in_c + 4 * c4);
							const int v0 = (int)(int8_t)v, v1 = (int)(int8_t)(v >> 8), v2 = (int)(int8_t)(v >> 16), v3 = v >> 24;
							m0 = v0 > m0 ? v0 : m0; m1 = v1 > m1 ? v1 : m1; m2 = v2 > m2 ? v2 : m2; m3 = v3 > m3 ? v3 : m3;
						}
					}
					EMM_ST32(o + b * lo.img + o_origin + y * o_row + x * oc_pitch + 4 * c4,
					         (uint32_t)(uint8_t)m0 | ((uint32_t)(uint8_t)m1 << 8) | ((uint32_t)(uint8_t)m2 << 16) | ((uint32_t)(uint8_t)m3 << 24));
				}
			}
			else if (L.type == ED_NET_POOL)
			{
				const int per_img = L.out_n;
				for (int i = lane; i < nb * per_img; i += 64)
				{
					const int b = i / per_img, e = i - b * per_img;
					const int pix = e / L.in_c, c = e - pix * L.in_c, y = pix / L.out_w, x = pix - y * L.out_w;
					int mx = -129;
					for (int ky = 0; ky < L.kh; ky++)
					{
						const int iy = y * L.sh - L.pad_h + ky;
						if ((unsigned)iy >= (unsigned)L.in_h) continue;
						for (int kx = 0; kx < L.kw; kx++)
						{
							const int ix = x * L.sw - L.pad_w + kx;
							if ((unsigned)ix >= (unsigned)L.in_w) continue;
							const int v = a[b * lin.img + (iy * L.in_w + ix) * L.in_c + c];
							mx = v > mx ? v : mx;
						}
					}
					o[b * lo.img + o_origin + y * o_row + x * oc_pitch + c] = (int8_t)mx;
				}
			}
			else /* softmax: arm_softmax_q7.c:215-260, one lane per input */
			{
				if (lane < nb)
				{
					const lds8 *v = a + lane * lin.img;
					lds8 *w = o + lane * lo.img;
					if (L.in_n <= 16)
					{
						/* the usual classifier width: one 16-byte read, everything else in registers */
						const v4i raw = EMM_LD128(v);
						const uint32_t rw[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
						int base = -128;
#pragma unroll
						for (int i = 0; i < 16; i++) { const int x = (int)(int8_t)(rw[i >> 2] >> (8 * (i & 3))); if (i < L.in_n && x > base) base = x; }
						base -= 8;
						int sum = 0;
#pragma unroll
						for (int i = 0; i < 16; i++) { const int x = (int)(int8_t)(rw[i >> 2] >> (8 * (i & 3))); if (i < L.in_n) sum += 1 << emm_med3(x - base, 0, 7); }
						const int output_base = (1 << 20) / sum;
						uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
						for (int i = 0; i < 16; i++)
						{
							const int x = (int)(int8_t)(rw[i >> 2] >> (8 * (i & 3)));
							const int r = emm_med3(output_base >> emm_med3(13 + base - x, 0, 31), -128, 127);
							if (i < L.in_n) ow[i >> 2] |= (uint32_t)(uint8_t)r << (8 * (i & 3));
						}
						EMM_ST128(w, ((v4i){(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]}));
					}
					else
					{
						int base = -128;
						for (int i = 0; i < L.in_n; i++) base = v[i] > base ? v[i] : base;
						base -= 8;
						int sum = 0;
						for (int i = 0; i < L.in_n; i++) sum += 1 << emm_med3(v[i] - base, 0, 7);
						const int output_base = (1 << 20) / sum;
						for (int i = 0; i < L.in_n; i++) w[i] = (int8_t)emm_med3(output_base >> emm_med3(13 + base - v[i], 0, 31), -128, 127);
					}
				}
			}
			EMM_ST(2 + 5 * li)
			emm_sync();
			/* outputs (the layouts of the logits layer's and the last layer's outputs are compact) */
			if (li == logits_layer && logits)
				for (int i = lane; i < nb * out_n; i += 64)
					logits[(u0 + i / out_n) * out_n + i % out_n] = o[(i / out_n) * lo.img + i % out_n];
			if (li == n_layers - 1)
			{
				if (has_softmax && softmax)
					for (int i = lane; i < nb * out_n; i += 64)
						softmax[(u0 + i / out_n) * out_n + i % out_n] = o[(i / out_n) * lo.img + i % out_n];
				if (argmax && lane < nb)
				{
					const lds8 *v = o + lane * lo.img;
					int best = 0, mx = -129;
					for (int i = 0; i < out_n; i++)
						if (v[i] > mx) { mx = v[i]; best = i; }
					argmax[u0 + lane] = best;
				}
			}
		}
		EMM_ST(46)
		emm_sync(); /* the next batch overwrites both buffers */
	}
#if EMM_STAMP
	if (g_emm_dbg && threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 48; i++) g_emm_dbg[i] = stamp_[i];
#endif
}

extern "C" int ed_launch_net_mfma(const ed_net_plan_t *dev_plan, const ed_mm_plan_t *dev_mm, const int8_t *dev_frag,
                                  const int32_t *dev_seeds, int lds_bytes, int batch, int waves, int frag_mode, const int8_t *in, int64_t n,
                                  int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu, hipStream_t stream)
{
	if (n <= 0) return 0;
	if (waves < 1 || waves > EMM_MAX_THREADS / 64) return (int)hipErrorInvalidValue;
	int per_cu = (160 * 1024) / (lds_bytes + 256);
	if (per_cu > 32 / waves) per_cu = 32 / waves;
	if (per_cu < 1) per_cu = 1;
	const int64_t per_block = (int64_t)batch * waves;
	int64_t blocks = (n + per_block - 1) / per_block;
	if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
	const int resident = frag_mode == 2;
	const void *fn = resident ? (const void *)ed_net_mfma_kernel<true> : (const void *)ed_net_mfma_kernel<false>;
	static int max_lds_set[2] = {0, 0};
	if (lds_bytes > max_lds_set[resident])
	{
		/* more than 64 KB of dynamic LDS has to be asked for */
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
		if (e != hipSuccess) return (int)e;
		max_lds_set[resident] = lds_bytes;
	}
	void *kargs[] = {(void *)&dev_plan, (void *)&dev_mm, (void *)&dev_frag, (void *)&dev_seeds, (void *)&in, (void *)&n, (void *)&in_stride,
	                 (void *)&logits, (void *)&softmax, (void *)&argmax};
	return (int)hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(64 * waves), kargs, (size_t)lds_bytes, stream);
}
