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
#define EMM_MAX_THREADS 1024

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

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

struct emm_layout { int hp, wp, py, px, img; }; /* how an activation tensor lies in LDS: padded dims, origin, bytes per image */

__device__ __forceinline__ emm_layout emm_in_layout(const ed_mm_layer_t *ML, const ed_net_layer_t *PL, int n_layers, int li)
{
	emm_layout l;
	if (li < n_layers)
	{
		l.hp = ML[li].in_hp; l.wp = ML[li].in_wp; l.py = ML[li].in_py; l.px = ML[li].in_px; l.img = ML[li].in_img;
	}
	else
	{
		const ed_net_layer_t &L = PL[n_layers - 1];
		l.hp = L.out_h; l.wp = L.out_w; l.py = 0; l.px = 0; l.img = ((L.out_n + 15) & ~15) + 16;
	}
	return l;
}

__device__ __forceinline__ void emm_zero(int8_t *buf, int bytes, int lane)
{
	for (int i = lane * 16; i < bytes; i += 64 * 16) *reinterpret_cast<uint4 *>(buf + i) = make_uint4(0, 0, 0, 0);
}

/* Order this wave's LDS writes before its following LDS reads: DS instructions of a wave are issued and serviced in
 * order; the (code-less) wave barrier keeps the compiler from moving memory operations across. */
__device__ __forceinline__ void emm_sync() { __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(EMM_MAX_THREADS) void ed_net_mfma_kernel(const ed_net_plan_t *__restrict__ P, const ed_mm_plan_t *__restrict__ M,
                                                                 const int8_t *__restrict__ frag, const int32_t *__restrict__ seeds,
                                                                 const int8_t *__restrict__ in, int64_t n, int64_t in_stride,
                                                                 int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                                                 int32_t *__restrict__ argmax)
{
	extern __shared__ __attribute__((aligned(16))) int8_t emm_lds[];
	const int n_layers = P->n_layers, batch = M->batch, buf_bytes = M->buf_bytes;
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, h = lane >> 5;
	const int n_threads = blockDim.x, n_waves = n_threads >> 6;
	/* small tables, copied once per workgroup: chunk offsets of every layer | seeds | layer records; then the weight
	 * fragments (mode 2); then one slice per wave: two ping-pong activation buffers and the expansion buffer */
	int8_t *tbl = emm_lds;
	const int n_koff = M->n_koff, n_seeds = M->n_seeds;
	int *koff_all = reinterpret_cast<int *>(tbl);
	int32_t *seeds_l = reinterpret_cast<int32_t *>(tbl + ((4 * n_koff + 15) & ~15));
	ed_net_layer_t *PL = reinterpret_cast<ed_net_layer_t *>(reinterpret_cast<int8_t *>(seeds_l) + ((4 * n_seeds + 15) & ~15));
	ed_mm_layer_t *MLs = reinterpret_cast<ed_mm_layer_t *>(reinterpret_cast<int8_t *>(PL) + ((n_layers * (int)sizeof(ed_net_layer_t) + 15) & ~15));
	int8_t *fragl = tbl + M->tbl_bytes;
	const int frag_mode = M->frag_mode;
	int8_t *slice = fragl + M->frag_lds + wave * (2 * buf_bytes + M->x_bytes);
	int8_t *bufs[2] = {slice, slice + buf_bytes};
	int8_t *xbuf = slice + 2 * buf_bytes;
	{
		for (int i = threadIdx.x; i < n_koff; i += n_threads) koff_all[i] = M->koff[i];
		for (int i = threadIdx.x; i < n_seeds; i += n_threads) seeds_l[i] = seeds[i];
		const int *s1 = reinterpret_cast<const int *>(&P->L[0]);
		int *d1 = reinterpret_cast<int *>(PL);
		for (int i = threadIdx.x; i < n_layers * (int)(sizeof(ed_net_layer_t) / 4); i += n_threads) d1[i] = s1[i];
		const int *s2 = reinterpret_cast<const int *>(&M->L[0]);
		int *d2 = reinterpret_cast<int *>(MLs);
		for (int i = threadIdx.x; i < n_layers * (int)(sizeof(ed_mm_layer_t) / 4); i += n_threads) d2[i] = s2[i];
		if (frag_mode == 2)
		{
			const uint4 *src = reinterpret_cast<const uint4 *>(frag);
			uint4 *dst = reinterpret_cast<uint4 *>(fragl);
			for (int i = threadIdx.x; i < M->frag_bytes / 16; i += n_threads) dst[i] = src[i];
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
			const ed_net_layer_t L = PL[li];
			const ed_mm_layer_t ML = MLs[li];
			if (ML.skip) continue; /* a MaxPool taken in the epilogue of the layer in front of it */
			const int fused = ML.pool_h > 0, lnx = fused ? li + 2 : li + 1; /* the layer that consumes what this one stores */
			const emm_layout lin = emm_in_layout(MLs, PL, n_layers, li), lo = emm_in_layout(MLs, PL, n_layers, lnx);
			const int st_h = fused ? PL[li + 1].out_h : L.out_h, st_w = fused ? PL[li + 1].out_w : L.out_w; /* stored tensor */
			const int8_t *a = bufs[cur];
			int8_t *o = bufs[cur ^ 1];
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
				const int out_w = dense ? 1 : L.out_w, sh = dense ? 1 : L.sh;
				const int ph = fused ? ML.pool_h : 1, pw = fused ? ML.pool_w : 1, nwin = ph * pw; /* accumulator tiles per column */
				const int col_h = dense ? 1 : st_h, col_w = dense ? 1 : st_w;                       /* columns = stored pixels */
				const int8_t *bsrc = a;
				int img = lin.img;
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
						const uint32_t *s4 = reinterpret_cast<const uint32_t *>(a + (soff & ~3));
						const uint32_t sh = (uint32_t)(soff & 3) * 8;
						const uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2], w3 = s4[3], w4 = s4[4];
						uint32_t d[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
						                 __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
						const int keep = seg - 16 * j; /* bytes of this chunk that belong to the segment (>= 1) */
#pragma unroll
						for (int t = 0; t < 4; t++)
						{
							const int kb = keep - 4 * t;
							d[t] = kb >= 4 ? d[t] : (kb <= 0 ? 0u : d[t] & (0xffffffffu >> (8 * (4 - kb))));
						}
						*reinterpret_cast<uint4 *>(xbuf + b * ML.x_img + r * ML.pitch_y + xo * ML.pitch_x + 16 * j) = make_uint4(d[0], d[1], d[2], d[3]);
					}
					bsrc = xbuf;
					img = ML.x_img;
				}
				EMM_ST(1 + 5 * li)
				const int *koff = koff_all + ML.koff_off;
				emm_sync();
				const int pix_per_img = col_h * col_w, n_cols = nb * pix_per_img, n_ct = (n_cols + 31) / 32;
				const int rs = L.rs, lo_clamp = L.relu ? 0 : -128;
				const float inv_rt = 1.0f / (float)ML.n_rt, inv_ppi = 1.0f / (float)pix_per_img, inv_ow = 1.0f / (float)col_w;
				for (int t = 0; t < n_ct * ML.n_rt; t++)
				{
					int ct, rt;
					emm_divmod(t, ML.n_rt, inv_rt, ct, rt);
					const int q = ct * 32 + col;
					const bool live = q < n_cols;
					const int qq = live ? q : n_cols - 1;
					int b, pp, y, x;
					emm_divmod(qq, pix_per_img, inv_ppi, b, pp); emm_divmod(pp, col_w, inv_ow, y, x);
					const int8_t *fp = (frag_mode == 2 ? fragl + ML.frag_off : frag + ML.frag_off) + (size_t)rt * ML.n_ks * 1024 + lane * 16;
					v16i seedv;
					{
						const int32_t *sp = seeds_l + ML.seed_off + 32 * rt + 4 * h;
#pragma unroll
						for (int g = 0; g < 4; g++)
						{
							const v4i s4 = *reinterpret_cast<const v4i *>(sp + 8 * g);
							seedv[4 * g] = s4.x; seedv[4 * g + 1] = s4.y; seedv[4 * g + 2] = s4.z; seedv[4 * g + 3] = s4.w;
						}
					}
					/* one accumulator tile per position of the pooling window (one in all when nothing is fused); the
					 * window's conv pixels are (y * ph + wy, x * pw + wx) */
					v16i acc = seedv;
#pragma unroll
					for (int w = 0; w < 4; w++)
					{
						if (w >= nwin) break;
						const int wy = w / pw, wx = w - wy * pw;
						const int8_t *bp = bsrc + b * img + ((y * ph + wy) * sh) * ML.pitch_y + (x * pw + wx) * ML.pitch_x;
						v16i aw = seedv;
						/* operands of k-step s + 1 are fetched before the MFMA of k-step s */
						v4i av = *reinterpret_cast<const v4i *>(fp), bv = *reinterpret_cast<const v4i *>(bp + koff[h]);
						for (int s = 0; s < ML.n_ks; s++)
						{
							const int sn = s + 1 < ML.n_ks ? s + 1 : s;
							const v4i an = *reinterpret_cast<const v4i *>(fp + (size_t)sn * 1024);
							const v4i bn = *reinterpret_cast<const v4i *>(bp + koff[2 * sn + h]);
							aw = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, aw, 0, 0, 0);
							av = an; bv = bn;
						}
						if (w == 0) acc = aw;
						else
						{
#pragma unroll
							for (int i = 0; i < 16; i++) acc[i] = aw[i] > acc[i] ? aw[i] : acc[i];
						}
					}
					/* lane (column, h) holds rows 32 rt + 8 g + 4 h .. +3 in registers 4g..4g+3 */
					int8_t *op = o + b * lo.img + o_origin + y * o_row + x * oc_pitch;
#pragma unroll
					for (int g = 0; g < 4; g++)
					{
						const int r0 = 32 * rt + 8 * g + 4 * h;
						if (!live || r0 >= L.out_c) continue;
						const int v0 = emm_med3(acc[4 * g] >> rs, lo_clamp, 127), v1 = emm_med3(acc[4 * g + 1] >> rs, lo_clamp, 127);
						const int v2 = emm_med3(acc[4 * g + 2] >> rs, lo_clamp, 127), v3 = emm_med3(acc[4 * g + 3] >> rs, lo_clamp, 127);
						if ((L.out_c & 3) == 0)
							*reinterpret_cast<uint32_t *>(op + r0) = (uint32_t)(uint8_t)v0 | ((uint32_t)(uint8_t)v1 << 8) | ((uint32_t)(uint8_t)v2 << 16) | ((uint32_t)(uint8_t)v3 << 24);
						else
						{
							op[r0] = (int8_t)v0;
							if (r0 + 1 < L.out_c) op[r0 + 1] = (int8_t)v1;
							if (r0 + 2 < L.out_c) op[r0 + 2] = (int8_t)v2;
							if (r0 + 3 < L.out_c) op[r0 + 3] = (int8_t)v3;
						}
					}
				}
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
							const int v = *reinterpret_cast<const int *>(a + b * lin.img + (iy * L.in_w + ix) * L.in_c + 4 * c4);
							const int v0 = (int)(int8_t)v, v1 = (int)(int8_t)(v >> 8), v2 = (int)(int8_t)(v >> 16), v3 = v >> 24;
							m0 = v0 > m0 ? v0 : m0; m1 = v1 > m1 ? v1 : m1; m2 = v2 > m2 ? v2 : m2; m3 = v3 > m3 ? v3 : m3;
						}
					}
					*reinterpret_cast<uint32_t *>(o + b * lo.img + o_origin + y * o_row + x * oc_pitch + 4 * c4) =
					    (uint32_t)(uint8_t)m0 | ((uint32_t)(uint8_t)m1 << 8) | ((uint32_t)(uint8_t)m2 << 16) | ((uint32_t)(uint8_t)m3 << 24);
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
					const int8_t *v = a + lane * lin.img;
					int8_t *w = o + lane * lo.img;
					if (L.in_n <= 16)
					{
						/* the usual classifier width: one 16-byte read, everything else in registers */
						const uint4 raw = *reinterpret_cast<const uint4 *>(v);
						const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
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
						*reinterpret_cast<uint4 *>(w) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
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
					const int8_t *v = o + lane * lo.img;
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
                                  const int32_t *dev_seeds, int lds_bytes, int batch, int waves, const int8_t *in, int64_t n, int64_t in_stride,
                                  int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu, hipStream_t stream)
{
	if (n <= 0) return 0;
	if (waves < 1 || waves > EMM_MAX_THREADS / 64) return (int)hipErrorInvalidValue;
	int per_cu = (160 * 1024) / (lds_bytes + 256);
	if (per_cu > 32 / waves) per_cu = 32 / waves;
	if (per_cu < 1) per_cu = 1;
	const int64_t per_block = (int64_t)batch * waves;
	int64_t blocks = (n + per_block - 1) / per_block;
	if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
	static int max_lds_set = 0;
	if (lds_bytes > max_lds_set)
	{
		/* more than 64 KB of dynamic LDS has to be asked for */
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ed_net_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
		if (e != hipSuccess) return (int)e;
		max_lds_set = lds_bytes;
	}
	hipLaunchKernelGGL(ed_net_mfma_kernel, dim3((unsigned)blocks), dim3(64 * waves), (size_t)lds_bytes, stream, dev_plan, dev_mm,
	                   dev_frag, dev_seeds, in, n, in_stride, logits, softmax, argmax);
	return (int)hipGetLastError();
}
