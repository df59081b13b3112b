/*
 * mfcc_one_frame.h -- ed_mfcc_kernel's body (one 1024-sample frame per wavefront, fp32: the reference implementation of the
 * design in mfcc_kernels.hip, see the pipeline description there) and what it needs, in a header: mfcc_kernels.hip builds the
 * batch / stage-dump kernel from it, cnn_mfma_kernels.hip the one-launch microphone push (ed_kws1_kernel).
 */
#ifndef ED_MFCC_ONE_FRAME_H
#define ED_MFCC_ONE_FRAME_H

/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles */
#if !defined(ED_LAB) && (defined(ED_WPB))
#error "ED_WPB lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#ifndef ED_WPB
#define ED_WPB 8                 /* waves (= frames in flight) per workgroup                               */
#endif
#define ED_XBUF_FLOATS 1160      /* per-wave LDS: 526 complex transpose slots | spectrum S[516] at 576 | u,v at 1104 */
#define ED_S_OFF 576
#define ED_L_OFF 1104
#define ED_FIXTAB_FLOATS (2 * 64 * 4 + 4 * 64 * 2) /* dct | split twiddles, then (NLO+NHI) x 64 weight quads */


#define ED_MFCC_BOUNDS __launch_bounds__(64 * ED_WPB)

#include "mfcc_fft.h"

/* First sample of frame f: (f / fpg) * group_stride + (f % fpg) * frame_step (f is wave-uniform, < 2^31). */
__device__ __forceinline__ const int16_t *ed_frame_ptr(const ed_mfcc_args_t &a, uint32_t f)
{
	uint32_t g = 0, i = f;
	if (a.frames_per_group < a.n_frames)
	{
		g = f / (uint32_t)a.frames_per_group;
		i = f - g * (uint32_t)a.frames_per_group;
	}
	return a.audio + ((int64_t)g * a.group_stride + (int64_t)i * a.frame_step);
}

template <bool ALIGNED>
__device__ __forceinline__ void ed_load_frame(const int16_t *fp, int lane, uint32_t (&v)[8])
{
	if (ALIGNED)
	{
		const uint32_t *fp32 = reinterpret_cast<const uint32_t *>(fp);
		/* the samples are read once: non-temporal loads (global_load_dword ... nt) keep them from displacing what the caches are
		 * for (tables, the feature rows the CNN reads next): +2.2 ... +2.6 % on the 65 536-frame launch, +1.0 % on the Q15
		 * kernel (interleaved A/B, profiles/r03_wave_priorities.txt); agent- / system-scope loads (sc1, sc0 sc1) and
		 * non-temporal STORES of the coefficients measure -0.4 ... -0.8 %. */
#pragma unroll
		for (int a = 0; a < 8; a++) v[a] = __builtin_nontemporal_load(fp32 + lane + 64 * a);
	}
	else
	{
		const uint16_t *fu = reinterpret_cast<const uint16_t *>(fp);
#pragma unroll
		for (int a = 0; a < 8; a++)
			v[a] = (uint32_t)fu[2 * (lane + 64 * a)] | ((uint32_t)fu[2 * (lane + 64 * a) + 1] << 16);
	}
}

/* The body of ed_mfcc_kernel as a function of where its LDS lies and which frames the calling wave takes, so that another
 * kernel can run it too (ed_kws1_kernel, cnn_mfma_kernels.hip: the one-frame microphone push in ONE launch). Every thread of
 * the workgroup must call it (table staging + one workgroup barrier); a wave takes frames f_first, f_first + f_stride, ...
 * below args.n_frames -- none at all if f_first is not below it. feat2: a second place for the int8 feature row (any address
 * space, e.g. LDS), or null. */
template <bool STAGES, bool ALIGNED, int NLO, int NHI, bool WINDOW = false>
__device__ __forceinline__ void ed_mfcc1_body(const ed_mfcc_args_t &args, const ed_mfcc_tables_t *__restrict__ tab, float *smem, int wave,
                                              uint32_t f_first, uint32_t f_stride, int8_t *feat2)
{
	const int lane = threadIdx.x & 63;
	const float4 *dctl = reinterpret_cast<const float4 *>(smem);                 /* [2][64] x 4 coefficients */
	const float2 *tpl = reinterpret_cast<const float2 *>(smem + 512);            /* [4][64] split twiddles   */
	const float4 *melw4 = reinterpret_cast<const float4 *>(smem + ED_FIXTAB_FLOATS); /* [NLO+NHI][64] quads   */
	float *xbuf = smem + ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + wave * ED_XBUF_FLOATS; /* wave-private */

	/* the first frame's samples go in flight before anything else: their HBM latency hides under the table staging */
	const uint32_t n_frames = (uint32_t)args.n_frames;
	const uint32_t stride = f_stride;
	uint32_t f = f_first;
	uint32_t raw[8];
	if (f < n_frames) ed_load_frame<ALIGNED>(ed_frame_ptr(args, f), lane, raw);

	{ /* the table block [dct4 | twp | mel_w4(NLO+NHI rows)] is laid out in global memory exactly as in LDS */
		const float4 *src = reinterpret_cast<const float4 *>(&tab->dct4[0][0][0]);
		float4 *dst = reinterpret_cast<float4 *>(smem);
		for (int t = threadIdx.x; t < (ED_FIXTAB_FLOATS + (NLO + NHI) * 256) / 4; t += blockDim.x) dst[t] = src[t];
	}

	/* pass-1/2 twiddles: resident in registers for the whole persistent loop */
	float t1r[8], t1i[8], t2r[8], t2i[8];
#pragma unroll
	for (int p = 1; p < 8; p++)
	{
		const float2 a = *reinterpret_cast<const float2 *>(&tab->tw1[p][lane][0]);
		const float2 b = *reinterpret_cast<const float2 *>(&tab->tw2[p][lane][0]);
		t1r[p] = a.x; t1i[p] = a.y; t2r[p] = b.x; t2i[p] = b.y;
	}
	__syncthreads();
	const int mel_slo4 = tab->mel_slo4[lane], mel_shi4 = tab->mel_shi4[lane];
	const int band = tab->mel_band[lane]; /* this column's narrow band b; its wide band is 31 - b */
	const float spec_scale = tab->spec_scale;
	const float log_offset = tab->log_offset;
	const bool do_log = tab->always_log || args.use_log;
	/* after the two transposes this lane holds Z[k0 + 64r] in register r (ED_K0: natural order with the LDS
	 * transpose, octal-digit-swapped with the DPP one) */
	const int k0 = ED_K0(lane);
	const int k0p = (64 - k0) & 63;                          /* low 6 bits of the partner index 512 - k    */
	const int pull = k0p << 2; /* lane that holds it, as a byte address */
	const int hi3 = lane >> 3, lo3 = lane & 7;
	float2 *xc = reinterpret_cast<float2 *>(xbuf);
	(void)hi3; (void)lo3; (void)xc;
	if (lane < 3) xbuf[ED_S_OFF + 513 + lane] = 0.0f;        /* spectrum padding: read by the last quad only */

	for (; f < n_frames; f += stride)
	{
		/* ---- 1. unpack this frame, then put the next frame's loads in flight */
		float re[8], im[8];
#pragma unroll
		for (int a = 0; a < 8; a++)
		{
			re[a] = (float)(int16_t)(raw[a] & 0xffffu);
			im[a] = (float)(int16_t)(raw[a] >> 16);
		}
		if (WINDOW) /* variant TF: tf.signal.stft's Hann window on the float32 samples, as (w[2m], w[2m+1]) per packed point */
		{
#pragma unroll
			for (int a = 0; a < 8; a++)
			{
				const float2 w = *reinterpret_cast<const float2 *>(&tab->window2[lane + 64 * a][0]);
				re[a] *= w.x;
				im[a] *= w.y;
			}
		}
		if (f + stride < n_frames) ed_load_frame<ALIGNED>(ed_frame_ptr(args, f + stride), lane, raw);

		/* ---- 2a. pass 1: DFT over a (registers), twiddle W512^(lane*p); lane = 8b + c */
		ed_radix8(re, im);
#pragma unroll
		for (int p = 1; p < 8; p++)
		{
			const float wr = t1r[p], wi = t1i[p];
			float xr = re[p], xi = im[p];
			re[p] = xr * wr - xi * wi;
			im[p] = xr * wi + xi * wr;
		}
		/* transpose 1: register p <-> lane bits 3..5 (b): (lane 8b+c, reg p) -> (lane 8p+c, reg b) */
		ed_transpose8<3, 4, 5>(re, lane);
		ed_transpose8<3, 4, 5>(im, lane);

		/* ---- 2b. pass 2: DFT over b, twiddle W64^(c*q); lane = 8p + c */
		ed_radix8(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			const float wr = t2r[q], wi = t2i[q];
			float xr = re[q], xi = im[q];
			re[q] = xr * wr - xi * wi;
			im[q] = xr * wi + xi * wr;
		}
		/* transpose 2 through the wave-private LDS buffer: (lane 8p+c, reg q) -> (lane p+8q, reg c); slot
		 * 66c + p + 8q is conflict-free for the ds_write_b64 (16-lane groups) and the ds_read_b64 alike */
#pragma unroll
		for (int q = 0; q < 8; q++) xc[66 * lo3 + hi3 + 8 * q] = make_float2(re[q], im[q]);
		ed_wave_sync();
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			float2 v = xc[66 * c + lane];
			re[c] = v.x; im[c] = v.y;
		}
		ed_wave_sync();

		/* ---- 2c. pass 3: DFT over c  ->  reg r holds Z[k0 + 64r], k0 = p + 8q */
		ed_radix8(re, im);

		/* ---- 3. real-FFT split. The partner Z[512-k] of k = k0 + 64m (m < 4) is register 7-m of the lane whose
		 *         k0 is (64 - k0) % 64: pulled through the LDS crossbar (ds_bpermute, no LDS memory, one trip).
		 *         Lane 0 (k0 = 0) is its own partner, one register further up: Z[512 - 64m] = its register 8-m. */
		float slo[4], shi[4];
		float flr[4], fli[4], fhr[4], fhi[4]; /* X[k], X[512-k] for the stage dump */
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			float2 pz;
			pz.x = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(re[7 - m])));
			pz.y = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(im[7 - m])));
			if (lane == 0) pz = make_float2(re[(8 - m) & 7], im[(8 - m) & 7]);
			const float2 tw = tpl[64 * m + lane];       /* W1024^(k0 + 64m)                                  */
			float ar = re[m] + pz.x, ai = im[m] - pz.y; /* A  = Z[k] + conj Z[512-k]          = 2 E[k]      */
			float br = re[m] - pz.x, bi = im[m] + pz.y; /* B  = Z[k] - conj Z[512-k]; O2 = -i*B = 2 O[k]    */
			float tr = tw.x * bi + tw.y * br;           /* T  = W1024^k * (bi - i*br)                        */
			float ti = tw.y * bi - tw.x * br;
			float xr = ar + tr, xi = ai + ti;           /* 2 X[k]                                            */
			float yr = ar - tr, yi = ai - ti;           /* conj(2 X[512-k])                                  */
			/* |2X| by v_sqrt_f32 (1 ulp). The spectrum's scale (1/2 and the variant's normalisation) is folded
			 * into the mel weights; it is applied explicitly only where the spectrum itself is dumped. */
			slo[m] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
			shi[m] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
			if (STAGES) { flr[m] = 0.5f * xr; fli[m] = 0.5f * xi; fhr[m] = 0.5f * yr; fhi[m] = -0.5f * yi; }
		}
		/* k = 256 pairs with itself: X[256] = conj(Z[256]) (lane 0, reg 4) */
		const float s256 = 2.0f * __builtin_amdgcn_sqrtf(re[4] * re[4] + im[4] * im[4]);

		/* ---- 4. spectrum to LDS; S[513..515] were zeroed before the loop and only ever meet zero weights */
		float *S = xbuf + ED_S_OFF; /* disjoint from the transpose slots 0..525 */
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			S[k0 + 64 * m] = slo[m];
			S[512 - k0 - 64 * m] = shi[m];
		}
		if (lane == 0) S[256] = s256;
		if (STAGES)
		{
			if (args.fft)
			{
				float2 *F = reinterpret_cast<float2 *>(args.fft) + (int64_t)f * 513;
#pragma unroll
				for (int m = 0; m < 4; m++)
				{
					F[k0 + 64 * m] = make_float2(flr[m], fli[m]);
					F[512 - k0 - 64 * m] = make_float2(fhr[m], fhi[m]);
				}
				if (lane == 0) F[256] = make_float2(re[4], -im[4]);
			}
		}
		ed_wave_sync();
		if (STAGES && args.spec)
		{
			for (int k = lane; k < 513; k += 64) args.spec[(int64_t)f * 513 + k] = S[k] * spec_scale;
		}

		/* ---- 5. mel filterbank, balanced: lane (b = lane&15, r = lane>>4) sums quarter r of the narrow band b
		 *         and of the wide band 31-b; all quad reads of a part are issued before the first use */
		const float4 *S4 = reinterpret_cast<const float4 *>(S);
		float alo0 = 0.0f, alo1 = 0.0f, ahi0 = 0.0f, ahi1 = 0.0f;
#pragma unroll
		for (int t = 0; t < NLO; t++)
		{
			const float4 s = S4[mel_slo4 + t], w = melw4[t * 64 + lane];
			alo0 = fmaf(s.x, w.x, alo0); alo1 = fmaf(s.y, w.y, alo1);
			alo0 = fmaf(s.z, w.z, alo0); alo1 = fmaf(s.w, w.w, alo1);
		}
#pragma unroll
		for (int t = 0; t < NHI; t++)
		{
			/* at most three quad pairs (24 registers) in flight: bounds the register footprint of this stage */
			if (t % 3 == 0) __builtin_amdgcn_sched_barrier(0);
			const float4 s = S4[mel_shi4 + t], w = melw4[(NLO + t) * 64 + lane];
			ahi0 = fmaf(s.x, w.x, ahi0); ahi1 = fmaf(s.y, w.y, ahi1);
			ahi0 = fmaf(s.z, w.z, ahi0); ahi1 = fmaf(s.w, w.w, ahi1);
		}
		__builtin_amdgcn_sched_barrier(0);
		/* the four quarters live in the four 16-lane rows: sum them with VALU row swaps (no LDS trip) */
		const float elo = ed_sum_rows(alo0 + alo1), ehi = ed_sum_rows(ahi0 + ahi1);
		const float llo = do_log ? __logf(elo + log_offset) : elo; /* band b    */
		const float lhi = do_log ? __logf(ehi + log_offset) : ehi; /* band 31-b */
		if (STAGES && lane < 16)
		{
			if (args.mel) { args.mel[(int64_t)f * 32 + band] = elo; args.mel[(int64_t)f * 32 + 31 - band] = ehi; }
			if (args.logmel) { args.logmel[(int64_t)f * 32 + band] = llo; args.logmel[(int64_t)f * 32 + 31 - band] = lhi; }
		}

		/* ---- 6. DCT-II through cos symmetry: y[c] = sum_{n<16} D[n][c] * (L[n] + (-1)^c L[31-n]);
		 *         lane (c = lane&31, h = lane>>5) sums n = 8h..8h+7 */
		float *Lb = xbuf + ED_L_OFF; /* u[16] | v[16], 16-B aligned, behind the spectrum */
		if (lane < 16) { Lb[band] = llo + lhi; Lb[16 + band] = llo - lhi; }
		ed_wave_sync();
		const float4 *L4 = reinterpret_cast<const float4 *>(Lb + 16 * (lane & 1) + 8 * (lane >> 5));
		const float4 v0 = L4[0], v1 = L4[1], w0 = dctl[lane], w1 = dctl[64 + lane];
		float d = v0.x * w0.x, d1 = v1.x * w1.x;
		d = fmaf(v0.y, w0.y, d); d1 = fmaf(v1.y, w1.y, d1);
		d = fmaf(v0.z, w0.z, d); d1 = fmaf(v1.z, w1.z, d1);
		d = fmaf(v0.w, w0.w, d); d1 = fmaf(v1.w, w1.w, d1);
		d = ed_sum_halves(d + d1);
		ed_wave_sync(); /* Lb / S are rewritten by the next frame */

		/* ---- 7. store */
		if (lane < args.n_coef)
		{
			if (args.mfcc) args.mfcc[(int64_t)f * args.n_coef + lane] = d;
			if (args.feat)
			{
				float q = d * args.feat_scale;
				q = fminf(fmaxf(q, -128.0f), 127.0f);
				args.feat[(int64_t)f * args.n_coef + lane] = (int8_t)__float2int_rn(q);
				if (feat2) feat2[(int64_t)f * args.n_coef + lane] = (int8_t)__float2int_rn(q);
			}
		}
	}
}


#endif
