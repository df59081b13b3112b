/*
 * cnn_mfma_kernels.hip -- the int8 keyword-spotting CNN on the gfx950 matrix cores (fast path).
 *
 * Same arithmetic, bit for bit, as the reference's NNoM/CMSIS-NN CPU path (see cnn_kernels.hip for the
 * reference citations per layer); only the order of the exact int32 accumulation differs, which integer
 * addition does not notice. Requantisation is out = relu(ssat8((acc + (bias << bl) + round) >> rs)); max-pool
 * is applied to the int32 accumulators BEFORE requantisation, which is exact because the requantisation is a
 * monotone non-decreasing function of the accumulator.
 *
 * Every layer is the GEMM  D[out_channel][pixel] = sum_k A[out_channel][k] * B[k][pixel]  on
 * v_mfma_i32_32x32x32_i8 (A, B: 16 bytes per lane; lane l: A[row l&31][k 16*(l>>5)+j], B[k 16*(l>>5)+j][col l&31];
 * D: col l&31, row (reg&3) + 8*(reg>>2) + 4*(l>>5) -- verified with exact integer data, tools/ubench/mfma_i8.hip):
 *   - A = weights, pre-packed on the host as operand fragments (ed_cnn_mfma_model_t), staged once per
 *     workgroup in LDS, fetched with one conflict-free ds_read_b128 per MFMA;
 *   - B = activations straight from the HWC int8 buffers in LDS: the 16 bytes a lane needs are 16 consecutive
 *     input channels of one tap (conv2-4) or one 16-byte-padded input row (conv1, Toeplitz form), i.e. one
 *     aligned ds_read_b128 -- no im2col buffer;
 *   - D puts 4 consecutive output channels of one pixel into 4 consecutive registers of a lane, so the
 *     epilogue packs them into one dword and stores HWC int8 directly where the next layer reads.
 *   - the two rows of a max-pool window are computed as two accumulator tiles over the same lanes
 *     (even / odd input row), pooled by an element-wise max.
 *
 * One 512-thread workgroup (8 waves) processes 32 utterances per iteration; tiles of a layer are dealt
 * round-robin to the waves and layers are separated by workgroup barriers. LDS: 61 KB of weight fragments +
 * 32 x 2992 B of activations (two aliased regions per utterance) = 157 KB -> one workgroup per CU.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edison_internal.h"

#define EDM_U 32
#define EDM_WAVES 8
#define EDM_THREADS (64 * EDM_WAVES)
#define EDM_REGA 1120 /* in' [31][16] (496)  ->  p2 [5][7][32] (1120)  ->  c4 [3][32] (96)   */
#define EDM_REGB 1872 /* p1 [13][9][16] (1872)  ->  c3 [3][5][64] (960)                        */
#define EDM_UTT (EDM_REGA + EDM_REGB)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i edm_ld16(const unsigned char *p) { return *reinterpret_cast<const v4i *>(p); }

__device__ __forceinline__ int edm_med3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* requantise 4 consecutive accumulators (+ their seeds) with ReLU and pack them into one HWC dword */
__device__ __forceinline__ uint32_t edm_pack_relu(int a0, int a1, int a2, int a3, const int32_t *seed, int rs)
{
	const uint32_t b0 = (uint32_t)edm_med3((a0 + seed[0]) >> rs, 0, 127);
	const uint32_t b1 = (uint32_t)edm_med3((a1 + seed[1]) >> rs, 0, 127);
	const uint32_t b2 = (uint32_t)edm_med3((a2 + seed[2]) >> rs, 0, 127);
	const uint32_t b3 = (uint32_t)edm_med3((a3 + seed[3]) >> rs, 0, 127);
	return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

__device__ __forceinline__ int edm_max(int a, int b) { return a > b ? a : b; }

__global__ __launch_bounds__(EDM_THREADS) void ed_cnn_mfma_kernel(const ed_cnn_mfma_model_t *__restrict__ model,
                                                                 const int8_t *__restrict__ feat, int64_t n_utt,
                                                                 int64_t feat_stride, int8_t *__restrict__ logits,
                                                                 int8_t *__restrict__ softmax,
                                                                 int32_t *__restrict__ argmax)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const ed_cnn_mfma_model_t &M = *reinterpret_cast<const ed_cnn_mfma_model_t *>(smem);
	unsigned char *acts = smem + sizeof(ed_cnn_mfma_model_t);

	{ /* stage the weight fragments once per workgroup */
		const v4i *src = reinterpret_cast<const v4i *>(model);
		v4i *dst = reinterpret_cast<v4i *>(smem);
		for (int i = threadIdx.x; i < (int)(sizeof(ed_cnn_mfma_model_t) / 16); i += EDM_THREADS) dst[i] = src[i];
	}
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int col = lane & 31, h = lane >> 5;
	const unsigned char *afrag = reinterpret_cast<const unsigned char *>(&M) + lane * 16;
	const int a1_off = (int)offsetof(ed_cnn_mfma_model_t, a1), a2_off = (int)offsetof(ed_cnn_mfma_model_t, a2);
	const int a3_off = (int)offsetof(ed_cnn_mfma_model_t, a3), a4_off = (int)offsetof(ed_cnn_mfma_model_t, a4);
	const int afc_off = (int)offsetof(ed_cnn_mfma_model_t, afc);

	for (int64_t base = (int64_t)blockIdx.x * EDM_U; base < n_utt; base += (int64_t)gridDim.x * EDM_U)
	{
		const int nb = (int)((n_utt - base) < EDM_U ? (n_utt - base) : EDM_U);
		__syncthreads(); /* previous iteration's dense stage has finished reading region A */

		/* ---- input: feat[u][31][13] -> in'[u][31][16] (3 zero bytes of padding per row) */
		for (int r = threadIdx.x; r < EDM_U * ED_IN_H; r += EDM_THREADS)
		{
			const int u = r / ED_IN_H, y = r - u * ED_IN_H;
			uint32_t d[4] = {0, 0, 0, 0};
			if (u < nb)
			{
				const uint8_t *g = reinterpret_cast<const uint8_t *>(feat) + (base + u) * feat_stride + y * ED_IN_W;
#pragma unroll
				for (int j = 0; j < ED_IN_W; j++) d[j >> 2] |= (uint32_t)g[j] << (8 * (j & 3));
			}
			*reinterpret_cast<uint4 *>(acts + u * EDM_UTT + y * 16) = make_uint4(d[0], d[1], d[2], d[3]);
		}
		__syncthreads();

		/* ---- conv1 5x5x1->16 + ReLU + pool(2,1): Toeplitz GEMM, 144 rows (x,o) x 80 k (5 padded input rows).
		 *      columns = (utt, pooled row py): 32 x 13 = 13 column tiles; two accumulators = input rows 2py / 2py+1 */
		for (int t = wave; t < 13; t += EDM_WAVES)
		{
			const int q = t * 32 + col, u = q / 13, py = q - u * 13;
			const unsigned char *inb = acts + u * EDM_UTT + (2 * py) * 16;
			v4i be[3], bo[3];
#pragma unroll
			for (int s = 0; s < 3; s++)
			{
				const int c = (2 * s + h) < 4 ? (2 * s + h) : 4; /* k-chunk = input row y + c; chunk 5 meets zero weights */
				be[s] = edm_ld16(inb + c * 16);
				bo[s] = edm_ld16(inb + (c + 1) * 16);
			}
			unsigned char *p1 = acts + u * EDM_UTT + EDM_REGA + (py * 9) * 16;
#pragma unroll 1
			for (int rt = 0; rt < 5; rt++)
			{
				v16i ae = {0}, ao = {0};
#pragma unroll
				for (int s = 0; s < 3; s++)
				{
					const v4i a = edm_ld16(afrag + a1_off + (rt * 3 + s) * 1024);
					ae = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, be[s], ae, 0, 0, 0);
					ao = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bo[s], ao, 0, 0, 0);
				}
#pragma unroll
				for (int g = 0; g < 4; g++)
				{
					const int x = 2 * rt + (g >> 1), o0 = 4 * h + 8 * (g & 1);
					if (x < 9)
						*reinterpret_cast<uint32_t *>(p1 + x * 16 + o0) =
						    edm_pack_relu(edm_max(ae[4 * g], ao[4 * g]), edm_max(ae[4 * g + 1], ao[4 * g + 1]),
						                  edm_max(ae[4 * g + 2], ao[4 * g + 2]), edm_max(ae[4 * g + 3], ao[4 * g + 3]),
						                  &M.b1[o0], M.rs1);
				}
			}
		}
		__syncthreads();

		/* ---- conv2 3x3x16->32 + ReLU + pool(2,1): K = 9 taps x 16 ch (5 k-steps of 2 taps); columns =
		 *      (utt, py, x): 32 x 35 = 35 column tiles; two accumulators = conv rows 2py / 2py+1 */
		for (int t = wave; t < 35; t += EDM_WAVES)
		{
			const int q = t * 32 + col, u = q / 35, r = q - u * 35, py = r / 7, x = r - py * 7;
			const unsigned char *p1 = acts + u * EDM_UTT + EDM_REGA + ((2 * py) * 9 + x) * 16;
			v16i ae = {0}, ao = {0};
#pragma unroll
			for (int s = 0; s < 5; s++)
			{
				const int tap = (2 * s + h) < 8 ? (2 * s + h) : 8; /* tap 9 meets zero weights */
				const int ky = tap / 3, kx = tap - 3 * ky;
				const v4i a = edm_ld16(afrag + a2_off + s * 1024);
				const v4i b0 = edm_ld16(p1 + (ky * 9 + kx) * 16);
				const v4i b1 = edm_ld16(p1 + ((ky + 1) * 9 + kx) * 16);
				ae = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, ae, 0, 0, 0);
				ao = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, ao, 0, 0, 0);
			}
			unsigned char *p2 = acts + u * EDM_UTT + (py * 7 + x) * 32;
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const int o0 = 8 * g + 4 * h;
				*reinterpret_cast<uint32_t *>(p2 + o0) =
				    edm_pack_relu(edm_max(ae[4 * g], ao[4 * g]), edm_max(ae[4 * g + 1], ao[4 * g + 1]),
				                  edm_max(ae[4 * g + 2], ao[4 * g + 2]), edm_max(ae[4 * g + 3], ao[4 * g + 3]), &M.b2[o0],
				                  M.rs2);
			}
		}
		__syncthreads();

		/* ---- conv3 3x3x32->64 + ReLU: 9 k-steps (tap, 16-channel half); columns = (utt, y, x): 32 x 15 = 15
		 *      column tiles; two accumulators = output channels 0-31 / 32-63 */
		for (int t = wave; t < 15; t += EDM_WAVES)
		{
			const int q = t * 32 + col, u = q / 15, r = q - u * 15, y = r / 5, x = r - y * 5;
			const unsigned char *p2 = acts + u * EDM_UTT + (y * 7 + x) * 32 + 16 * h;
			v16i a0 = {0}, a1 = {0};
#pragma unroll
			for (int s = 0; s < 9; s++)
			{
				const int ky = s / 3, kx = s - 3 * ky;
				const v4i b = edm_ld16(p2 + (ky * 7 + kx) * 32);
				a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + a3_off + s * 1024), b, a0, 0, 0, 0);
				a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + a3_off + (9 + s) * 1024), b, a1, 0, 0, 0);
			}
			unsigned char *c3 = acts + u * EDM_UTT + EDM_REGA + (y * 5 + x) * 64;
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const int o0 = 8 * g + 4 * h;
				*reinterpret_cast<uint32_t *>(c3 + o0) =
				    edm_pack_relu(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3], &M.b3[o0], M.rs3);
				*reinterpret_cast<uint32_t *>(c3 + 32 + o0) =
				    edm_pack_relu(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3], &M.b3[32 + o0], M.rs3);
			}
		}
		__syncthreads();

		/* ---- conv4 3x3x64->32 + ReLU: 18 k-steps (tap, 32-channel half, 16-channel lane half); columns =
		 *      (utt, x): 32 x 3 = 3 column tiles */
		for (int t = wave; t < 3; t += EDM_WAVES)
		{
			const int q = t * 32 + col, u = q / 3, x = q - u * 3;
			const unsigned char *c3 = acts + u * EDM_UTT + EDM_REGA + x * 64 + 16 * h;
			v16i acc = {0};
#pragma unroll
			for (int s = 0; s < 18; s++)
			{
				const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
				const v4i b = edm_ld16(c3 + (ky * 5 + kx) * 64 + 32 * (s & 1));
				acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + a4_off + s * 1024), b, acc, 0, 0, 0);
			}
			unsigned char *c4 = acts + u * EDM_UTT + x * 32;
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const int o0 = 8 * g + 4 * h;
				*reinterpret_cast<uint32_t *>(c4 + o0) =
				    edm_pack_relu(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], &M.b4[o0], M.rs4);
			}
		}
		__syncthreads();

		/* ---- dense 96->10, softmax, argmax: one column tile (column = utterance), wave 0 */
		if (wave == 0)
		{
			const unsigned char *c4 = acts + col * EDM_UTT + 16 * h;
			v16i acc = {0};
#pragma unroll
			for (int s = 0; s < 3; s++)
				acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + afc_off + s * 1024), edm_ld16(c4 + 32 * s), acc, 0, 0, 0);
			/* lane (utt, h) holds rows 4h..4h+3 in regs 0-3 and rows 8+4h.. in regs 4-7 (only rows 8, 9 exist) */
			int lg[10];
			int mine[4], hi2[2];
#pragma unroll
			for (int i = 0; i < 4; i++)
			{
				const int v = (acc[i] + M.bfc[4 * h + i]) >> M.rsfc;
				mine[i] = edm_med3(v, -128, 127);
			}
#pragma unroll
			for (int i = 0; i < 2; i++) hi2[i] = edm_med3((acc[4 + i] + M.bfc[8 + i]) >> M.rsfc, -128, 127);
#pragma unroll
			for (int i = 0; i < 4; i++)
			{
				const int other = __shfl(mine[i], col + 32); /* rows 4..7 live in the upper half-wave */
				lg[i] = mine[i];
				lg[4 + i] = other;
			}
			lg[8] = hi2[0]; lg[9] = hi2[1];
			if (h == 0 && col < nb)
			{
				/* arm_softmax_q7 (portable branch) and nnom_predict's first-maximum rule */
				int mx = -128;
#pragma unroll
				for (int i = 0; i < 10; i++) mx = lg[i] > mx ? lg[i] : mx;
				const int sbase = mx - 8;
				int sum = 0;
#pragma unroll
				for (int i = 0; i < 10; i++) sum += 1 << edm_med3(lg[i] - sbase, 0, 7);
				const int output_base = (1 << 20) / sum;
				int best = 0, bv = -129;
				uint32_t lw[3] = {0, 0, 0}, sw[3] = {0, 0, 0};
#pragma unroll
				for (int i = 0; i < 10; i++)
				{
					const int v = edm_med3(output_base >> edm_med3(13 + sbase - lg[i], 0, 31), -128, 127);
					if (v > bv) { bv = v; best = i; }
					lw[i >> 2] |= (uint32_t)(uint8_t)lg[i] << (8 * (i & 3));
					sw[i >> 2] |= (uint32_t)(uint8_t)v << (8 * (i & 3));
				}
				const int64_t uo = (base + col) * ED_FC_O; /* 10-byte records: 2-byte aligned */
				if (logits)
				{
					uint16_t *p = reinterpret_cast<uint16_t *>(logits + uo);
					p[0] = (uint16_t)lw[0]; p[1] = (uint16_t)(lw[0] >> 16); p[2] = (uint16_t)lw[1];
					p[3] = (uint16_t)(lw[1] >> 16); p[4] = (uint16_t)lw[2];
				}
				if (softmax)
				{
					uint16_t *p = reinterpret_cast<uint16_t *>(softmax + uo);
					p[0] = (uint16_t)sw[0]; p[1] = (uint16_t)(sw[0] >> 16); p[2] = (uint16_t)sw[1];
					p[3] = (uint16_t)(sw[1] >> 16); p[4] = (uint16_t)sw[2];
				}
				if (argmax) argmax[base + col] = best;
			}
		}
	}
}

static int g_cnn_mfma_ready = 0;

/* feat_stride = bytes between consecutive utterances' feature maps: 403 for packed utterances, 13 for the
 * sliding windows of a stream (window i = feature rows i..i+30 of one long [rows][13] buffer). */
extern "C" int ed_launch_cnn_mfma(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                  int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                  hipStream_t stream)
{
	if (n_utt <= 0) return 0;
	const size_t lds = sizeof(ed_cnn_mfma_model_t) + (size_t)EDM_U * EDM_UTT;
	if (!g_cnn_mfma_ready)
	{
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ed_cnn_mfma_kernel),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return (int)e;
		g_cnn_mfma_ready = 1;
	}
	int64_t blocks = (n_utt + EDM_U - 1) / EDM_U;
	if (blocks > n_cu) blocks = n_cu; /* 157 KB of LDS: one workgroup per CU */
	hipLaunchKernelGGL(ed_cnn_mfma_kernel, dim3((unsigned)blocks), dim3(EDM_THREADS), lds, stream, dev_model, feat, n_utt,
	                   feat_stride, logits, softmax, argmax);
	return (int)hipGetLastError();
}
