// Verify the operand / result lane maps of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 on gfx950 with exact
// integer data (asymmetric random A and B).   hipcc --offload-arch=gfx950 mfma_i8.hip -o mfma_i8 && ./mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// assumed: lane l holds A[row = l&31][k = 16*(l>>5) + j], B[k = 16*(l>>5) + j][col = l&31], j = 0..15 (16 bytes)
__global__ void k32(const int8_t *A, const int8_t *B, int *D)
{
	const int l = threadIdx.x;
	v4i a = *reinterpret_cast<const v4i *>(A + (l & 31) * 32 + 16 * (l >> 5));  // A row-major [32][32]
	int8_t bb[16];
	for (int j = 0; j < 16; j++) bb[j] = B[(16 * (l >> 5) + j) * 32 + (l & 31)]; // B row-major [k][n]
	v4i b; memcpy(&b, bb, 16);
	v16i c = {0};
	c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
	// assumed C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
	for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// assumed: lane l holds A[row = l&15][k = 16*(l>>4) + j], B[k = 16*(l>>4)+j][col = l&15]; C: col = l&15, row = 4*(l>>4) + reg
__global__ void k16(const int8_t *A, const int8_t *B, int *D)
{
	const int l = threadIdx.x;
	v4i a = *reinterpret_cast<const v4i *>(A + (l & 15) * 64 + 16 * (l >> 4));  // A [16][64]
	int8_t bb[16];
	for (int j = 0; j < 16; j++) bb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)]; // B [64][16]
	v4i b; memcpy(&b, bb, 16);
	v4i c = {0};
	c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; r++) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main()
{
	srand(5);
	int8_t hA[32 * 64], hB[64 * 32]; int hD[32 * 32], ref[32 * 32];
	for (auto &v : hA) v = (int8_t)(rand() % 256 - 128);
	for (auto &v : hB) v = (int8_t)(rand() % 256 - 128);
	int8_t *dA, *dB; int *dD;
	hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
	hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
	// 32x32x32
	k32<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; bad += (s != hD[i * 32 + j]); }
	printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024\n", bad);
	// 16x16x64
	k16<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 16 * 16 * 4, hipMemcpyDeviceToHost);
	bad = 0;
	for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += hA[i * 64 + k] * hB[k * 16 + j]; bad += (s != hD[i * 16 + j]); }
	printf("mfma_i32_16x16x64_i8 layout check: %d mismatches of 256\n", bad);
	return 0;
}
