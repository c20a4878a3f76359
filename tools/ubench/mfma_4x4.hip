// check of the lane layout of v_mfma_f32_4x4x1_16b_f32 as phase 1 of sdf_tiles_span uses it: 16 blocks of
// 4 rows x 4 columns; A: lane 4 b + i holds row i of block b, B: lane 4 b + j holds column j of block b,
// D: register i of lane 4 b + j = A[b][i] * B[b][j] + C.  With A = group term of group 4 t + (lane & 3) (the same
// in every block) and B = the lane's OWN pixel term, register i of every lane is its own pixel against group 4 t + i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *ga, const float *pb, float *out)
{
	const int l = threadIdx.x;
	for (int t = 0; t < 8; t++) {
		f32x4 acc = {1000.0f, 1000.0f, 1000.0f, 1000.0f};
		acc = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[4 * t + (l & 3)], pb[l], acc, 0, 0, 0);
		for (int i = 0; i < 4; i++)
			out[(4 * t + i) * 64 + l] = acc[i];
	}
}
int main()
{
	float hg[32], hp[64], hout[32 * 64];
	for (int i = 0; i < 32; i++) hg[i] = 1.0f + i;
	for (int i = 0; i < 64; i++) hp[i] = 100.0f * (i + 1);
	float *d; hipMalloc(&d, (32 + 64 + 2048) * 4);
	hipMemcpy(d, hg, 128, hipMemcpyHostToDevice); hipMemcpy(d + 32, hp, 256, hipMemcpyHostToDevice);
	k<<<1, 64>>>(d, d + 32, d + 96);
	hipMemcpy(hout, d + 96, 8192, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int g = 0; g < 32; g++)
		for (int p = 0; p < 64; p++)
			bad += hout[g * 64 + p] != 1000.0f + hg[g] * hp[p];
	printf("4x4x1 layout: %d mismatches of 2048\n", bad);
	return bad != 0;
}
