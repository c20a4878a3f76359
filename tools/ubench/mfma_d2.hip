// check of the f32 MFMA form of the group-bound distances (phase 1 of sdf_tiles_span):
// D2[g][p] = c_g + pp_p - 2 ax_g x_p - 2 ay_g y_p for 32 groups x 32 pixels from two v_mfma_f32_32x32x2_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *ax, const float *ay, const float *px, const float *py, float *out)
{
	const int l = threadIdx.x, r = l & 31, h = l >> 5;
	const float c = ax[r] * ax[r] + ay[r] * ay[r];
	const float pp = px[r] * px[r] + py[r] * py[r];
	// A[i][k]: lane holds row i = r, k = h.  B[k][j]: lane holds col j = r, k = h.
	const float aC = h == 0 ? c : 1.0f, bC = h == 0 ? 1.0f : pp;            // c_g * 1 + 1 * pp_p
	const float aXY = h == 0 ? -2.0f * ax[r] : -2.0f * ay[r], bXY = h == 0 ? px[r] : py[r];
	f32x16 acc = {0};
	acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aC, bC, acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aXY, bXY, acc, 0, 0, 0);
	for (int reg = 0; reg < 16; reg++) {
		const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h; // group
		out[row * 32 + r] = acc[reg];                        // col = pixel r
	}
}
int main()
{
	float hax[32], hay[32], hpx[32], hpy[32], hout[1024];
	for (int i = 0; i < 32; i++) {
		hax[i] = 0.37f * i - 5.25f; hay[i] = 11.0f - 0.73f * i; hpx[i] = i - 15.5f; hpy[i] = 0.5f * i - 7.5f;
	}
	float *d; hipMalloc(&d, (128 + 1024) * 4);
	hipMemcpy(d, hax, 128, hipMemcpyHostToDevice); hipMemcpy(d + 32, hay, 128, hipMemcpyHostToDevice);
	hipMemcpy(d + 64, hpx, 128, hipMemcpyHostToDevice); hipMemcpy(d + 96, hpy, 128, hipMemcpyHostToDevice);
	k<<<1, 64>>>(d, d + 32, d + 64, d + 96, d + 128);
	hipMemcpy(hout, d + 128, 4096, hipMemcpyDeviceToHost);
	double worst = 0;
	for (int g = 0; g < 32; g++)
		for (int p = 0; p < 32; p++) {
			const double want = (hpx[p] - (double)hax[g]) * (hpx[p] - (double)hax[g]) + (hpy[p] - (double)hay[g]) * (hpy[p] - (double)hay[g]);
			worst = fmax(worst, fabs(hout[g * 32 + p] - want));
		}
	printf("max |mfma D2 - exact D2| = %.3g (values up to ~1000)\n", worst);
	return worst < 1e-3 ? 0 : 1;
}
