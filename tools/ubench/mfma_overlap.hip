// micro-benchmark: do f32 MFMA and f32 VALU work of the waves of a SIMD overlap?  Per iteration and wave:
// 48 VALU fma and / or matrix work of 12288 MACs in one of three shapes: 48 x 4x4x1 (2 passes each), 12 x 16x16x4
// (8 passes), 6 x 32x32x2 (16 passes).  mode 0: VALU only; 1: MFMA only; 2: both, interleaved in program order;
// 3: both, one phase after the other.  4 waves per SIMD (256 x 4 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int SHAPE> __global__ __launch_bounds__(256, 4) void k(float *out, int iters, float a, float b)
{
	float v[8];
	f32x4 acc[8];
	f32x16 big[2];
	for (int i = 0; i < 8; i++) { v[i] = threadIdx.x + i; acc[i] = (f32x4){a, b, a, b}; }
	for (int i = 0; i < 16; i++) { big[0][i] = a; big[1][i] = b; }
	auto valu = [&](int r) {
#pragma unroll
		for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], a, b);
		(void)r;
	};
	auto mfma = [&](int r) { // one sixth of the matrix work
		if (SHAPE == 0) {
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
		} else if (SHAPE == 1) {
			acc[(2 * r) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(2 * r) & 7], 0, 0, 0);
			acc[(2 * r + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(2 * r + 1) & 7], 0, 0, 0);
		} else if (SHAPE == 2) {
			big[r & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[r & 1], 0, 0, 0);
		} else {
			const f16x8 ha = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b};
			big[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, big[r & 1], 0, 0, 0);
		}
	};
	for (int it = 0; it < iters; it++) {
		if (MODE == 0 || MODE == 3) {
#pragma unroll
			for (int r = 0; r < 6; r++) valu(r);
		}
		if (MODE == 1 || MODE == 3) {
#pragma unroll
			for (int r = 0; r < 6; r++) mfma(r);
		}
		if (MODE == 2) {
#pragma unroll
			for (int r = 0; r < 6; r++) { mfma(r); valu(r); }
		}
		asm volatile("" ::: "memory");
	}
	float s = 0;
	for (int i = 0; i < 8; i++) s += v[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	for (int i = 0; i < 16; i++) s += big[0][i] + big[1][i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
	float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int iters = 2000, wg_per_cu = 8, grid = 256 * wg_per_cu;
	const char *modes[] = {"48 fma", "mfma only", "mfma + 48 fma interleaved", "48 fma then mfma"};
	const char *shapes[] = {"48 x 4x4x1", "12 x 16x16x4", "6 x 32x32x2", "6 x 32x32x16 f16"};
	for (int shape = 0; shape < 4; shape++)
		for (int mode = 0; mode < 4; mode++) {
			float ms = 0;
			for (int rep = 0; rep < 2; rep++) {
				(void)hipEventRecord(e0);
#define L(M, S) if (mode == M && shape == S) k<M, S><<<grid, 256>>>(d, iters, 0.999f, 0.001f);
				L(0, 0) L(1, 0) L(2, 0) L(3, 0) L(0, 1) L(1, 1) L(2, 1) L(3, 1) L(0, 2) L(1, 2) L(2, 2) L(3, 2) L(0, 3) L(1, 3) L(2, 3) L(3, 3)
				(void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
			}
			printf("%-17s %-28s %.3f ms  %.1f cycles per wave-iteration per SIMD @2.4GHz\n", shapes[shape], modes[mode], ms, ms * 1e-3 * 2.4e9 / ((double)iters * wg_per_cu));
		}
	return 0;
}
