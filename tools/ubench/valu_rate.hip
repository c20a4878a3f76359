// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_med3_u32/v_min_u32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
	unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = 11 * u0;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) { // 8 independent scalar FMAs
#pragma unroll
			for (int r = 0; r < 4; r++) {
				x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
				x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
			}
		} else if (MODE == 1) { // 4 independent packed FMAs (same flops as 8 scalar)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
				asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
				asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
				asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
			}
		} else if (MODE == 2) { // integer min/med3 mix like the top-4 insert
#pragma unroll
			for (int r = 0; r < 8; r++) {
				unsigned key = u4 + r + i;
				asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(u3) : "v"(u2), "v"(key));
				asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(u2) : "v"(u1), "v"(key));
				asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(u1) : "v"(u0), "v"(key));
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(u0) : "v"(key));
			}
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + u0 + u1 + u2 + u3;
}
int main()
{
	float *d; hipMalloc(&d, 256 * 8192 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4000;
	for (int wg_per_cu : {1, 2, 4, 8}) {
		int grid = 256 * wg_per_cu;
		for (int mode = 0; mode < 3; mode++) {
			float ms = 0;
			for (int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0);
				if (mode == 0) k<0><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
				if (mode == 1) k<1><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
				if (mode == 2) k<2><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
				hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
			}
			double instr_per_wave = (double)iters * (mode == 1 ? 16 : 32); // mode 1 issues 16 packed instrs (= 32 FMAs/lane)
			double waves_per_simd = wg_per_cu; // 4 waves per WG over 4 SIMDs
			double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
			printf("waves/SIMD %d mode %d (%s): %.3f ms, %.2f cycles per wave-instr per SIMD @2.4GHz\n", wg_per_cu, mode,
			       mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32" : "med3/min u32", ms, cyc);
		}
	}
	return 0;
}
