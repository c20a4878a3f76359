// micro-benchmark: issue cost of the ops of the bounded-group kernel's staging / bounds / decide code (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, unsigned m, double da, double db)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
	unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
	double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) { REP8(asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %4, 31" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m));) }
		if (MODE == 1) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da));) }
		if (MODE == 2) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));) }
		if (MODE == 3) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(da));) }
		if (MODE == 4) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
		if (MODE == 5) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
		if (MODE == 6) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
		if (MODE == 7) { REP8(asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da));) }
		if (MODE == 8) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f64 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(d0), "+v"(u1), "+v"(d2), "+v"(u3) : "v"(da), "v"(m) : "vcc");) }
		if (MODE == 9) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
		if (MODE == 10) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));) }
		if (MODE == 11) { REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
		if (MODE == 12) { REP8(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
		if (MODE == 13) { REP8(asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
		if (MODE == 14) { REP8(asm volatile("v_ceil_f64 %0, %0\n v_ceil_f64 %1, %1\n v_ceil_f64 %2, %2\n v_ceil_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
		if (MODE == 15) { REP8(asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_fmas_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da) : "vcc");) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3 + (float)(d0 + d1 + d2 + d3);
}
int main()
{
	float *d; hipMalloc(&d, 256 * 4096 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000, wg_per_cu = 8, grid = 256 * wg_per_cu;
	const char *names[] = {"v_alignbit_b32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f32", "v_sqrt_f32", "v_cvt_f32_f64", "v_max_f64", "cmp_f64+cndmask", "v_rcp_f64", "v_mul_f32", "v_floor_f32", "v_cvt_f64_i32", "v_ffbl_b32", "v_ceil_f64", "div_scale/fmas/fixup f64"};
	for (int mode = 0; mode < 16; mode++) {
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
#define L(M) if (mode == M) k<M><<<grid, 256>>>(d, iters, 1.0001f, 0.5f, 12345u, 1.0000001, 0.9999999);
			L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
			hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
		}
		printf("%-26s %.3f ms  %.2f cycles per wave-instr per SIMD @2.4GHz (8 waves/SIMD)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wg_per_cu));
	}
	return 0;
}
