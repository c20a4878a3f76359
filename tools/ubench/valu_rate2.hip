// micro-benchmark: per-instruction issue cost of the ops used in the filter loop (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, unsigned m)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
	unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
	for (int i = 0; i < iters; i++) {
		// 4 independent chains, 8 repeats = 32 instrs per iteration
		if (MODE == 0) { REP8(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m));) }
		if (MODE == 1) { REP8(asm volatile("v_med3_u32 %0, %0, %4, %5\n v_med3_u32 %1, %1, %4, %5\n v_med3_u32 %2, %2, %4, %5\n v_med3_u32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "v"(u0 ^ m));) }
		if (MODE == 2) { REP8(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));) }
		if (MODE == 3) { REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
		if (MODE == 4) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "s"(i));) }
		if (MODE == 5) { REP8(asm volatile("v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));) }
		if (MODE == 6) { REP8(asm volatile("v_fma_f32 %0, -%0, %4, %5\n v_fma_f32 %1, -%1, %4, %5\n v_fma_f32 %2, -%2, %4, %5\n v_fma_f32 %3, -%3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
		if (MODE == 7) { REP8(asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m));) }
		if (MODE == 8) { REP8(asm volatile("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_min3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "v"(u0 ^ m));) }
		if (MODE == 9) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m) : "vcc");) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3;
}
int main()
{
	float *d; hipMalloc(&d, 256 * 4096 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4000, wg_per_cu = 8, grid = 256 * wg_per_cu;
	const char *names[] = {"v_min_u32", "v_med3_u32", "v_min_f32", "v_med3_f32", "v_and_or_b32(sgpr)", "v_sub_f32", "v_fma_f32", "v_max_u32", "v_min3_u32", "cmp+cndmask"};
	for (int mode = 0; mode < 10; mode++) {
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
#define L(M) if (mode == M) k<M><<<grid, 256>>>(d, iters, 1.0001f, 0.5f, 12345u);
			L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9)
			hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
		}
		printf("%-20s %.3f ms  %.2f cycles per wave-instr per SIMD @2.4GHz (8 waves/SIMD)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wg_per_cu));
	}
	return 0;
}
