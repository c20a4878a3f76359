// micro-benchmark: issue cost of packed f16 VALU ops and 16-bit LDS reads (candidates for the group-bound pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, unsigned a, unsigned b)
{
	__shared__ unsigned short tab[4096];
	for (int i = threadIdx.x; i < 4096; i += 256)
		tab[i] = (unsigned short)(i * 7);
	__syncthreads();
	unsigned u0 = threadIdx.x | 0x3c003c00u, u1 = u0 + 3, u2 = u0 + 5, u3 = u0 + 7;
	unsigned addr = (threadIdx.x & 63) * 2;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) { REP8(asm volatile("v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
		if (MODE == 1) { REP8(asm volatile("v_pk_min_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_min_f16 %2, %2, %4\n v_pk_min_f16 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
		if (MODE == 2) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a), "v"(b));) }
		if (MODE == 3) { REP8(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_mul_f16 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
		if (MODE == 4) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a), "v"(b));) }
		if (MODE == 5) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
		if (MODE == 6) { REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %4\n v_cvt_pkrtz_f16_f32 %1, %1, %4\n v_cvt_pkrtz_f16_f32 %2, %2, %4\n v_cvt_pkrtz_f16_f32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
		if (MODE == 7) { // 16-bit LDS reads into the two halves of a register + one packed add (the table form of D^2)
			REP8(asm volatile("ds_read_u16_d16 %0, %4\n ds_read_u16_d16_hi %0, %4 offset:128\n ds_read_u16_d16 %1, %4 offset:256\n ds_read_u16_d16_hi %1, %4 offset:384\n s_waitcnt lgkmcnt(0)\n v_pk_add_f16 %2, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2) : "v"(u3), "v"(addr));)
		}
		if (MODE == 8) { REP8(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a), "v"(b));) }
		if (MODE == 9) { REP8(asm volatile("v_pk_max_f16 %0, %0, %4\n v_pk_max_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %4\n v_pk_max_f16 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = (float)(u0 + u1 + u2 + u3);
}
int main()
{
	float *d; hipMalloc(&d, 256 * 4096 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000, wg_per_cu = 8, grid = 256 * wg_per_cu;
	const char *names[] = {"v_pk_add_f16", "v_pk_min_f16", "v_pk_fma_f16", "v_pk_mul_f16", "v_and_or_b32", "v_lshl_or_b32", "v_cvt_pkrtz_f16_f32", "4x ds_read_u16_d16 + pk_add (per 6 instrs)", "v_bfi_b32", "v_pk_max_f16"};
	for (int mode = 0; mode < 10; mode++) {
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
#define L(M) if (mode == M) k<M><<<grid, 256>>>(d, iters, 0x3c003c00u, 0x38003800u);
			L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9)
			hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
		}
		const double n = mode == 7 ? 8 : 32; // instruction groups per iteration
		printf("%-44s %.3f ms  %.2f cycles per wave-instr%s per SIMD @2.4GHz (8 waves/SIMD)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ((double)iters * n * wg_per_cu), mode == 7 ? " GROUP" : "");
	}
	return 0;
}
