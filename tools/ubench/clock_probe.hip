// clock_probe: the shader clock the chip actually holds while a VALU-bound kernel runs, and the issue cost of the two
// instruction classes the raster kernel is made of at THAT clock (round-2's table, profiles/r2_ubench.txt, converted its
// times with an assumed 2.4 GHz).  In-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, "DVFS
// give-back" item 6), stamped once around the loop, median over waves, after ~1 s of back-to-back launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters, float a, float b)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 11, u5 = u0 * 13, u6 = u0 * 17, u7 = u0 * 19;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			if (MODE == 0) { // 8 independent v_fma_f32
				x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
				x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
			} else { // 8 independent v_min_u32 (the integer / min / select class)
				const unsigned key = u7 + (unsigned)(r + i);
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(u0) : "v"(key)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u1) : "v"(key));
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(u2) : "v"(key)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u3) : "v"(key));
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(u4) : "v"(key)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u5) : "v"(key));
				asm volatile("v_min_u32 %0, %0, %1" : "+v"(u6) : "v"(key)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(u7) : "v"(u0));
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
	if ((threadIdx.x & 63) == 0) {
		const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
		stamps[2 * w] = t1 - t0;
		stamps[2 * w + 1] = r1 - r0;
	}
}

int main()
{
	const int wg_per_cu = 8, grid = 256 * wg_per_cu, iters = 4000; // 8 waves per SIMD
	float *d; hipMalloc(&d, sizeof(float) * 256 * grid);
	unsigned long long *ds; hipMalloc(&ds, sizeof(unsigned long long) * 2 * 4 * grid);
	std::vector<unsigned long long> hs(2 * 4 * grid);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int mode = 0; mode < 2; mode++) {
		float ms = 0, total = 0;
		for (int rep = 0; total < 1000.0f || rep < 3; rep++) { // ~1 s of back-to-back launches, the last one is reported
			hipEventRecord(e0);
			if (mode == 0) k<0><<<grid, 256>>>(d, ds, iters, 1.0001f, 0.5f);
			else k<1><<<grid, 256>>>(d, ds, iters, 1.0001f, 0.5f);
			hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
			total += ms;
		}
		hipMemcpy(hs.data(), ds, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost);
		std::vector<double> mhz;
		for (size_t w = 0; w < hs.size() / 2; w++)
			if (hs[2 * w + 1])
				mhz.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 100.0);
		std::sort(mhz.begin(), mhz.end());
		const double clock = mhz[mhz.size() / 2] * 1e6;
		const double instr_per_wave = (double)iters * 32, waves_per_simd = wg_per_cu;
		const double sec_per_instr = ms * 1e-3 / (instr_per_wave * waves_per_simd);
		std::printf("%-10s %.3f ms per launch; in-kernel shader clock: median %.0f MHz (min %.0f, max %.0f over %zu waves); "
		            "%.2f cycles per wave64 instruction per SIMD at that clock (%.2f if 2.4 GHz were assumed)\n",
		            mode == 0 ? "v_fma_f32" : "v_min_u32", ms, clock / 1e6, mhz.front(), mhz.back(), mhz.size(), sec_per_instr * clock,
		            sec_per_instr * 2.4e9);
	}
	return 0;
}
