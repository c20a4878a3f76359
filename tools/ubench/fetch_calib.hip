// calibration of rocprofv3 FETCH_SIZE for 8-byte-per-lane coalesced loads (our staging pattern)
// and 16-byte-per-lane loads: each kernel reads exactly BYTES once.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double *p, size_t n, double *out) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; double s = 0; for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 12345.678) out[0] = s; }
__global__ void read16(const double2 *p, size_t n, double *out) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; double s = 0; for (; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; } if (s == 12345.678) out[0] = s; }
int main() {
	const size_t bytes = 1ull << 30; // 1 GiB > Infinity Cache
	double *d, *o; hipMalloc(&d, bytes); hipMalloc(&o, 8); hipMemset(d, 0, bytes);
	for (int r = 0; r < 2; r++) { read8<<<4096, 256>>>(d, bytes / 8, o); read16<<<4096, 256>>>((double2 *)d, bytes / 16, o); }
	hipDeviceSynchronize(); printf("read %zu bytes per kernel\n", bytes); return 0;
}
