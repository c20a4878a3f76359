// sdf_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4) for the SDF raster.
//
// What it computes, per glyph and per output pixel (bit-exact with the reference):
//   renderer_precise()              /root/reference/src/render/renderer_precise.rs:8-84
//   min_distance_to_line_segment()  src/render/rtree_segments.rs:40-68
//   Segment::project_point_on / squared_distance_to_point   src/geometry/segment.rs:54-99
//   Point::squared_distance_to      src/geometry/point.rs:38-42
//
// Arithmetic contract: IEEE binary64, every operation rounded separately (compiled with
// -ffp-contract=off: no v_fma/v_fmac contraction of a*b+c), correctly rounded divide and
// sqrt, round-half-away quantisation.  Winding sign is the order-independent sum the
// reference builds with a sorted sweep (renderer_precise.rs:41-67).
//
// Mapping (MI355X): a "tile" = 256 consecutive OUTPUT bytes of one glyph bitmap (stores are
// coalesced row-major), one pixel per lane of a 256-thread workgroup (4 wave64).  Segments are
// staged through LDS in chunks of 256.  The default kernel (sdf_tiles_span, at the end of this
// file) sweeps up to 4 tiles of a glyph per staged chunk and looks, per pixel, only at the runs of
// the outline that bounds cannot rule out; the earlier generations above it are kept for A/B
// measurements and as independent bit-exact implementations in the tests.  The path is VALU bound,
// not HBM bound: see DESIGN.md §4 for the algorithm, the exactness argument and the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>

#include "outline_kernels.h" // PlanHeader (guard of the chunk-box pass behind the device front-end)
#include "sdf_kernels.h"

namespace vgsdf {

constexpr int TPB = VGSDF_TILE_PIXELS; // 256 threads, one pixel each
constexpr int SEG_CHUNK = 512;         // segments per LDS stage: 7 * 512 * 8 B = 28 KiB

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap so that
// each XCD walks a contiguous range of tiles: tiles of one glyph (which re-read the same
// segment list) then hit the same L2.  Pure performance hint; any placement is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n_and_flag)
{
	constexpr uint32_t X = 8;
	if (n_and_flag & 0x80000000u) // host asked for dispatch order == list order
		return b;
	const uint32_t n = n_and_flag;
	uint32_t per = n / X, rem = n % X;
	uint32_t xcd = b % X, idx = b / X;
	// XCDs [0, rem) own per+1 tiles, the rest own per tiles
	uint32_t start = xcd * per + (xcd < rem ? xcd : rem);
	return start + idx;
}

// Rust `n.round() as u8` on a value already clamped to [0,255]
__device__ __forceinline__ uint8_t quantise(double best_sq, bool inside)
{
	double d = sqrt(best_sq);            // rtree_segments.rs:67 (correctly rounded)
	if (inside)
		d = -d;                          // renderer_precise.rs:71-73
	d = d * (256.0 / 8.0) + 64.0;        // :75  (two roundings, no FMA)
	double n = 255.0 - d;                // :76
	n = n < 0.0 ? 0.0 : n;
	n = n > 255.0 ? 255.0 : n;
	return (uint8_t)(int)round(n);       // :79  half away from zero
}

// Exact squared distance from p to segment (v,w): Segment::squared_distance_to_point,
// segment.rs:54-72,96-99 with Point::squared_distance_to, point.rs:38-42.  dx,dy,l2 are the
// reference's (w.x - v.x), (w.y - v.y) and v.squared_distance_to(w), bit for bit.
__device__ __forceinline__ double exact_dist_sq(double px, double py, double vx, double vy, double wx,
                                                double wy, double dx, double dy, double l2)
{
	const double pvx = px - vx, pvy = py - vy;
	const double t = (pvx * dx + pvy * dy) / l2; // NaN when l2 == 0 (0/0): masked by at_v below
	double qx = vx + t * dx, qy = vy + t * dy;
	const bool at_v = (l2 == 0.0) | (t < 0.0); // segment.rs:59-61, :65-66
	const bool at_w = t > 1.0;                 // :67-68
	qx = at_w ? wx : qx;
	qy = at_w ? wy : qy;
	qx = at_v ? vx : qx;
	qy = at_v ? vy : qy;
	const double ex = qx - px, ey = qy - py; // point.rs:39-40 (other - self)
	return ex * ex + ey * ey;
}

// ---------------------------------------------------------------------------------------
// Variant 1: brute force.  Every pixel evaluates every segment.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void sdf_tiles_brute(const GlyphDesc *__restrict__ glyphs,
                                                       const uint2 *__restrict__ tiles,
                                                       uint32_t n_tiles,
                                                       const double *__restrict__ seg_sx,
                                                       const double *__restrict__ seg_sy,
                                                       const double *__restrict__ seg_ex,
                                                       const double *__restrict__ seg_ey, uint32_t seg_stride,
                                                       uint8_t *__restrict__ out)
{
	__shared__ double s_vx[SEG_CHUNK], s_vy[SEG_CHUNK], s_wx[SEG_CHUNK], s_wy[SEG_CHUNK];
	__shared__ double s_dx[SEG_CHUNK], s_dy[SEG_CHUNK], s_l2[SEG_CHUNK];

	const uint32_t tile = xcd_remap(blockIdx.x, n_tiles);
	const uint2 t = tiles[tile];
	const GlyphDesc g = glyphs[t.x];
	const uint32_t npix = g.w * g.h;
	const uint32_t o = t.y + threadIdx.x; // index into the glyph's output bitmap
	const bool active = o < npix;
	const uint32_t oc = active ? o : npix - 1;
	const uint32_t row = oc / g.w;       // output row (top row first)
	const uint32_t x = oc - row * g.w;
	const uint32_t y = g.h - 1 - row;    // renderer_precise.rs:78 inverts Y
	const double px = (double)x + ((double)g.x0 + 0.5); // :27,62
	const double py = (double)y + ((double)g.y0 + 0.5); // :28,34

	double best = __builtin_huge_val(); // rtree_segments.rs:57
	int wn = 0;

	for (uint32_t c0 = 0; c0 < g.n_seg; c0 += SEG_CHUNK) {
		const uint32_t cnt = min((uint32_t)SEG_CHUNK, g.n_seg - c0);
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < cnt; i += TPB) {
			const size_t s = (size_t)(g.seg_off + c0 + i) * seg_stride; // stride 1: four SoA arrays; 4: {sx, sy, ex, ey} records
			const double vx = seg_sx[s], vy = seg_sy[s], wx = seg_ex[s], wy = seg_ey[s];
			const double dx = wx - vx, dy = wy - vy; // segment.rs:63 (w.x - v.x), (w.y - v.y)
			s_vx[i] = vx;
			s_vy[i] = vy;
			s_wx[i] = wx;
			s_wy[i] = wy;
			s_dx[i] = dx;
			s_dy[i] = dy;
			s_l2[i] = dx * dx + dy * dy; // point.rs:38-42 v.squared_distance_to(w)
		}
		__syncthreads();

		for (uint32_t i = 0; i < cnt; i++) {
			const double vx = s_vx[i], vy = s_vy[i], wx = s_wx[i], wy = s_wy[i];
			const double dx = s_dx[i], dy = s_dy[i], l2 = s_l2[i];

			const double d2 = exact_dist_sq(px, py, vx, vy, wx, wy, dx, dy, l2);
			best = d2 < best ? d2 : best;                // rtree_segments.rs:60-62

			// --- winding: renderer_precise.rs:41-51, 63-66 ---
			const bool up = (vy <= py) & (wy > py);
			const bool down = (vy > py) & (wy <= py);
			if (up | down) {
				const double tc = (py - vy) / dy;        // e.y - s.y == dy bit for bit
				const double xc = vx + tc * dx;
				if (xc <= px)
					wn -= up ? 1 : -1;
			}
		}
	}

	if (active)
		out[g.out_off + o] = quantise(best, wn != 0);
}

// ---------------------------------------------------------------------------------------
// Shared pieces of the filtered kernels (the default kernel at the end of this file, and the earlier
// generations kept in tools/experiments/sdf_retired.inc for development builds).
// ---------------------------------------------------------------------------------------
constexpr int FCHUNK = 256;      // segments per LDS stage: 20 B filter record + 32 B exact end points each
constexpr int DELTA_CAP = 2048;  // winding histogram cells per span: rows * (w + 1)

// smallest integer n in [A, B] with (double)n + c >= v   (B if none): exact f64 compares
__device__ __forceinline__ int first_ge(double v, double c, int A, int B)
{
	double a = ceil(v - c);
	a = a < (double)A ? (double)A : a;
	a = a > (double)B ? (double)B : a; // NaN ends up inside [A, B] too; the compares below are then false
	int n = (int)a;
	if (n > A && (double)(n - 1) + c >= v)
		n--;
	else if (n < B && (double)n + c < v)
		n++;
	return n;
}

// h(F): bound on |Ft - D| for a filter value F, coordinates bounded by M:
// 64 u M sqrt(F) + 32 u F + 2^-34 M^2 with u = 2^-24, evaluated with upward slack.
__device__ __forceinline__ float filter_err(float F, float M)
{
	return 1.001f * (3.814697265625e-06f * M * __builtin_sqrtf(F) + 1.9073486328125e-06f * F +
	                 5.820766091346741e-11f * M * M);
}

__device__ __forceinline__ float sc_filter(float rpx, float rpy, float vx, float vy, float dx, float dy, float inv)
{
	const float pvx = rpx - vx, pvy = rpy - vy;
	const float t = __builtin_amdgcn_fmed3f(__builtin_fmaf(pvy, dy, pvx * dx) * inv, 0.0f, 1.0f);
	const float ex = __builtin_fmaf(-t, dx, pvx), ey = __builtin_fmaf(-t, dy, pvy);
	return __builtin_fmaf(ey, ey, ex * ex);
}

#ifdef VGSDF_DEV_VARIANTS
} // namespace vgsdf
#include "sdf_retired.inc" // tools/experiments/: earlier kernel generations (A/B measurements, `make dev` only: -I../tools/experiments)
namespace vgsdf {
#endif

// ---------------------------------------------------------------------------------------
// Chunk boxes (batch preparation, run once per resident batch): the bounding box of every chunk
// of 256 consecutive segments of a glyph, in f32 relative to the glyph origin (x0, y0).  The span
// kernel reads one float4 per chunk and skips chunks that can neither cross its sample rows nor
// hold a segment within reach of its pixels: a glyph with a long segment list (Noto Sans has one
// with 4368 segments = 18 chunks) no longer drags every tile through every chunk, and its chain
// of chunks stops setting the makespan of a small batch.
// Index of chunk c of glyph g: (seg_off[g] >> 8) + g + c — unique for a monotone seg_off, so no
// offset table is needed; the table has (total segments >> 8) + n_glyphs + 1 entries.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sdf_chunk_boxes(const GlyphDesc *__restrict__ glyphs, uint32_t n_glyphs,
                                                      const double *__restrict__ seg_sx,
                                                      const double *__restrict__ seg_sy,
                                                      const double *__restrict__ seg_ex,
                                                      const double *__restrict__ seg_ey, uint32_t seg_stride,
                                                      float4 *__restrict__ boxes, const PlanHeader *__restrict__ guard,
                                                      unsigned long long seg_cap)
{
	// behind the device front-end: the segment arrays hold nothing (and are smaller than the descriptors say)
	// when the batch is in error or did not fit its capacity guess; the host then launches again
	if (guard != nullptr && (guard->error || guard->n_segments > seg_cap))
		return;
	// grid = (glyphs, CHUNK_BOX_Y): chunk c of a glyph is taken by the wave with blockIdx.y == c % gridDim.y, so
	// the 18 chunks of a long glyph are not walked one after the other by a single wave
	// one wave per (glyph, chunk mod gridDim.y); four glyphs per workgroup (fewer, larger workgroups to dispatch)
	const uint32_t gi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (gi >= n_glyphs)
		return;
	const GlyphDesc g = glyphs[gi];
	if (g.n_seg <= 2 * FCHUNK) // the span kernel asks for boxes only when a glyph has more than two chunks
		return;
	for (uint32_t c = blockIdx.y; (uint64_t)c * FCHUNK < g.n_seg; c += gridDim.y) {
		const uint32_t c0 = c * FCHUNK;
		const uint32_t cnt = min((uint32_t)FCHUNK, g.n_seg - c0);
		float x0 = __builtin_inff(), y0 = __builtin_inff(), x1 = -__builtin_inff(), y1 = -__builtin_inff();
		bool bad = false;
		for (uint32_t i = lane; i < cnt; i += 64) {
			const size_t s = (size_t)(g.seg_off + c0 + i) * seg_stride;
			const double rvx = seg_sx[s] - (double)g.x0, rvy = seg_sy[s] - (double)g.y0;
			const double rwx = seg_ex[s] - (double)g.x0, rwy = seg_ey[s] - (double)g.y0;
			const float a = (float)rvx, b = (float)rvy, cx = (float)rwx, d = (float)rwy;
			bad |= !(fabsf(a) < 3.0e38f) || !(fabsf(b) < 3.0e38f) || !(fabsf(cx) < 3.0e38f) || !(fabsf(d) < 3.0e38f);
			x0 = fminf(x0, fminf(a, cx));
			x1 = fmaxf(x1, fmaxf(a, cx));
			y0 = fminf(y0, fminf(b, d));
			y1 = fmaxf(y1, fmaxf(b, d));
		}
		for (int sh = 32; sh > 0; sh >>= 1) {
			x0 = fminf(x0, __shfl_xor(x0, sh));
			y0 = fminf(y0, __shfl_xor(y0, sh));
			x1 = fmaxf(x1, __shfl_xor(x1, sh));
			y1 = fmaxf(y1, __shfl_xor(y1, sh));
		}
		const bool any_bad = __any(bad);
		if (lane == 0) {
			// non-finite coordinates: a box that contains everything (the chunk is never skipped)
			const float inf = __builtin_inff();
			boxes[chunk_box_index(g.seg_off, gi, c)] = any_bad ? make_float4(-inf, -inf, inf, inf) : make_float4(x0, y0, x1, y1);
		}
	}
}

constexpr uint32_t SPAN_TILES = 4; // tiles a workgroup of the default kernel sweeps per staged chunk

#include "sdf_span_kernel.inc" // the default kernel: bounded groups over spans (sdf_tiles_span)

#ifdef VGSDF_DEV_VARIANTS
// development builds: the same kernel source a second time, with in-kernel stamps (cdna_hip_programming.md, "In-kernel
// stamps": read the SHARES, never this instance's run time) and counters, summed per wave into g_span_dbg
__device__ unsigned long long g_span_dbg[24];
#define VG_SPAN_KERNEL sdf_tiles_span_stamped
#define VG_SPAN_STAMPED 1
#define VG_STAMP(region)                                                                      \
	do {                                                                                      \
		unsigned long long t_;                                                                \
		__builtin_amdgcn_sched_barrier(0);                                                    \
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
		__builtin_amdgcn_sched_barrier(0);                                                    \
		if ((region) >= 0)                                                                    \
			st_acc[(region) < 0 ? 0 : (region)] += t_ - st_last;                              \
		st_last = t_;                                                                         \
	} while (0)
#define VG_COUNT(stmt) stmt
#include "sdf_span_kernel.inc"
#undef VG_SPAN_STAMPED
#endif

} // namespace vgsdf

// ---------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------
extern "C" int vgsdf_filtered_delta_cap(void) { return vgsdf::DELTA_CAP; }

extern "C" size_t vgsdf_chunk_box_bytes(uint64_t n_segments, uint32_t n_glyphs)
{
	return sizeof(float4) * (size_t)((n_segments >> 8) + n_glyphs + 2);
}

extern "C" int vgsdf_launch_chunk_boxes(const vgsdf::GlyphDesc *glyphs, uint32_t n_glyphs, const double *sx, const double *sy,
                                        const double *ex, const double *ey, uint32_t seg_stride, void *boxes,
                                        const void *guard, unsigned long long seg_cap, hipStream_t stream)
{
	if (n_glyphs == 0)
		return 0;
	hipLaunchKernelGGL(vgsdf::sdf_chunk_boxes, dim3((n_glyphs + 3) / 4, 8), dim3(256), 0, stream, glyphs, n_glyphs, sx, sy, ex, ey,
	                   seg_stride, (float4 *)boxes, (const vgsdf::PlanHeader *)guard, seg_cap);
	return (int)hipGetLastError();
}

// The span kernel over a work list that outline_plan is still writing when this is enqueued: `grid` workgroups
// (a guess >= plan->n_main, else plan->ok is 0 and nothing runs), list order.
extern "C" int vgsdf_launch_span_planned(const vgsdf::GlyphDesc *glyphs, const uint2 *tiles, uint32_t grid, const double *sx,
                                         const double *sy, const double *ex, const double *ey, uint32_t seg_stride, uint8_t *out,
                                         const void *boxes, const void *plan, hipStream_t stream)
{
	if (grid == 0)
		return 0;
	hipLaunchKernelGGL(vgsdf::sdf_tiles_span, dim3(grid), dim3(vgsdf::TPB), 0, stream, glyphs, tiles, grid | 0x80000000u, sx, sy,
	                   ex, ey, seg_stride, out, (const float4 *)boxes, (const vgsdf::PlanHeader *)plan);
	return (int)hipGetLastError();
}

// kernel ids: 50 = the bounded-group span kernel (public variant 0), 1 = brute force, also the fallback
// for the tiles the host routes there: glyphs too wide for the winding histogram, or with >= 2^24
// segments.  Everything else exists only in development builds (-DVGSDF_DEV_VARIANTS).
extern "C" int vgsdf_kernel_known(int kernel)
{
#ifdef VGSDF_DEV_VARIANTS
	switch (kernel) {
	case 10: case 12: case 22: case 23: case 30: case 45: case 58:
		return 1;
	}
#endif
	return kernel == 1 || kernel == 50;
}

extern "C" int vgsdf_launch_tiles(int variant, int list_order, const vgsdf::GlyphDesc *glyphs,
                                  const uint2 *tiles, uint32_t n_tiles_in, const double *sx,
                                  const double *sy, const double *ex, const double *ey, uint32_t seg_stride,
                                  uint8_t *out, const void *boxes, hipStream_t stream)
{
	if (n_tiles_in == 0)
		return 0;
	const dim3 grid(n_tiles_in);
	// kernel argument: tile count, top bit set = dispatch in list order (no per-XCD remap)
	const uint32_t n_tiles = n_tiles_in | (list_order ? 0x80000000u : 0u);
	if (variant == 1)
		hipLaunchKernelGGL(vgsdf::sdf_tiles_brute, grid, dim3(vgsdf::TPB), 0, stream, glyphs,
		                   tiles, n_tiles, sx, sy, ex, ey, seg_stride, out);
	else if (variant == 50) // bounded groups over spans of up to 4 tiles (tile list: first pixel | T)
		hipLaunchKernelGGL(vgsdf::sdf_tiles_span, grid, dim3(vgsdf::TPB), 0, stream, glyphs, tiles, n_tiles, sx, sy, ex, ey,
		                   seg_stride, out, (const float4 *)boxes, (const vgsdf::PlanHeader *)nullptr);
#ifdef VGSDF_DEV_VARIANTS
	// development builds only (`make dev`): the stamped instance of the span kernel and the earlier generations
	else if (seg_stride != 1 && variant != 58) // the earlier generations read SoA arrays only
		return (int)hipErrorInvalidValue;
#define VG_LAUNCH_PK(A, C, P)                                                                            \
	hipLaunchKernelGGL((vgsdf::sdf_tiles_pk<A, C, P>), grid, dim3(vgsdf::TPB / P), 0, stream, glyphs, tiles,  \
	                   n_tiles, sx, sy, ex, ey, out)
#define VG_LAUNCH_HIER(A, L)                                                                               \
	hipLaunchKernelGGL((vgsdf::sdf_tiles_hier<A, L>), grid, dim3(vgsdf::TPB), 0, stream, glyphs, tiles, n_tiles,   \
	                   sx, sy, ex, ey, out)
#define VG_LAUNCH_FILTERED(A, C)                                                                          \
	hipLaunchKernelGGL((vgsdf::sdf_tiles_filtered<A, C>), grid, dim3(vgsdf::TPB), 0, stream, glyphs, \
	                   tiles, n_tiles, sx, sy, ex, ey, out)
	else if (variant == 58) { // diagnostic: s_memtime stamps per phase; prints the shares of wave time on stderr
		unsigned long long h[24] = {0};
		(void)hipMemcpyToSymbolAsync(HIP_SYMBOL(vgsdf::g_span_dbg), h, sizeof(h), 0, hipMemcpyHostToDevice, stream);
		hipLaunchKernelGGL(vgsdf::sdf_tiles_span_stamped, grid, dim3(vgsdf::TPB), 0, stream, glyphs, tiles, n_tiles, sx, sy, ex,
		                   ey, seg_stride, out, (const float4 *)boxes, (const vgsdf::PlanHeader *)nullptr);
		(void)hipStreamSynchronize(stream);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vgsdf::g_span_dbg), sizeof(h), 0, hipMemcpyDeviceToHost);
		static const char *names[12] = {"overhead", "wait chunk-top barrier", "stage", "wait post-stage barrier", "wave-level cull",
		                                "-", "phase 1", "phase 2", "decide", "exact evaluation", "quantise+state", "epilogue"};
		unsigned long long tot = 0;
		for (int r = 0; r < 12; r++)
			tot += h[r];
		std::fprintf(stderr, "[vgsdf stamps] %llu waves, %.0f ticks per wave\n", h[12], h[12] ? (double)tot / (double)h[12] : 0.0);
		for (int r = 0; r < 12; r++)
			std::fprintf(stderr, "[vgsdf stamps]   %-26s %6.2f %%\n", names[r], tot ? 100.0 * (double)h[r] / (double)tot : 0.0);
		std::fprintf(stderr, "[vgsdf counts] wave tile-chunks %llu (+ %llu without a surviving group), surviving groups %.1f per wave tile-chunk, pairs %llu (%.1f), rounds %llu (%.2f), waves with undecided lanes %llu (%.3f per wave tile-chunk), undecided lanes %llu\n",
		             h[15], h[19], h[15] + h[19] ? (double)h[18] / (double)(h[15] + h[19]) : 0.0, h[13], h[15] ? (double)h[13] / (double)h[15] : 0.0, h[14],
		             h[15] ? (double)h[14] / (double)h[15] : 0.0, h[16], h[15] ? (double)h[16] / (double)h[15] : 0.0, h[17]);
	}
	else if (variant == 30) // bounded groups on 256-pixel tiles
		VG_LAUNCH_HIER(0, false);
	else if (variant == 45) // ... exact evaluation only where the byte is undecided
		VG_LAUNCH_HIER(0, true);
	else if (variant == 22) // packed filter, grouped selection
		VG_LAUNCH_PK(0, false, 1);
	else if (variant == 23) // ... with per-wave culling
		VG_LAUNCH_PK(0, true, 1);
	else if (variant == 12) // scalar filter, top-4 keys
		VG_LAUNCH_FILTERED(0, false);
	else if (variant == 10) // ... with per-wave culling
		VG_LAUNCH_FILTERED(0, true);
#endif
	else
		return (int)hipErrorInvalidValue;
	return (int)hipGetLastError();
}
