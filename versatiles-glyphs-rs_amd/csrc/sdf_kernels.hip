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
// Mapping (MI355X): one 256-thread workgroup = 4 wave64 = one "tile" of 256 consecutive
// OUTPUT bytes of one glyph bitmap (so stores are coalesced row-major), one pixel per
// lane.  Segments are staged through LDS in SoA chunks with the per-segment invariants
// (w-v, |w-v|^2) computed once at staging time; every lane then walks the chunk with
// wave-uniform (broadcast) LDS reads.  The pair loop is FP64-VALU bound, not HBM bound
// (≈16 flop per 32/(w*h) bytes): see DESIGN.md for the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdf_kernels.h"

namespace vgsdf {

constexpr int TPB = VGSDF_TILE_PIXELS; // 256 threads, one pixel each
constexpr int SEG_CHUNK = 512;         // segments per LDS stage: 7 * 512 * 8 B = 28 KiB

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap so that
// each XCD walks a contiguous range of tiles: tiles of one glyph (which re-read the same
// segment list) then hit the same L2.  Pure performance hint; any placement is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
	constexpr uint32_t X = 8;
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

// ---------------------------------------------------------------------------------------
// Variant 1: brute force.  Every pixel evaluates every segment.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void sdf_tiles_brute(const GlyphDesc *__restrict__ glyphs,
                                                       const uint2 *__restrict__ tiles,
                                                       uint32_t n_tiles,
                                                       const double *__restrict__ seg_sx,
                                                       const double *__restrict__ seg_sy,
                                                       const double *__restrict__ seg_ex,
                                                       const double *__restrict__ seg_ey,
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
			const uint32_t s = g.seg_off + c0 + i;
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

			// --- distance: segment.rs:54-72 ---
			const double pvx = px - vx, pvy = py - vy;
			const double t = (pvx * dx + pvy * dy) / l2; // NaN when l2 == 0 (0/0) — see below
			double qx = vx + t * dx, qy = vy + t * dy;
			const bool at_v = (l2 == 0.0) | (t < 0.0);   // :59-61, :65-66
			const bool at_w = t > 1.0;                   // :67-68
			qx = at_w ? wx : qx;
			qy = at_w ? wy : qy;
			qx = at_v ? vx : qx;
			qy = at_v ? vy : qy;
			const double ex = qx - px, ey = qy - py;     // point.rs:39-40 (other - self)
			const double d2 = ex * ex + ey * ey;
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

} // namespace vgsdf

// ---------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------
extern "C" int vgsdf_launch_tiles(int variant, const vgsdf::GlyphDesc *glyphs, const uint2 *tiles,
                                  uint32_t n_tiles, const double *sx, const double *sy,
                                  const double *ex, const double *ey, uint8_t *out,
                                  hipStream_t stream)
{
	if (n_tiles == 0)
		return 0;
	(void)variant;
	hipLaunchKernelGGL(vgsdf::sdf_tiles_brute, dim3(n_tiles), dim3(vgsdf::TPB), 0, stream, glyphs,
	                   tiles, n_tiles, sx, sy, ex, ey, out);
	return (int)hipGetLastError();
}
