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
// generations kept in dev/sdf_retired.inc for development builds).
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
#include "dev/sdf_retired.inc" // earlier kernel generations (A/B measurements, development builds only)
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
__device__ __forceinline__ uint32_t chunk_box_index(uint32_t seg_off, uint32_t glyph, uint32_t c)
{
	return (seg_off >> 8) + glyph + c;
}

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

// ---------------------------------------------------------------------------------------
// Variant 0 = 50 (default): bounded groups over SPANS.  Same bounds, filter, bin decision and exact
// fallback as sdf_tiles_hier<.., LAZY>; what changes is the unit of work of a workgroup: up to
// SPAN = 4 consecutive 256-pixel tiles of ONE glyph.  A chunk of segments is staged (f64 loads,
// f32 records, row crossings, group bounds) ONCE and then swept by the span's tiles one after the
// other, so a glyph of <= 1024 pixels (99.7 % of Noto Sans) reads its segments from HBM/L2 once
// and pays the staging arithmetic once, instead of once per tile.  Per-pixel state between chunks
// (upper bound of the squared distance, the two candidate bytes) lives in LDS.
//   tiles[i] = (glyph, first pixel | T): T in 1..4 tiles, chosen by the host so that the rows the
//   span touches fit the winding histogram.
// Because quantisation is monotone in the distance, the minimum over chunks can be taken on the
// BYTES: outside the nearest segment gives the largest byte, inside the smallest; both are kept
// until the winding number (complete only after the last chunk) picks one.
// ---------------------------------------------------------------------------------------
constexpr uint32_t SPAN_TILES = 4;

template <int ABL>
__global__ __launch_bounds__(TPB, 4) void sdf_tiles_span(const GlyphDesc *__restrict__ glyphs,
                                                      const uint2 *__restrict__ tiles, uint32_t n_tiles,
                                                      const double *__restrict__ seg_sx,
                                                      const double *__restrict__ seg_sy,
                                                      const double *__restrict__ seg_ex,
                                                      const double *__restrict__ seg_ey, uint32_t seg_stride,
                                                      uint8_t *__restrict__ out, const float4 *__restrict__ boxes,
                                                      unsigned long long *__restrict__ dbg, const PlanHeader *__restrict__ plan)
{
	// Behind the device front-end the launch is enqueued before the host has seen the plan: the grid is a guess
	// (>= the work list, or the plan says "not ok": some capacity was too small and the host launches again)
	if (plan != nullptr && (!plan->ok || blockIdx.x >= plan->n_main))
		return;
	// ABL & 256 (development builds): s_memtime stamps per phase, summed per wave into dbg[region]
	// (cdna_hip_programming.md, "In-kernel stamps"); read the SHARES, never this build's run time.
	unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
	unsigned long long cn_pairs = 0, cn_rounds = 0, cn_tc = 0, cn_fb = 0, cn_fbl = 0; // counters of the diagnostic build
	auto STAMP = [&](int region) {
		if constexpr ((ABL & 256) != 0) {
			unsigned long long t;
			__builtin_amdgcn_sched_barrier(0);
			asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
			__builtin_amdgcn_sched_barrier(0);
			if (region >= 0)
				st_acc[region] += t - st_last;
			st_last = t;
		}
	};
	(void)dbg;
	STAMP(-1);
	constexpr uint32_t GRP = 8, NGRP = FCHUNK / GRP; // 32 groups per chunk: one mask bit each
	static_assert(NGRP == 32 && TPB == FCHUNK, "one candidate bit per group, one staging thread per record");
	__shared__ __attribute__((aligned(16))) float s_vx[FCHUNK], s_vy[FCHUNK], s_dx[FCHUNK], s_dy[FCHUNK], s_inv[FCHUNK];
	__shared__ double e_vx[FCHUNK], e_vy[FCHUNK], e_wx[FCHUNK], e_wy[FCHUNK]; // exact endpoints
	__shared__ int s_delta[DELTA_CAP];
	__shared__ uint32_t s_mbits[2]; // coordinate bound of the chunk being staged (slot = parity of processed chunks)
	__shared__ __attribute__((aligned(16))) float s_gx[NGRP], s_gy[NGRP], s_gr[NGRP]; // anchor, radius
	__shared__ float st_ub2[SPAN_TILES * TPB];    // per pixel: upper bound of the squared distance so far
	__shared__ uint16_t st_byte[SPAN_TILES * TPB]; // per pixel: byte if inside (low), byte if outside (high)
	constexpr uint32_t QCAP = 256;                 // pooled (pixel, group) pairs per wave and sweep; beyond: per-lane walk
	__shared__ uint16_t q_pair[TPB / 64][QCAP];    // (lane << 5) | group
	__shared__ uint32_t q_min[TPB / 64][64];       // per pixel of the wave: smallest filter value (bits)

	const uint32_t tid = threadIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t tile = xcd_remap(blockIdx.x, n_tiles);
	const uint2 t = tiles[tile];
	const GlyphDesc g = glyphs[t.x];
	const uint32_t npix = g.w * g.h;
	const double x0c = (double)g.x0 + 0.5, y0c = (double)g.y0 + 0.5;
	const uint32_t p0 = t.y & ~255u, T = min(max(t.y & 255u, 1u), SPAN_TILES);
	const uint32_t p_end = min(p0 + T * (uint32_t)TPB, npix); // pixels [p0, p_end) of the glyph's bitmap

	const uint32_t r_first = p0 / g.w, r_last = (p_end - 1) / g.w;
	const int y_hi = (int)(g.h - 1 - r_first), y_lo = (int)(g.h - 1 - r_last);
	// one histogram row = w + 1 cells (the last one collects "no column left of the pixel row's end"), padded to an odd
	// stride so that the per-row scans after the chunk loop (one thread per row) spread over the LDS banks
	const uint32_t stride = (g.w + 1) | 1u;
	const double band_lo = (double)y_lo + y0c, band_hi = (double)y_hi + y0c; // lowest / highest sample row of the span
	const uint32_t n_delta = (r_last - r_first + 1) * stride; // <= DELTA_CAP (host-checked)
	for (uint32_t i = tid; i < n_delta; i += TPB)
		s_delta[i] = 0;
	for (uint32_t k = 0; k < T; k++) {
		st_ub2[k * TPB + tid] = __builtin_inff();
		st_byte[k * TPB + tid] = 0x00FFu; // neutral: 255 for the min (inside), 0 for the max (outside)
	}

	const float wh = (float)max(g.w, g.h);
	// Origin of every f32 quantity of the filter (records, anchors, pixel centres): the middle of the bitmap,
	// an integer point.  The filter's error bound h(F) scales with M = the largest |coordinate| relative to
	// that origin, so the middle halves it against the corner (x0, y0) (and with it the share of pixels whose
	// byte needs the exact f64 evaluation).  Pixel centres stay exact in f32 (half-integers).
	const int hw = (int)(g.w >> 1), hh = (int)(g.h >> 1);
	const double ox = (double)g.x0 + (double)hw, oy = (double)g.y0 + (double)hh;
	const float mpix = 0.5f * wh + 1.0f; // >= |pixel centre - origin| in both axes
	uint32_t par = 0;
	if (tid == 0)
		s_mbits[0] = __float_as_uint(mpix);
	const float mabs0 = fmaxf(fmaxf(fabsf((float)g.x0), fabsf((float)g.y0)),
	                          fmaxf(fabsf((float)g.x0 + (float)g.w), fabsf((float)g.y0 + (float)g.h)));

	// the box test pays for glyphs with several chunks; with one or two there is nothing to gain
	const bool use_boxes = boxes != nullptr && !(ABL & 128) && g.n_seg > 2 * FCHUNK;
	for (uint32_t c0 = 0; c0 < g.n_seg; c0 += FCHUNK) {
		const uint32_t cnt = min((uint32_t)FCHUNK, g.n_seg - c0);
		STAMP(0); // prologue / loop overhead
		__syncthreads(); // previous chunk fully consumed (and s_delta / state initialised on the first trip)
		STAMP(1); // wait at the chunk-top barrier
		if (use_boxes) {
			// ---- chunk box test (workgroup-uniform).  Skip the chunk if its box lies strictly outside the
			// span's band of sample rows (so none of its segments crosses one of them: no winding
			// contribution) AND farther than SAT from every pixel of the span: a segment beyond SAT is either
			// not the minimum, or the byte is saturated whatever the minimum is.  f32 roundings (box, pixel
			// centres) are covered by padk.  (Tightening R with the pixels' running upper bounds was
			// measured: the bookkeeping costs more than the extra skips on real fonts.) ----
			const float4 bb = boxes[chunk_box_index(g.seg_off, t.x, c0 / FCHUNK)];
			const float mag = fmaxf(fmaxf(fabsf(bb.x), fabsf(bb.y)), fmaxf(fmaxf(fabsf(bb.z), fabsf(bb.w)), wh));
			const float padk = 0.02f + 2.0e-6f * mag;
			const float ry0 = (float)y_lo + 0.5f, ry1 = (float)y_hi + 0.5f; // sample rows of the span, relative to y0
			const float dy = fmaxf(bb.y - ry1, ry0 - bb.w) - padk;
			float dx = fmaxf(bb.x - ((float)g.w - 0.5f), 0.5f - bb.z) - padk;
			dx = dx > 0.0f ? dx : 0.0f;
			const float R = 6.2f + padk; // SAT
			if (dy > 0.0f && __builtin_fmaf(dy, dy, dx * dx) > R * R)
				continue; // NaN anywhere -> comparison false -> the chunk is processed
		}
		// ---- stage (thread i <-> record i): exact endpoints, f32 record, coordinate bound, group bounds,
		// row crossings.  Everything a thread needs from other threads here comes from lanes of its own wave
		// (a group of 8 records, the crossings of the wave's 64 segments), so there is one workgroup barrier
		// behind the stage and none inside it. ----
		constexpr float INFL = 1.0f + 1.0f / 512.0f;
		{
			const uint32_t i = tid;
			float mf = 0.0f;
			float fvx = 1.0e18f, fvy = 1.0e18f, fdx = 0.0f, fdy = 0.0f, finv = 0.0f; // padding: a record that can never win (F = 2e36)
			uint32_t nrow = 0; // sample rows of the span this thread's segment crosses: yy in [ya, ya + nrow)
			int ya = 0;
			if (i < cnt) {
				const size_t s = (size_t)(g.seg_off + c0 + i) * seg_stride; // stride 1: SoA arrays; 4: {sx, sy, ex, ey} records
				const double vx = seg_sx[s], vy = seg_sy[s], wx = seg_ex[s], wy = seg_ey[s];
				e_vx[i] = vx;
				e_vy[i] = vy;
				e_wx[i] = wx;
				e_wy[i] = wy;
				const double dx = wx - vx, dy = wy - vy;
				const double l2 = dx * dx + dy * dy;
				fvx = (float)(vx - ox);
				fvy = (float)(vy - oy);
				fdx = (float)dx;
				fdy = (float)dy;
				// 1 / |d|^2 to ~3 units of f32 roundoff (conversion + v_rcp_f32): the closest point of the f32 record
				// moves by <= 17 u M along the segment, inside the 28.6 u M the filter's error bound h(F) allows
				finv = (l2 > 1e-20 && l2 < 1e30) ? __builtin_amdgcn_rcpf((float)l2) : 0.0f;
				// coordinate bound from the f32 values (end point as v + d: <= 2.5 ulp(M) off), rounded up
				mf = fmaxf(fmaxf(fabsf(fvx), fabsf(fvy)), fmaxf(fabsf(fvx + fdx), fabsf(fvy + fdy))) * 1.00001f;
				mf = mf >= 0.0f ? mf : __builtin_inff(); // NaN -> inf ("no usable bound")
				// crossings, renderer_precise.rs:41-51: up (+1) s.y <= py < e.y; down (-1) e.y <= py < s.y.
				// Rows crossed: lo <= py < hi with py = yy + y0c exactly as the reference forms it.  Most segments of
				// a real font are a fraction of a pixel tall and most lie outside the span's band of rows: two
				// exact compares reject those before the integer brackets are computed.
				const bool up = vy < wy;
				const double lo = up ? vy : wy, hi = up ? wy : vy;
				if (!(ABL & 1) && vy != wy && hi > band_lo && lo <= band_hi) {
					ya = first_ge(lo, y0c, y_lo, y_hi + 1);
					const int yb = first_ge(hi, y0c, y_lo, y_hi + 1); // first row with py >= hi
					nrow = yb > ya ? (uint32_t)(yb - ya) : 0u;
				}
			}
			s_vx[i] = fvx;
			s_vy[i] = fvy;
			s_dx[i] = fdx;
			s_dy[i] = fdy;
			s_inv[i] = finv;
			// coordinate bound: wave maximum first, one LDS atomic per wave (non-negative floats order like uints)
			uint32_t mb = __float_as_uint(mf);
			for (int sh = 32; sh > 0; sh >>= 1)
				mb = max(mb, (uint32_t)__shfl_xor((int)mb, sh));
			if (lane == 0)
				atomicMax(&s_mbits[par], mb);

			// ---- group bounds (8 consecutive records = 8 lanes): anchor = start vertex of the middle member, radius
			// over all end points.  The pad that depends on the workgroup's coordinate bound is added in phase 1. ----
			{
				const uint32_t gb = tid & ~(GRP - 1);
				const uint32_t ai = min(gb + GRP / 2, cnt - 1); // (a lane of this group whenever the group holds a record)
				const float ax = __shfl(fvx, (int)(ai & 63u)), ay = __shfl(fvy, (int)(ai & 63u));
				float r2 = 0.0f;
				if (i < cnt) {
					const float wx = fvx + fdx, wy = fvy + fdy;
					const float ex = fvx - ax, ey = fvy - ay, fx = wx - ax, fy = wy - ay;
					const float dv = __builtin_fmaf(ey, ey, ex * ex), dw = __builtin_fmaf(fy, fy, fx * fx);
					r2 = dv > dw ? dv : dw;
				}
				for (int sh = 1; sh < (int)GRP; sh <<= 1) {
					const float other = __shfl_xor(r2, sh);
					r2 = other > r2 ? other : r2;
				}
				if ((tid & (GRP - 1)) == 0) {
					const bool empty = gb >= cnt;
					s_gx[tid / GRP] = empty ? 1.0e18f : ax;
					s_gy[tid / GRP] = empty ? 1.0e18f : ay;
					s_gr[tid / GRP] = empty ? 0.0f : __builtin_sqrtf(r2) * (INFL * INFL * 1.004f); // 1.004: see phase 1
				}
			}

			// ---- row crossings of the wave's 64 segments, one (segment, row) pair per lane and round: a vertical stem
			// crosses 20 rows, its neighbours none, and a per-thread loop would keep the whole workgroup waiting
			// at the barrier for the one long segment (2.6 trips per wave on average, up to 24, against 7.5
			// crossings per wave in total). ----
			{
				uint32_t incl = nrow; // inclusive prefix sum over the wave: DPP row shifts + row broadcasts
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false); // row_shr:1
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false); // row_shr:2
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false); // row_shr:4
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false); // row_shr:8
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				auto cross = [&](double vx, double vy, double dx, double dy, bool up, int yy) {
					const double pyy = (double)yy + y0c;
					const double tc = (pyy - vy) / dy;
					const double xc = vx + tc * dx;               // :45-46 / :48-49
					const int k = first_ge(xc, x0c, 0, (int)g.w); // first column with xc <= px (:63)
					if (k < (int)g.w)
						atomicAdd(&s_delta[(uint32_t)(y_hi - yy) * stride + (uint32_t)k], up ? -1 : 1); // wn -= sign
				};
				if (total != 0 && total <= QCAP) {
					// (lane, row) pairs of the wave, pooled in LDS (rows of a span: < 1024, host-checked through DELTA_CAP)
					uint32_t off = incl - nrow;
					for (uint32_t r = 0; r < nrow; r++)
						q_pair[wv][off++] = (uint16_t)((lane << 10) | (uint32_t)(y_hi - (ya + (int)r)));
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					for (uint32_t base = 0; base < total; base += 64) {
						const uint32_t idx = base + lane;
						if (idx < total) {
							const uint32_t e = q_pair[wv][idx];
							const uint32_t si = (tid & ~63u) + (e >> 10); // the segment's record (staged by this wave)
							const double vx = e_vx[si], vy = e_vy[si], wx = e_wx[si], wy = e_wy[si];
							cross(vx, vy, wx - vx, wy - vy, vy < wy, y_hi - (int)(e & 1023u));
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
				} else if (total != 0) { // more pairs than the pool holds (tall glyphs): every thread walks its own rows
					if (nrow != 0) {
						const double vx = e_vx[i], vy = e_vy[i], wx = e_wx[i], wy = e_wy[i];
						for (uint32_t r = 0; r < nrow; r++)
							cross(vx, vy, wx - vx, wy - vy, vy < wy, ya + (int)r);
					}
				}
			}
		}
		STAMP(2); // stage: loads, records, crossings, coordinate bound, group bounds
		__syncthreads();
		STAMP(3); // wait at the post-stage barrier
		const float Mc = __uint_as_float(s_mbits[par]);
		par ^= 1u;
		if (tid == 0)
			s_mbits[par] = __float_as_uint(mpix); // the next processed chunk's accumulator (nobody touches it before the next chunk-top barrier)
		const bool sane = Mc < 1.0e6f;   // else: no usable f32 bound -> every segment is evaluated exactly
		const bool bounded = Mc < 4096.0f; // group bounds have a useful margin
		const float pad = 0.01f + 1.0e-5f * Mc;
		const uint32_t n_groups = (cnt + GRP - 1) / GRP;

		const float4 *q_vx = reinterpret_cast<const float4 *>(s_vx), *q_vy = reinterpret_cast<const float4 *>(s_vy);
		const float4 *q_dx = reinterpret_cast<const float4 *>(s_dx), *q_dy = reinterpret_cast<const float4 *>(s_dy);
		const float4 *q_inv = reinterpret_cast<const float4 *>(s_inv);
		const float M = Mc;
		const float e64 = 5.6843418860808015e-14f * M * (M + mabs0 + M); // 2^-44 M (M + Mabs)
		// h(F) + e64 as one fused expression in the decide step: h(F) = 1.001 (64 u M sqrt(F) + 32 u F + 2^-34 M^2)
		const float herr_c1 = 1.001f * 3.814697265625e-06f * M, herr_c3 = 1.001f * 5.820766091346741e-11f * M * M + e64;

		// ---- sweep: every tile of the span against the staged chunk ----
#pragma unroll 1
		for (uint32_t k = 0; k < T; k++) {
			// compiler barrier: keeps the (loop-invariant, wave-uniform) group-bound loads inside the loop
			// instead of parking 96 VGPRs of them across it
			asm volatile("" ::: "memory");
			const uint32_t o = p0 + k * TPB + tid;
			// a wave whose 64 pixels all lie past the end of the bitmap (last tile of the glyph) has nothing to do
			if (p0 + k * TPB + (tid & ~63u) >= npix)
				continue;
			const uint32_t oc = o < npix ? o : npix - 1;
			const uint32_t row = oc / g.w;
			const uint32_t x = oc - row * g.w;
			const uint32_t y = g.h - 1 - row;
			const float rpx = (float)((int)x - hw) + 0.5f, rpy = (float)((int)y - hh) + 0.5f; // pixel centre relative to the origin
			float ub2 = st_ub2[k * TPB + tid];
			STAMP(0);

			// ---- phase 1: candidate groups of this lane ----
			uint32_t cand = n_groups >= 32 ? 0xFFFFFFFFu : ((1u << n_groups) - 1u);
			if (ABL & 32)
				cand &= 1u;
			if (bounded && !(ABL & 32)) {
				// D_g^2 of two groups per register, upper 16 bits of each float (truncation: the stored value
				// is <= the true one; the high half read as a float is < true * (1 + 2^-7)): keeps the kernel
				// inside the 128-VGPR budget of 4 waves per SIMD.  The test below inflates U + r_g by 1.004
				// (> sqrt(1 + 2^-7)), so every true candidate still passes.
				uint32_t D2p[NGRP / 2];
				uint32_t dmin = __float_as_uint(ub2);
				const float4 *gx4 = reinterpret_cast<const float4 *>(s_gx), *gy4 = reinterpret_cast<const float4 *>(s_gy);
				const float4 *gr4 = reinterpret_cast<const float4 *>(s_gr);
#pragma unroll
				for (uint32_t b = 0; b < NGRP / 4; b++) {
					if (b * 4 < n_groups) {
						const float4 ax = gx4[b], ay = gy4[b];
						const float axs[4] = {ax.x, ax.y, ax.z, ax.w}, ays[4] = {ay.x, ay.y, ay.z, ay.w};
						uint32_t d2[4];
#pragma unroll
						for (int j = 0; j < 4; j++) {
							const float ddx = rpx - axs[j], ddy = rpy - ays[j];
							d2[j] = __float_as_uint(__builtin_fmaf(ddy, ddy, ddx * ddx));
						}
						// d2 >= +0: unsigned order of the bits is float order; one v_min3_u32 per two groups
						asm("v_min3_u32 %0, %1, %2, %3" : "=v"(dmin) : "v"(dmin), "v"(d2[0]), "v"(d2[1]));
						asm("v_min3_u32 %0, %1, %2, %3" : "=v"(dmin) : "v"(dmin), "v"(d2[2]), "v"(d2[3]));
						D2p[b * 2] = __builtin_amdgcn_perm(d2[1], d2[0], 0x07060302u);     // hi16(d2[1]) : hi16(d2[0])
						D2p[b * 2 + 1] = __builtin_amdgcn_perm(d2[3], d2[2], 0x07060302u);
					} else {
						D2p[b * 2] = D2p[b * 2 + 1] = 0x7F7F7F7Fu; // 3.4e38 : 3.4e38 (finite: the test below stays NaN-free)
					}
				}
				ub2 = __uint_as_float(dmin);
				float U = (__builtin_sqrtf(ub2) * INFL + pad) * INFL;
				U = U < 6.2f ? U : 6.2f; // SAT: beyond it the byte is saturated whatever the minimum is
				const float Ui = (U + pad * INFL) * 1.004f; // s_gr is stored with the same factor; it lacks the pad of r_g
				// One bit per group, 3-4 VALU ops each: tt = (U + r_g) 1.004, diff = tt^2 - D_g^2 (sign bit set
				// <=> not a candidate; -3.4e38 for the groups past n_groups), shifted in with v_alignbit.  The
				// bits arrive inverted and in reverse order: fixed once with v_not / v_bfrev.
				uint32_t rej = 0xFFFFFFFFu;
#pragma unroll
				for (uint32_t b = 0; b < NGRP / 4; b++) {
					float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					if (b * 4 < n_groups)
						r = gr4[b];
					const float rs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const uint32_t pk = D2p[b * 2 + j / 2];
						const float d2 = __uint_as_float((j & 1) ? pk : (pk << 16));
						const float tt = Ui + rs[j];
						const float diff = __builtin_fmaf(tt, tt, -d2);
						rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(diff), 31); // (rej << 1) | sign(diff)
					}
				}
				cand &= __builtin_bitreverse32(~rej); // (cand starts as the mask of the chunk's real groups)
			}


			STAMP(6); // phase 1
			// ---- phase 2: smallest f32 filter value over the lane's candidate groups (F >= +0: unsigned
			// order of the bits is float order) ----
			uint32_t k1 = 0xFFFFFFFFu;
			auto group_min = [&](uint32_t acc, uint32_t gq, float qx, float qy) { // min(acc, F of the 8 members of group gq)
				// two halves of 4 records: 20 + 4 live values instead of 40 + 8 (the kernel sits at the 128-VGPR
				// limit of 4 waves per SIMD; spills would go to scratch, i.e. to memory)
#pragma unroll
				for (uint32_t hq = 0; hq < 2; hq++) {
					const float4 vx4 = q_vx[2 * gq + hq], vy4 = q_vy[2 * gq + hq], dx4 = q_dx[2 * gq + hq], dy4 = q_dy[2 * gq + hq],
					             iv4 = q_inv[2 * gq + hq];
					const uint32_t f0 = __float_as_uint(sc_filter(qx, qy, vx4.x, vy4.x, dx4.x, dy4.x, iv4.x));
					const uint32_t f1 = __float_as_uint(sc_filter(qx, qy, vx4.y, vy4.y, dx4.y, dy4.y, iv4.y));
					const uint32_t f2 = __float_as_uint(sc_filter(qx, qy, vx4.z, vy4.z, dx4.z, dy4.z, iv4.z));
					const uint32_t f3 = __float_as_uint(sc_filter(qx, qy, vx4.w, vy4.w, dx4.w, dy4.w, iv4.w));
					uint32_t mn;
					asm("v_min3_u32 %0, %1, %2, %3" : "=v"(mn) : "v"(acc), "v"(f0), "v"(f1));
					asm("v_min3_u32 %0, %1, %2, %3" : "=v"(acc) : "v"(mn), "v"(f2), "v"(f3));
					if (hq == 0)
						__builtin_amdgcn_sched_barrier(0);
				}
				return acc;
			};
			if (sane && !(ABL & 2)) {
				if (o >= npix)
					cand = 0; // padding lanes of the last tile own no pixel
				// The lanes of a wave hold very different numbers of candidate groups (mean 1.7 per chunk,
				// busiest lane 4-5).  Instead of every lane walking its own list while the others idle, the
				// wave pools its (pixel, group) pairs in LDS and deals them out evenly: 64 pairs per round,
				// results merged with ds_min_u32 on the owning pixel's slot.
				const uint32_t c = (uint32_t)__builtin_popcount(cand);
				uint32_t incl = c; // inclusive prefix sum over the wave: DPP row shifts + row broadcasts
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false); // row_shr:1
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false); // row_shr:2
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false); // row_shr:4
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false); // row_shr:8
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
				incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				if constexpr ((ABL & 256) != 0) {
					cn_pairs += total;
					cn_rounds += (total + 63) / 64;
					cn_tc += 1;
				}
				if (total <= QCAP && !(ABL & 64)) {
					uint32_t off = incl - c;
					q_min[wv][lane] = 0xFFFFFFFFu;
					uint32_t m = cand;
					while (m) {
						q_pair[wv][off++] = (uint16_t)((lane << 5) | (uint32_t)__builtin_ctz(m));
						m &= m - 1;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					for (uint32_t base = 0; base < total && !(ABL & 512); base += 64) {
						const uint32_t idx = base + lane;
						const bool valid = idx < total;
						const uint32_t e = q_pair[wv][valid ? idx : base];
						const uint32_t src = e >> 5;
						const float qx = __shfl(rpx, (int)src), qy = __shfl(rpy, (int)src);
						const uint32_t mn = group_min(0xFFFFFFFFu, e & 31u, qx, qy);
						if (valid)
							atomicMin(&q_min[wv][src], mn);
					}
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
					k1 = q_min[wv][lane];
					if (ABL & 1024)
						k1 = 0xFFFFFFFFu;
				} else {
					uint32_t m = cand;
					while (m) {
						const uint32_t gq = (uint32_t)__builtin_ctz(m);
						m &= m - 1;
						k1 = group_min(k1, gq, rpx, rpy);
					}
				}
			}

			// ---- decide.  The chunk's minimum C lies in [LB, U] (every candidate's F is >= the smallest
			// one, L(f) = f - h(f) - e64 is increasing for f >= c^2, c = 1.001 * 64 u M; segments of
			// non-candidate groups are beyond the true minimum or beyond SAT, 6.16^2 = 38).  Both bytes
			// are functions of the bin q = floor(32 sqrt(C) + 1/2) (renderer_precise.rs:71-79: 191 - q
			// outside, 191 + q inside); if the whole interval falls into one bin no f64 work is needed.
			// The reference's own roundings move 32 sqrt(C) by < 1e-12, the f32 evaluation by < 1e-4. ----
			STAMP(7); // phase 2 (pooling + filter rounds)
			bool have = false;
			uint32_t b_in = 255u, b_out = 0u;
			double best = __builtin_huge_val(); // rtree_segments.rs:57
			const double px = (double)x + x0c, py = (double)y + y0c; // renderer_precise.rs:27-28,34,62
			auto exact_lds = [&](uint32_t i) {
				const double vx = e_vx[i], vy = e_vy[i], wx = e_wx[i], wy = e_wy[i];
				const double dx = wx - vx, dy = wy - vy; // segment.rs:63
				const double d2 = exact_dist_sq(px, py, vx, vy, wx, wy, dx, dy, dx * dx + dy * dy);
				best = d2 < best ? d2 : best; // rtree_segments.rs:60-62
			};
			if (!sane) {
				for (uint32_t j = 0; j < cnt; j++)
					exact_lds(j);
				have = true;
			} else if (k1 != 0xFFFFFFFFu) {
				// C lies in [f1 - e, f1 + e], e = h(f1) + e64 (one fused expression; the 1.001 of filter_err covers its
				// roundings).  Bin number of the reference's quantisation: floor(32 sqrt(C) + 1/2).  With 8 e <= f1,
				// |32 sqrt(C) - 32 sqrt(f1)| <= 16.55 e / sqrt(f1) on the whole interval (sqrt is concave: upper side
				// e / (2 sqrt f1), lower side e / (sqrt f1 + sqrt(f1 - e)) <= 1.034 e / (2 sqrt f1)), so if
				// s = 32 sqrt(f1) + 1/2 is farther than dl = 17 e / sqrt(f1) + 3e-4 from the next integer (3e-4: f32
				// roundings of s, < 3e-5, and of the reference's own arithmetic, < 1e-12) every C of the interval
				// falls into bin floor(s).  Beyond C = 35.9 (32 sqrt(C) + 1/2 >= 192) both bytes are saturated (0 / 255)
				// whatever the bin is.  NaN anywhere -> every comparison false -> not decided -> exact evaluation.
				const float f1 = __uint_as_float(k1);
				const float sq = __builtin_sqrtf(f1);
				const float e = __builtin_fmaf(herr_c1, sq, __builtin_fmaf(1.9092559814453125e-06f, f1, herr_c3));
				float U = f1 + e;
				const float uf = U * (1.0f + 1.0f / 1048576.0f);
				ub2 = uf < ub2 ? uf : ub2; // bounds the later chunks' candidates too
				const float s = __builtin_fmaf(32.0f, sq, 0.5f);
				const float dl = __builtin_fmaf(17.0f * e, __builtin_amdgcn_rcpf(sq), 3.0e-4f);
				const float fr = __builtin_amdgcn_fractf(s);
				const float mn = fr < 1.0f - fr ? fr : 1.0f - fr;
				const bool far = f1 - e > 35.9f;
				const bool decided = far || (8.0f * e <= f1 && mn > dl);
				if (decided) {
					const int q = far ? 200 : (int)s;
					b_in = (uint32_t)min(191 + q, 255);
					b_out = (uint32_t)max(191 - q, 0);
				}
				if (!(U >= 0.0f))
					U = __builtin_inff();
				STAMP(8); // decide
				if constexpr ((ABL & 256) != 0) {
					const unsigned long long amb = __ballot(!decided);
					cn_fb += amb != 0;
					cn_fbl += (unsigned long long)__builtin_popcountll(amb);
				}
				if (!decided && !(ABL & 4)) {
					// Rescan of the lane's candidate groups against a threshold Tk with L(Tk) > U (L increasing
					// above it): fixed-point iteration for the crossing, pushed up, then VERIFIED; if the check
					// fails nothing is excluded (Tk = inf).  Everything at or below Tk is evaluated exactly.
					float Tk = U + e64;
					for (int it = 0; it < 3; it++)
						Tk = U + e64 + filter_err(Tk, M);
					Tk = Tk * 1.001f + 1e-30f;
					if (!(Tk - filter_err(Tk, M) - e64 > U))
						Tk = __builtin_inff();
					uint32_t m = cand;
					while (m) {
						const uint32_t gq = (uint32_t)__builtin_ctz(m);
						m &= m - 1;
						// members at or below the threshold (padded records have F = 2e36), 4 records at a time
						uint32_t hit = 0;
#pragma unroll 1
						for (uint32_t hq = 0; hq < 2; hq++) {
							const float4 vx4 = q_vx[2 * gq + hq], vy4 = q_vy[2 * gq + hq], dx4 = q_dx[2 * gq + hq], dy4 = q_dy[2 * gq + hq],
							             iv4 = q_inv[2 * gq + hq];
							hit |= !(sc_filter(rpx, rpy, vx4.x, vy4.x, dx4.x, dy4.x, iv4.x) > Tk) ? (1u << (4 * hq)) : 0u;
							hit |= !(sc_filter(rpx, rpy, vx4.y, vy4.y, dx4.y, dy4.y, iv4.y) > Tk) ? (2u << (4 * hq)) : 0u;
							hit |= !(sc_filter(rpx, rpy, vx4.z, vy4.z, dx4.z, dy4.z, iv4.z) > Tk) ? (4u << (4 * hq)) : 0u;
							hit |= !(sc_filter(rpx, rpy, vx4.w, vy4.w, dx4.w, dy4.w, iv4.w) > Tk) ? (8u << (4 * hq)) : 0u;
						}
						while (hit) {
							const uint32_t j = gq * GRP + (uint32_t)__builtin_ctz(hit);
							hit &= hit - 1;
							if (j < cnt)
								exact_lds(j);
						}
					}
					have = true;
				}
			}
			STAMP(9); // exact fallback (rescan + f64)
			if (have) {
				b_in = quantise(best, true);
				b_out = quantise(best, false);
			}
			// minimum over the chunks, taken on the bytes (monotone in the distance)
			const uint32_t old = st_byte[k * TPB + tid];
			b_in = min(b_in, old & 255u);
			b_out = max(b_out, old >> 8);
			st_byte[k * TPB + tid] = (uint16_t)(b_in | (b_out << 8));
			st_ub2[k * TPB + tid] = ub2;
			STAMP(10); // quantise + state update
		}
	}

	// ---- winding number = prefix sum of the row's histogram up to the pixel's column (renderer_precise.rs:58-66).
	// One thread per row turns its row into inclusive prefix sums in place (exact integer sums, any order);
	// every pixel then reads one cell and picks its byte. ----
	__syncthreads(); // all crossings recorded (n_seg == 0: the initial state / zeroed histogram must be visible)
	{
		const uint32_t n_rows = r_last - r_first + 1;
		for (uint32_t r = tid; r < n_rows; r += TPB) {
			int *drow = s_delta + r * stride;
			int acc = 0;
#pragma unroll 4
			for (uint32_t c = 0; c < g.w; c++) {
				acc += drow[c];
				drow[c] = acc;
			}
		}
	}
	__syncthreads();
	for (uint32_t k = 0; k < T; k++) {
		const uint32_t o = p0 + k * TPB + tid;
		if (o < npix) {
			const uint32_t row = o / g.w;
			const uint32_t x = o - row * g.w;
			const int wn = s_delta[(row - r_first) * stride + x];
			const uint32_t sb = st_byte[k * TPB + tid];
			out[g.out_off + o] = (uint8_t)(wn != 0 ? (sb & 255u) : (sb >> 8));
		}
	}
	STAMP(11); // epilogue: winding prefix sums, stores
	if constexpr ((ABL & 256) != 0) {
		if (dbg != nullptr && lane == 0) {
			for (int r = 0; r < 12; r++)
				atomicAdd(&dbg[r], st_acc[r]);
			atomicAdd(&dbg[12], 1ull);
			atomicAdd(&dbg[13], cn_pairs);
			atomicAdd(&dbg[14], cn_rounds);
			atomicAdd(&dbg[15], cn_tc);
			atomicAdd(&dbg[16], cn_fb);
			atomicAdd(&dbg[17], cn_fbl);
		}
	}
}

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
	hipLaunchKernelGGL((vgsdf::sdf_tiles_span<0>), dim3(grid), dim3(vgsdf::TPB), 0, stream, glyphs, tiles, grid | 0x80000000u, sx, sy,
	                   ex, ey, seg_stride, out, (const float4 *)boxes, (unsigned long long *)nullptr, (const vgsdf::PlanHeader *)plan);
	return (int)hipGetLastError();
}

// kernel ids: 50 = the bounded-group span kernel (public variant 0), 1 = brute force, also the fallback
// for the tiles the host routes there: glyphs too wide for the winding histogram, or with >= 2^24
// segments.  Everything else exists only in development builds (-DVGSDF_DEV_VARIANTS).
extern "C" int vgsdf_kernel_known(int kernel)
{
#ifdef VGSDF_DEV_VARIANTS
	switch (kernel) {
	case 10: case 12: case 22: case 23: case 30: case 45: case 51: case 52: case 53: case 54: case 55: case 56: case 57: case 58: case 60: case 61:
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
#define VG_LAUNCH_SPAN(A)                                                                                 \
	hipLaunchKernelGGL((vgsdf::sdf_tiles_span<A>), grid, dim3(vgsdf::TPB), 0, stream, glyphs, tiles, n_tiles,   \
	                   sx, sy, ex, ey, seg_stride, out, (const float4 *)boxes, (unsigned long long *)nullptr,                \
	                   (const vgsdf::PlanHeader *)nullptr)
	if (variant == 1)
		hipLaunchKernelGGL(vgsdf::sdf_tiles_brute, grid, dim3(vgsdf::TPB), 0, stream, glyphs,
		                   tiles, n_tiles, sx, sy, ex, ey, seg_stride, out);
	else if (variant == 50) // bounded groups over spans of up to 4 tiles (tile list: first pixel | T)
		VG_LAUNCH_SPAN(0);
#ifdef VGSDF_DEV_VARIANTS
	// development builds only (`make dev`): timing-only ablations (WRONG pixels) and earlier generations
	else if (seg_stride != 1 && !(variant >= 50 && variant <= 69)) // the earlier generations read SoA arrays only
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
	else if (variant == 51) // timing-only ablations of the span kernel: no phase 2 (and nothing after it)
		VG_LAUNCH_SPAN(2);
	else if (variant == 52) // ... no exact fallback
		VG_LAUNCH_SPAN(4);
	else if (variant == 53) // ... no phase 1 (one candidate group), no phase 2
		VG_LAUNCH_SPAN(34);
	else if (variant == 54) // ... staging only, without the row crossings
		VG_LAUNCH_SPAN(35);
	else if (variant == 55) // ... no row crossings
		VG_LAUNCH_SPAN(1);
	else if (variant == 56) // A/B: per-lane candidate walk instead of the pooled pairs
		VG_LAUNCH_SPAN(64);
	else if (variant == 57) // A/B: no chunk-box skipping
		VG_LAUNCH_SPAN(128);
	else if (variant == 60) // timing-only: pooled filter rounds kept, decide / fallback skipped
		VG_LAUNCH_SPAN(1024);
	else if (variant == 61) // timing-only: pair queue built, rounds and decide skipped
		VG_LAUNCH_SPAN(512);
	else if (variant == 58) { // diagnostic: s_memtime stamps per phase; prints the shares of wave time on stderr
		static unsigned long long *d_dbg = nullptr;
		if (!d_dbg && hipMalloc(&d_dbg, 24 * sizeof(unsigned long long)) != hipSuccess)
			return (int)hipErrorOutOfMemory;
		(void)hipMemsetAsync(d_dbg, 0, 24 * sizeof(unsigned long long), stream);
		hipLaunchKernelGGL((vgsdf::sdf_tiles_span<256>), grid, dim3(vgsdf::TPB), 0, stream, glyphs, tiles, n_tiles, sx, sy, ex,
		                   ey, seg_stride, out, (const float4 *)boxes, d_dbg, (const vgsdf::PlanHeader *)nullptr);
		unsigned long long h[24];
		(void)hipMemcpyAsync(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost, stream);
		(void)hipStreamSynchronize(stream);
		static const char *names[12] = {"overhead", "wait chunk-top barrier", "stage", "wait post-stage barrier", "group bounds",
		                                "wait post-bounds barrier", "phase 1", "phase 2", "decide", "exact fallback",
		                                "quantise+state", "epilogue"};
		unsigned long long tot = 0;
		for (int r = 0; r < 12; r++)
			tot += h[r];
		std::fprintf(stderr, "[vgsdf stamps] %llu waves, %.0f ticks per wave\n", h[12], h[12] ? (double)tot / (double)h[12] : 0.0);
		for (int r = 0; r < 12; r++)
			std::fprintf(stderr, "[vgsdf stamps]   %-26s %6.2f %%\n", names[r], tot ? 100.0 * (double)h[r] / (double)tot : 0.0);
		std::fprintf(stderr, "[vgsdf counts] wave tile-chunks %llu, pairs %llu (%.1f per wave tile-chunk), rounds %llu (%.2f), wave fallback events %llu (%.3f per wave tile-chunk), undecided lanes %llu\n",
		             h[15], h[13], h[15] ? (double)h[13] / (double)h[15] : 0.0, h[14], h[15] ? (double)h[14] / (double)h[15] : 0.0, h[16],
		             h[15] ? (double)h[16] / (double)h[15] : 0.0, h[17]);
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
