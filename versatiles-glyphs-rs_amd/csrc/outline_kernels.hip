// outline_kernels.hip — device front-end: outline commands ->
// flattened, closed rings -> scaled + shifted segments and the raster rect of every glyph,
// all on the GPU, bit-exact with the host/reference path:
//   RingBuilder          /root/reference/src/render/ring_builder.rs:26-117
//   add_quadratic_bezier / add_cubic_bezier   src/geometry/ring.rs:119-187
//   Ring::close          src/geometry/ring.rs:53-63
//   rings.scale / translate / get_bbox       src/geometry/rings.rs:50-70, point.rs:83-99
//   Renderer::prepare_glyph                  src/render/renderer.rs:64-91
//   Rings::get_segments                      src/geometry/rings.rs:75-81
// The host only parses font tables and records the OutlineBuilder callbacks (f32 font
// units); everything after that is f64 arithmetic replayed here operation by operation
// (-ffp-contract=off, IEEE divide).
//
// Pipeline: five launches, no intermediate point arrays, one read-back (the rects) for the host:
//   context  wave per glyph: is the ring non-empty in front of every command; commands that arrive in the compact
//            upload form (kind byte + the coordinates the kind carries) are expanded into 28-byte records here
//   count    wave per 64 commands: first flattening pass — how many points every command appends, their
//            bounding box (nothing else is stored)
//   rings    wave per glyph: point offsets inside the glyph, ring acceptance / closing rules, segment count,
//            bbox -> rect
//   plan     one workgroup: segment / output offsets of all glyphs (descriptors), the raster's work list
//            (spans, heaviest first), totals
//   emit     wave per 64 commands: second flattening pass; every point goes straight into the scaled + shifted
//            segment records as the start of one segment and the end of its predecessor
// The two flattening passes deal the points of 64 consecutive commands out to the 64 lanes of a wave in items
// of up to 8 points ("Leaf-parallel flattening" below); one command per thread left 63 lanes waiting for the
// lane with a 32-point curve (36 + 45 us per Noto Sans Regular font; now 7 + 21), and a wave-per-glyph form
// measured 2.5x slower still.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <stdint.h>

#include "outline_kernels.h"
#include "sdf_kernels.h"

namespace vgsdf {

constexpr double kTolSq = 0.01; // ring_builder.rs:62 `precision`, passed as tolerance_sq (:91,:108)
constexpr int kMaxStack = 18;   // the work list holds at most this many pending halves; flatness shrinks 4x per level,
                                // so that covers control polygons up to ~4e8 font units
constexpr uint32_t kMaxCurvePoints = 1u << 17;
// Non-finite or absurd control points never become flat (the reference would not terminate, or would emit
// ~2^52 points): a curve stops subdividing once its work list is full or it has emitted kMaxCurvePoints points;
// what is still pending then emits its end point only, so a curve yields at most kMaxCurvePoints + kMaxStack
// points and every loop below ends.  (A full work list alone does not bound the loop: the right spine of the
// subdivision tree is reached with an empty list at any depth.)

// Iterative de Casteljau, explicit LIFO stack, right half pushed first (ring.rs:119-144).
// Calls emit(x, y) for every point appended to the ring, in order.  Returns the count.
//
// The reference pushes (m, m2, e) then (s, m1, m) and pops the left half at once; what stays on its
// stack are the right halves of the ancestors.  A right half's start point is the end point of
// its left sibling's subtree, i.e. the point emitted last (the same f64 value m, carried down
// unchanged), so a pending entry needs only (control, end): 4 doubles.
// These sequential walks are the EXCEPTION (see "Leaf-parallel flattening" below): one lane of a wave at a
// time runs them, on a work list in LDS (kMaxStack entries of 8 doubles per wave).  A private array would live
// in scratch memory: 1.2 KB per lane made the runtime set scratch up per dispatch and cost both flattening
// kernels ~25 us each, whatever they computed.
struct QuadStack {
	double *stk; // [kMaxStack][4]
	__device__ __forceinline__ void push(int n, double cx, double cy, double ex, double ey)
	{
		stk[4 * n + 0] = cx, stk[4 * n + 1] = cy, stk[4 * n + 2] = ex, stk[4 * n + 3] = ey;
	}
	__device__ __forceinline__ void pop(int n, double &cx, double &cy, double &ex, double &ey)
	{
		cx = stk[4 * n + 0], cy = stk[4 * n + 1], ex = stk[4 * n + 2], ey = stk[4 * n + 3];
	}
};
constexpr int kSerialStackDoubles = kMaxStack * 8;

// CAPPED = false is for curves whose subdivision provably ends within 16 levels (see flatten_quad_any): no
// bookkeeping in the loop.  CAPPED = true bounds the loop as described at kMaxCurvePoints.
template <bool CAPPED, class Emit>
__device__ __forceinline__ uint32_t flatten_quad(double sx, double sy, double cx, double cy, double ex, double ey,
                                                 double *stk, Emit emit)
{
	QuadStack st;
	st.stk = stk;
	int n = 0; // pending right halves = the reference's stack size after its pop
	uint32_t count = 0;
	double qsx = sx, qsy = sy, qcx = cx, qcy = cy, qex = ex, qey = ey;
	for (;;) {
		const double dx = qsx + qex - qcx * 2.0; // ring.rs:129
		const double dy = qsy + qey - qcy * 2.0;
		// (non-finite control points never become flat: the reference would not terminate;
		// here the work list is bounded and the end point is emitted)
		if (dx * dx + dy * dy <= kTolSq || (CAPPED && (n + 2 > kMaxStack || count >= kMaxCurvePoints))) {
			emit(qex, qey);
			count++;
			if (n == 0)
				break;
			qsx = qex, qsy = qey; // the right half starts where its left sibling ended
			n--;
			st.pop(n, qcx, qcy, qex, qey);
			continue;
		}
		const double m1x = (qsx + qcx) / 2.0, m1y = (qsy + qcy) / 2.0; // point.rs:29-31
		const double m2x = (qcx + qex) / 2.0, m2y = (qcy + qey) / 2.0;
		const double mx = (m1x + m2x) / 2.0, my = (m1y + m2y) / 2.0;
		st.push(n, m2x, m2y, qex, qey); // right half (m, m2, e): start implied
		n++;
		qcx = m1x, qcy = m1y, qex = mx, qey = my; // left half (s, m1, m)
	}
	return count;
}

// A quadratic's deviation s + e - 2c is quartered by every halving (exactly, up to roundings of the size of an ulp
// of the coordinates), so its square falls 16-fold per level: with the root's at most 0.01 * 16^14 and
// coordinates below 1e9 (ulp 1e-7, far below the tolerance) the loop ends within 16 levels and the work list
// never holds more than 16 entries — every curve of a real font.  Anything else takes the capped loop.
template <class Emit>
__device__ __forceinline__ uint32_t flatten_quad_any(double sx, double sy, double cx, double cy, double ex, double ey,
                                                     double *stk, Emit emit)
{
	const double dx = sx + ex - cx * 2.0, dy = sy + ey - cy * 2.0;
	const double m = fmax(fmax(fmax(fabs(sx), fabs(sy)), fmax(fabs(cx), fabs(cy))), fmax(fabs(ex), fabs(ey)));
	if (dx * dx + dy * dy <= 7.2e14 && m <= 1.0e9) // (false for NaN / inf)
		return flatten_quad<false>(sx, sy, cx, cy, ex, ey, stk, emit);
	return flatten_quad<true>(sx, sy, cx, cy, ex, ey, stk, emit);
}

// ring.rs:159-187
template <class Emit>
__device__ __forceinline__ uint32_t flatten_cubic(double sx, double sy, double ax, double ay, double bx, double by,
                                                  double ex, double ey, double *stk /*[kMaxStack][8], LDS*/, Emit emit)
{
	int n = 0;
	uint32_t count = 0;
	auto put = [&](int at, double v0, double v1, double v2, double v3, double v4, double v5, double v6, double v7) {
		double *e = stk + 8 * at;
		e[0] = v0, e[1] = v1, e[2] = v2, e[3] = v3, e[4] = v4, e[5] = v5, e[6] = v6, e[7] = v7;
	};
	put(n, sx, sy, ax, ay, bx, by, ex, ey);
	n++;
	while (n > 0) {
		n--;
		const double *t = stk + 8 * n;
		const double s0 = t[0], s1 = t[1], a0 = t[2], a1 = t[3], b0 = t[4], b1 = t[5], e0 = t[6], e1 = t[7];
		const double dx = (b0 + a0) - (s0 + e0);
		const double dy = (b1 + a1) - (s1 + e1);
		if (dx * dx + dy * dy <= kTolSq || n + 2 > kMaxStack || count >= kMaxCurvePoints) {
			emit(e0, e1);
			count++;
			continue;
		}
		const double p01x = (s0 + a0) / 2.0, p01y = (s1 + a1) / 2.0;
		const double p12x = (a0 + b0) / 2.0, p12y = (a1 + b1) / 2.0;
		const double p23x = (b0 + e0) / 2.0, p23y = (b1 + e1) / 2.0;
		const double p012x = (p01x + p12x) / 2.0, p012y = (p01y + p12y) / 2.0;
		const double p123x = (p12x + p23x) / 2.0, p123y = (p12y + p23y) / 2.0;
		const double mx = (p012x + p123x) / 2.0, my = (p012y + p123y) / 2.0;
		put(n, mx, my, p123x, p123y, p23x, p23y, e0, e1);
		n++;
		put(n, s0, s1, p01x, p01y, p012x, p012y, mx, my);
		n++;
	}
	return count;
}

// One command.  `ring_open` is false right after a CLOSE / at the glyph start, where the
// ring is empty and quad_to / curve_to are ignored (ring_builder.rs:83-85,99-101).
template <class Emit>
__device__ __forceinline__ uint32_t run_command(const OutlineCmd &c, bool ring_open, double lastx, double lasty, double *stk,
                                                Emit emit)
{
	switch (c.kind) {
	case CMD_MOVE: // ring_builder.rs:69-72 (save_ring happens in the ring pass)
	case CMD_LINE: // :75-77
		emit((double)c.x, (double)c.y);
		return 1;
	case CMD_QUAD: // :82-93
		if (!ring_open)
			return 0;
		return flatten_quad_any(lastx, lasty, (double)c.x1, (double)c.y1, (double)c.x, (double)c.y, stk, emit);
	case CMD_CURVE: // :98-110
		if (!ring_open)
			return 0;
		return flatten_cubic(lastx, lasty, (double)c.x1, (double)c.y1, (double)c.x2, (double)c.y2, (double)c.x, (double)c.y, stk, emit);
	default: // CMD_CLOSE
		return 0;
	}
}


// ring state in front of the 64 commands of one step of a wave: empty at the glyph start and after close();
// move_to and line_to make it non-empty; decided by the nearest earlier state-changing command (two ballots)
__device__ __forceinline__ bool ring_open_before(uint32_t kind, uint32_t lane, bool &carry)
{
	const unsigned long long opens = __ballot(kind == CMD_MOVE || kind == CMD_LINE);
	const unsigned long long closes = __ballot(kind == CMD_CLOSE);
	const unsigned long long before = (opens | closes) & ((1ull << lane) - 1ull);
	bool open = carry;
	if (before) {
		const int top = 63 - __builtin_clzll(before);
		open = (opens >> top) & 1ull;
	}
	const unsigned long long all = opens | closes;
	if (all) {
		const int top = 63 - __builtin_clzll(all);
		carry = (opens >> top) & 1ull;
	}
	return open;
}

// exclusive prefix sum over the wave (DPP row shifts + row broadcasts); total = sum over all lanes
__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v, uint32_t &total)
{
	uint32_t incl = v;
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false); // row_shr:1
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false); // row_shr:2
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false); // row_shr:4
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false); // row_shr:8
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
	incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
	total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	return incl - v;
}

// Rust `as i32` on f64: truncate, saturate, NaN -> 0
__device__ __forceinline__ int32_t to_i32(double v)
{
	if (v != v)
		return 0;
	if (v >= 2147483647.0)
		return 2147483647;
	if (v <= -2147483648.0)
		return (int32_t)0x80000000;
	return (int32_t)v;
}


// ---------------------------------------------------------------------------------------
// Leaf-parallel flattening of the 64 commands of a wave.
//
// The deviation s + e - 2c of a quadratic is quartered by every halving, exactly in real arithmetic: both halves
// of (s, c, e) have deviation (s + e - 2c) / 4.  So the reference's adaptive subdivision (ring.rs:119-144) is a
// COMPLETE binary tree of depth L = the first level with D / 16^L <= tolerance^2 (D = |s + e - 2c|^2 at the root),
// unless a rounding error pushes some node across the threshold.  The computed deviation of a node at level
// l <= 16 differs from d / 4^l by < (8 l + 8) 2^-52 M < 3e-14 M (every midpoint (a + b) / 2 rounds once, all
// points are convex combinations of the control points, |coordinate| <= M), so for M <= 1e6 the squared
// deviation near the threshold (|d| ~ 0.1) is off by < 6e-9 < 1e-6 * tolerance^2: when D / 16^L lies outside
// (1 +- 1e-6) tolerance^2 at levels L - 1 and L, the tree IS complete, the command appends exactly 2^L points
// and point r is the end point of the leaf reached from the root by the bits of r (most significant first) —
// the same midpoint operations on the same f64 values as the sequential walk, in any order.  (All 323 k
// quadratics of the 21 fixture fonts qualify; L is 4 or 5 for most.)
//
// A cubic's tree is adaptive, but its depth is bounded: with the second differences D1 = s - 2a + b, D2 = a - 2b + e
// the halves have (D1 / 4, (D1 + D2) / 8) and ((D1 + D2) / 8, D2 / 4), so M = max |component| of them falls at
// least 4-fold per level, and the reference's flatness measure (b + a) - (s + e) = -(D1 + D2) has a squared
// length <= 8 M^2: every node at depth Dmax = the first level with 16^Dmax >= 1.01 * 800 M0^2 is flat.  The
// 2^Dmax <= 64 "candidate" leaves of the complete tree of that depth are dealt out like a quadratic's points;
// the lane of a candidate item walks down from the root with the reference's own flatness test at every node
// and emits each adaptive leaf once, at the leaf's FIRST candidate.  A 64-bit mask of those first candidates per
// command (first flattening pass -> second) turns a candidate into the point's index: the number of mask bits
// below it.
//
// Everything else — a quadratic inside the margin, a cubic deeper than 6 levels, absurd coordinates, commands of a
// glyph whose transform is not monotone — keeps the sequential walk, one lane of the wave at a time.
//
// A wave therefore deals the points of its 64 commands out to its 64 lanes in ITEMS of up to 8 consecutive
// points (a subtree of depth <= 3; a line is an item of one point), 64 items per round: a curve of 32 points is
// four items and no longer keeps 63 lanes waiting, and the walk down to a subtree's root is shared by its 8 leaves.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kCubicFlag = 0x80000000u;
struct WaveQuads {
	double sx[64], sy[64], cx[64], cy[64], ex[64], ey[64]; // quadratic: s, c, e; cubic: s, a, e
	double bx[64], by[64];                                 // cubic: second control point
	uint32_t pre[65];             // exclusive sums of the commands' items
	uint32_t lev[64];             // depth of the (candidate) tree; kCubicFlag: cubic
	unsigned long long mask[64];  // cubic: first candidates of the adaptive leaves
};

struct CubicNode {
	double s0, s1, a0, a1, b0, b1, e0, e1;
};
__device__ __forceinline__ bool cubic_flat(const CubicNode &n)
{
	const double dx = (n.b0 + n.a0) - (n.s0 + n.e0); // ring.rs:171-172
	const double dy = (n.b1 + n.a1) - (n.s1 + n.e1);
	return dx * dx + dy * dy <= kTolSq;
}
__device__ __forceinline__ void cubic_split(const CubicNode &n, CubicNode &l, CubicNode &r)
{
	const double p01x = (n.s0 + n.a0) / 2.0, p01y = (n.s1 + n.a1) / 2.0; // ring.rs:176-182
	const double p12x = (n.a0 + n.b0) / 2.0, p12y = (n.a1 + n.b1) / 2.0;
	const double p23x = (n.b0 + n.e0) / 2.0, p23y = (n.b1 + n.e1) / 2.0;
	const double p012x = (p01x + p12x) / 2.0, p012y = (p01y + p12y) / 2.0;
	const double p123x = (p12x + p23x) / 2.0, p123y = (p12y + p23y) / 2.0;
	const double mx = (p012x + p123x) / 2.0, my = (p012y + p123y) / 2.0;
	l = CubicNode{n.s0, n.s1, p01x, p01y, p012x, p012y, mx, my};
	r = CubicNode{mx, my, p123x, p123y, p23x, p23y, n.e0, n.e1};
}
// depth bound of a cubic's adaptive tree (see above); false: not for the parallel rounds
__device__ __forceinline__ bool cubic_parallel_depth(const CubicNode &n, uint32_t &depth)
{
	const double d1x = n.s0 - 2.0 * n.a0 + n.b0, d1y = n.s1 - 2.0 * n.a1 + n.b1;
	const double d2x = n.a0 - 2.0 * n.b0 + n.e0, d2y = n.a1 - 2.0 * n.b1 + n.e1;
	const double M = fmax(fmax(fabs(d1x), fabs(d1y)), fmax(fabs(d2x), fabs(d2y)));
	const double m = fmax(fmax(fmax(fabs(n.s0), fabs(n.s1)), fmax(fabs(n.a0), fabs(n.a1))),
	                      fmax(fmax(fabs(n.b0), fabs(n.b1)), fmax(fabs(n.e0), fabs(n.e1))));
	depth = 0;
	if (!(M <= 1.0e6 && m <= 1.0e6)) // (false for NaN / inf)
		return false;
	const double need = 1.01 * 800.0 * M * M;
	double v = 1.0;
	uint32_t D = 0;
	while (v < need) {
		v *= 16.0;
		if (++D > 6)
			return false;
	}
	depth = D;
	return true;
}
// the adaptive leaves of the subtree under `n` that spans the candidates [c, c + 2^DL): emit(first candidate, end point)
template <int DL, class Emit>
__device__ __forceinline__ void cubic_subtree(const CubicNode &n, uint32_t c, bool &bound_broken, Emit &emit)
{
	if constexpr (DL == 0) {
		bound_broken |= !cubic_flat(n); // (cannot happen: the depth bound; the batch is refused if it does)
		emit(c, n.e0, n.e1);
	} else {
		if (cubic_flat(n)) {
			emit(c, n.e0, n.e1);
			return;
		}
		CubicNode l, r;
		cubic_split(n, l, r);
		cubic_subtree<DL - 1>(l, c, bound_broken, emit);
		cubic_subtree<DL - 1>(r, c + (1u << (DL - 1)), bound_broken, emit);
	}
}

// points this command hands to the parallel rounds (0: none or sequential), its tree depth
__device__ __forceinline__ uint32_t quad_parallel_points(double sx, double sy, double cx, double cy, double ex, double ey, uint32_t &lev)
{
	const double dx = sx + ex - cx * 2.0, dy = sy + ey - cy * 2.0; // ring.rs:129
	const double D = dx * dx + dy * dy;
	const double m = fmax(fmax(fmax(fabs(sx), fabs(sy)), fmax(fabs(cx), fabs(cy))), fmax(fabs(ex), fabs(ey)));
	lev = 0;
	if (!(D <= 7.2e14 && m <= 1.0e6)) // (false for NaN / inf)
		return 0;
	double v = D;
	uint32_t L = 0;
	while (v > kTolSq) { // <= 14 trips
		v *= 0.0625;
		L++;
	}
	const bool certain = v <= kTolSq * (1.0 - 1.0e-6) && (L == 0 || v * 16.0 > kTolSq * (1.0 + 1.0e-6));
	if (!certain)
		return 0;
	lev = L;
	return 1u << L;
}

// the complete subtree of depth D under (s, c, e), leaves in order: emit(end point of every leaf)
template <int D, class Emit>
__device__ __forceinline__ void quad_subtree(double sx, double sy, double cx, double cy, double ex, double ey, Emit &emit)
{
	if constexpr (D == 0) {
		emit(ex, ey);
	} else {
		const double m1x = (sx + cx) / 2.0, m1y = (sy + cy) / 2.0; // point.rs:29-31
		const double m2x = (cx + ex) / 2.0, m2y = (cy + ey) / 2.0;
		const double mx = (m1x + m2x) / 2.0, my = (m1y + m2y) / 2.0;
		quad_subtree<D - 1>(sx, sy, m1x, m1y, mx, my, emit); // left half (s, m1, m)
		quad_subtree<D - 1>(mx, my, m2x, m2y, ex, ey, emit); // right half (m, m2, e)
	}
}

constexpr uint32_t kItemDepth = 3; // an item = a subtree of up to 2^3 leaves

// The parallel rounds of one wave.  `n_items` / `lev` / control points: this lane's command (n_items = 0: it takes
// no part; for a cubic lev carries kCubicFlag and (cx, cy) / (bx, by) are its two control points).
// point(owner lane, index of the point inside its command, x, y) is called once per point of a line or quadratic,
// cubic_point(owner lane, first candidate of the leaf, x, y) once per leaf of a cubic, in order inside an item;
// done(owner lane) after the last point of an item.  Returns false if a cubic broke its depth bound.
template <class Point, class CubicPoint, class Done>
__device__ __forceinline__ bool wave_parallel_points(WaveQuads &w, uint32_t lane, uint32_t n_items, uint32_t lev, double sx, double sy,
                                                     double cx, double cy, double bx, double by, double ex, double ey, Point point,
                                                     CubicPoint cubic_point, Done done)
{
	uint32_t total;
	const uint32_t excl = wave_exclusive_sum(n_items, total);
	w.pre[lane] = excl;
	if (lane == 0)
		w.pre[64] = total;
	w.lev[lane] = lev;
	w.sx[lane] = sx, w.sy[lane] = sy, w.cx[lane] = cx, w.cy[lane] = cy, w.ex[lane] = ex, w.ey[lane] = ey;
	w.bx[lane] = bx, w.by[lane] = by;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	bool bound_broken = false;
	for (uint32_t q = lane; q < total; q += 64) {
		uint32_t k = 0; // the last command with pre[k] <= q (commands without items share their successor's offset)
#pragma unroll
		for (uint32_t step = 32; step > 0; step >>= 1)
			if (w.pre[k + step] <= q)
				k += step;
		const uint32_t lv = w.lev[k], L = lv & ~kCubicFlag, item = q - w.pre[k];
		const uint32_t Db = L < kItemDepth ? L : kItemDepth;
		if (lv & kCubicFlag) {
			CubicNode n{w.sx[k], w.sy[k], w.cx[k], w.cy[k], w.bx[k], w.by[k], w.ex[k], w.ey[k]};
			auto emit = [&](uint32_t c, double x, double y) { cubic_point(k, c, x, y); };
			bool in_item = true;
			// down to the item's root, the bits of `item` most significant first: an ancestor that is flat is a leaf
			// that spans this item and its neighbours; the first of them emits it
			for (uint32_t l = L - Db; l-- > 0;) {
				if (cubic_flat(n)) {
					if ((item & ((2u << l) - 1u)) == 0)
						emit(item << Db, n.e0, n.e1);
					in_item = false;
					break;
				}
				CubicNode lo, hi;
				cubic_split(n, lo, hi);
				n = ((item >> l) & 1u) ? hi : lo;
			}
			if (in_item) {
				const uint32_t c0 = item << Db;
				if (Db == 3)
					cubic_subtree<3>(n, c0, bound_broken, emit);
				else if (Db == 0)
					cubic_subtree<0>(n, c0, bound_broken, emit);
				else if (Db == 2)
					cubic_subtree<2>(n, c0, bound_broken, emit);
				else
					cubic_subtree<1>(n, c0, bound_broken, emit);
			}
			done(k);
			continue;
		}
		double qsx = w.sx[k], qsy = w.sy[k], qcx = w.cx[k], qcy = w.cy[k], qex = w.ex[k], qey = w.ey[k];
		// down to the item's root: the bits of `item`, most significant first
		for (uint32_t l = L - Db; l-- > 0;) {
			const double m1x = (qsx + qcx) / 2.0, m1y = (qsy + qcy) / 2.0; // point.rs:29-31
			const double m2x = (qcx + qex) / 2.0, m2y = (qcy + qey) / 2.0;
			const double mx = (m1x + m2x) / 2.0, my = (m1y + m2y) / 2.0;
			if ((item >> l) & 1u) { // right half (m, m2, e)
				qsx = mx, qsy = my, qcx = m2x, qcy = m2y;
			} else { // left half (s, m1, m)
				qcx = m1x, qcy = m1y, qex = mx, qey = my;
			}
		}
		uint32_t j = item << Db;
		auto emit = [&](double x, double y) {
			point(k, j, x, y);
			j++;
		};
		if (Db == 3)
			quad_subtree<3>(qsx, qsy, qcx, qcy, qex, qey, emit);
		else if (Db == 0)
			quad_subtree<0>(qsx, qsy, qcx, qcy, qex, qey, emit);
		else if (Db == 2)
			quad_subtree<2>(qsx, qsy, qcx, qcy, qex, qey, emit);
		else
			quad_subtree<1>(qsx, qsy, qcx, qcy, qex, qey, emit);
		done(k);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	return !bound_broken;
}

// items a command of 2^depth points / candidates hands to the parallel rounds
__device__ __forceinline__ uint32_t items_of_depth(uint32_t depth) { return depth > kItemDepth ? 1u << (depth - kItemDepth) : 1u; }

// f64 <-> unsigned key with the same order (for integer atomics on LDS)
__device__ __forceinline__ unsigned long long f64_key(double v)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
	return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

// ---------------------------------------------------------------------------------------
// context: one wave per glyph.  Is the ring non-empty when command c arrives?  (quad_to / curve_to are ignored
// on an empty ring, ring_builder.rs:83-85,99-101.)
// ---------------------------------------------------------------------------------------
// PACKED: the commands arrive as one kind byte each plus only the coordinates their kind carries (move / line 2
// floats, quad 4, curve 6, close none: ~12 bytes per command of a TrueType font instead of 28 on the PCIe link);
// this pass also expands them into the OutlineCmd records every later pass reads (`cmds_out`).
template <bool PACKED>
__global__ __launch_bounds__(64) void outline_context(const OutlineCmd *__restrict__ cmds, const uint32_t *__restrict__ cmd_off,
                                                      const double *__restrict__ scale, uint32_t n_glyphs,
                                                      uint8_t *__restrict__ cmd_open, uint32_t *__restrict__ error_flag,
                                                      const uint8_t *__restrict__ kinds, const float *__restrict__ coords,
                                                      const uint32_t *__restrict__ dat_off, OutlineCmd *__restrict__ cmds_out)
{
	const uint32_t g = blockIdx.x, lane = threadIdx.x;
	if (g >= n_glyphs)
		return;
	const uint32_t c0 = cmd_off[g], c1 = cmd_off[g + 1];
	const double sc = scale[g];
	// bit 1: the glyph's transform is not monotone increasing (scale <= 0 or not finite): the count pass then takes
	// the bounding boxes on the transformed points
	const uint8_t odd = (sc > 0.0 && sc < __builtin_huge_val()) ? 0 : 2;
	bool carry = false;
	uint32_t d_at = 0, d_end = 0; // this glyph's coordinates: coords[d_at, d_end)
	if constexpr (PACKED) {
		d_at = dat_off[g];
		d_end = dat_off[g + 1];
	}
	for (uint32_t base = c0; base < c1; base += 64) {
		const uint32_t c = base + lane;
		uint32_t k = 0xFFu;
		if constexpr (PACKED) {
			if (c < c1)
				k = kinds[c];
			const uint32_t nf = k <= CMD_LINE ? 2u : (k == CMD_QUAD ? 4u : (k == CMD_CURVE ? 6u : 0u));
			uint32_t total;
			const uint32_t at = d_at + wave_exclusive_sum(nf, total);
			d_at += total;
			if (c < c1) {
				OutlineCmd o;
				o.x1 = o.y1 = o.x2 = o.y2 = o.x = o.y = 0.0f;
				o.kind = k;
				if (at + nf > d_end) { // the offsets do not match the kinds: nothing is read past the glyph's range
					atomicOr(error_flag, 8u);
					o.kind = CMD_CLOSE;
					k = CMD_CLOSE;
				} else if (nf == 2) {
					o.x = coords[at], o.y = coords[at + 1];
				} else if (nf == 4) {
					o.x1 = coords[at], o.y1 = coords[at + 1], o.x = coords[at + 2], o.y = coords[at + 3];
				} else if (nf == 6) {
					o.x1 = coords[at], o.y1 = coords[at + 1], o.x2 = coords[at + 2], o.y2 = coords[at + 3];
					o.x = coords[at + 4], o.y = coords[at + 5];
				}
				cmds_out[c] = o;
			}
		} else {
			k = c < c1 ? cmds[c].kind : 0xFFu;
		}
		const bool open = ring_open_before(k, lane, carry);
		if (c < c1) {
			cmd_open[c] = (uint8_t)((open ? 1 : 0) | odd);
			if (k > CMD_CLOSE) // not one of the five callbacks: every kernel treats it as a no-op, the batch is refused
				atomicOr(error_flag, 2u);
		}
	}
}

constexpr int kFlattenThreads = 64; // one wave per workgroup

// ---------------------------------------------------------------------------------------
// count: thread per command — the first flattening pass (ring.rs:119-187).  Stores the number of points the
// command appends and the bounding box of those points: of the raw points when the glyph's transform is
// monotone (scale > 0: the ring pass transforms four numbers instead of every point), else of the
// transformed points (point.rs:96-99, 83-86).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFlattenThreads) void outline_count(const OutlineCmd *__restrict__ cmds,
                                                                 const uint8_t *__restrict__ cmd_open, uint32_t n_cmds,
                                                                 const uint32_t *__restrict__ cmd_off, uint32_t n_glyphs,
                                                                 const double *__restrict__ scale,
                                                                 const double *__restrict__ shift_x,
                                                                 uint32_t *__restrict__ counts, double4 *__restrict__ cmd_box,
                                                                 unsigned long long *__restrict__ cmd_mask,
                                                                 uint32_t *__restrict__ error_flag)
{
	__shared__ double s_stack[kSerialStackDoubles];
	__shared__ WaveQuads s_w;
	__shared__ unsigned long long s_box[4][64]; // per command: min x, min y (keys), max x, max y
	const uint32_t lane = threadIdx.x;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = i < n_cmds;
	OutlineCmd cmd;
	cmd.kind = CMD_CLOSE;
	cmd.x = cmd.y = cmd.x1 = cmd.y1 = cmd.x2 = cmd.y2 = 0.0f;
	uint8_t ctx = 0;
	if (valid) {
		cmd = cmds[i];
		ctx = cmd_open[i];
	}
	const bool open = (ctx & 1) != 0;
	// when the ring is open the previous command of the glyph emitted at least one point and ended on its own
	// (x, y): that is the current point of the ring
	const double lx = open ? (double)cmds[i - 1].x : 0.0, ly = open ? (double)cmds[i - 1].y : 0.0;
	const double inf = __builtin_huge_val();
	const bool raw_boxes = (ctx & 2) == 0;
	// parallel rounds: move_to / line_to (one point, a tree of depth 0), the quadratics with a complete tree, the
	// cubics with a depth bound of at most 6
	uint32_t n_par = 0, n_items = 0, lev = 0;
	bool cubic = false;
	if (valid && raw_boxes) {
		if (cmd.kind == CMD_MOVE || cmd.kind == CMD_LINE) {
			n_par = n_items = 1;
		} else if (cmd.kind == CMD_QUAD && open) {
			n_par = quad_parallel_points(lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x, (double)cmd.y, lev);
			n_items = n_par ? items_of_depth(lev) : 0u;
		} else if (cmd.kind == CMD_CURVE && open) {
			const CubicNode root{lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x2, (double)cmd.y2, (double)cmd.x, (double)cmd.y};
			cubic = cubic_parallel_depth(root, lev);
			if (cubic) {
				n_items = items_of_depth(lev);
				lev |= kCubicFlag;
			}
		}
	}
	s_box[0][lane] = s_box[1][lane] = f64_key(inf);
	s_box[2][lane] = s_box[3][lane] = f64_key(-inf);
	s_w.mask[lane] = 0;
	{
		double bx0 = inf, by0 = inf, bx1 = -inf, by1 = -inf; // box of the item being walked
		unsigned long long leaves = 0;                        // first candidates of the cubic leaves of the item
		auto grow = [&](double x, double y) {
			bx0 = fmin(bx0, x); // bbox.rs:64-69 (fmin / fmax skip NaN)
			by0 = fmin(by0, y);
			bx1 = fmax(bx1, x);
			by1 = fmax(by1, y);
		};
		const bool ok = wave_parallel_points(
		    s_w, lane, n_items, lev, lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x2, (double)cmd.y2, (double)cmd.x, (double)cmd.y,
		    [&](uint32_t, uint32_t, double x, double y) { grow(x, y); },
		    [&](uint32_t, uint32_t c, double x, double y) {
			    grow(x, y);
			    leaves |= 1ull << c;
		    },
		    [&](uint32_t k) {
			    atomicMin(&s_box[0][k], f64_key(bx0));
			    atomicMin(&s_box[1][k], f64_key(by0));
			    atomicMax(&s_box[2][k], f64_key(bx1));
			    atomicMax(&s_box[3][k], f64_key(by1));
			    if (leaves)
				    atomicOr(&s_w.mask[k], leaves);
			    bx0 = inf, by0 = inf, bx1 = -inf, by1 = -inf;
			    leaves = 0;
		    });
		if (!ok)
			atomicOr(error_flag, 4u); // (a cubic deeper than its bound: never seen; the batch is refused rather than wrong)
	}
	if (cubic) {
		n_par = (uint32_t)__builtin_popcountll(s_w.mask[lane]);
		cmd_mask[i] = s_w.mask[lane];
	}
	if (valid && n_items != 0) {
		counts[i] = n_par;
		cmd_box[i] = make_double4(key_f64(s_box[0][lane]), key_f64(s_box[1][lane]), key_f64(s_box[2][lane]), key_f64(s_box[3][lane]));
	}
	// sequential walks, one lane at a time (they share the wave's work list in LDS); quad_to / curve_to on an empty
	// ring and close() append nothing and need no walk
	const bool walk = valid && n_items == 0 && (cmd.kind == CMD_MOVE || cmd.kind == CMD_LINE || ((cmd.kind == CMD_QUAD || cmd.kind == CMD_CURVE) && open));
	if (valid && n_items == 0 && !walk) {
		counts[i] = 0;
		cmd_box[i] = make_double4(inf, inf, -inf, -inf);
	}
	for (unsigned long long todo = __ballot(walk); todo; todo &= todo - 1) {
		if (lane != (uint32_t)__builtin_ctzll(todo))
			continue;
		double minx = inf, miny = inf, maxx = -inf, maxy = -inf;
		double sc = 1.0, dx = 0.0;
		if (!raw_boxes) { // rare: find the glyph (last g with cmd_off[g] <= i) for its scale and shift
			uint32_t lo = 0, hi = n_glyphs;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (cmd_off[mid] <= i)
					lo = mid;
				else
					hi = mid;
			}
			sc = scale[lo];
			dx = shift_x[lo];
		}
		counts[i] = run_command(cmd, open, lx, ly, s_stack, [&](double x, double y) {
			if (!raw_boxes) {
				x *= sc;
				y *= sc;
				x += dx;
				y += 0.0;
			}
			minx = fmin(minx, x); // bbox.rs:64-69 (fmin / fmax skip NaN)
			miny = fmin(miny, y);
			maxx = fmax(maxx, x);
			maxy = fmax(maxy, y);
		});
		cmd_box[i] = make_double4(minx, miny, maxx, maxy);
	}
}

// ---------------------------------------------------------------------------------------
// rings: one wave per glyph, lane <-> command, no serial walk:
//   pass 1: exclusive sums of the commands' point counts = every command's first point inside the glyph.
//   pass 2: RingBuilder::save_ring + Ring::close (ring_builder.rs:33-54, ring.rs:53-63).  With `open` = the ring
//     is non-empty in front of a command (context pass): a command STARTS a ring if it is a move_to, or a
//     line_to on an empty ring (ring_builder.rs:69-77); it ENDS the ring collected so far if that ring is open
//     and the command is a move_to or a close (save_ring); the glyph's end saves the last open ring
//     (into_rings, :26-29).  The lane of the ending command writes the ring's record.  A ring's first point is
//     the point of the command that opened it and its last point is the end point of its last command (every
//     command inside an open ring ends on its own (x, y)), so the rules need no stored points.
//   pass 3: bbox of the accepted rings' points = union of their commands' boxes pushed through the transform
//     (scale > 0 and the shift are monotone; for any other scale the count pass took the boxes on the
//     transformed points), then Renderer::prepare_glyph (renderer.rs:64-91, 122-137).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void outline_rings(const OutlineCmd *__restrict__ cmds, const uint32_t *__restrict__ cmd_off,
                                                    const uint8_t *__restrict__ cmd_open, const double *__restrict__ scale,
                                                    const double *__restrict__ shift_x, uint32_t n_glyphs,
                                                    const uint32_t *__restrict__ counts, uint32_t *__restrict__ pt_local,
                                                    const double4 *__restrict__ cmd_box, RingRec *__restrict__ rings,
                                                    uint32_t *__restrict__ cmd_ring, OutlineRect *__restrict__ rects,
                                                    uint32_t *__restrict__ error_flag)
{
	const uint32_t g = blockIdx.x, lane = threadIdx.x;
	if (g >= n_glyphs)
		return;
	const uint32_t c0 = cmd_off[g], c1 = cmd_off[g + 1];
	const double sc = scale[g], dx = shift_x[g];
	const double inf = __builtin_huge_val();
	const bool raw_boxes = sc > 0.0 && sc < inf; // (the count pass made the same decision)
	// What the passes hand each other (point offsets, ring of a command, ring accepted) also stays in LDS for glyphs
	// of up to kRingsLds commands (every glyph of the fixture fonts): no round trip through memory between the passes
	constexpr uint32_t kRingsLds = 1024;
	__shared__ uint32_t s_pt[kRingsLds + 1], s_ring[kRingsLds];
	__shared__ uint8_t s_acc[kRingsLds];
	const bool in_lds = c1 - c0 <= kRingsLds;
	auto pt_at = [&](uint32_t c) { return in_lds ? s_pt[c - c0] : pt_local[c + g]; }; // c in [c0, c1]

	// ---- pass 1 ----
	uint32_t pbase = 0;
	unsigned long long total64 = 0;
	for (uint32_t base = c0; base < c1; base += 64) {
		const uint32_t c = base + lane;
		const bool valid = c < c1;
		const uint32_t cnt = valid ? counts[c] : 0u;
		uint32_t total;
		const uint32_t excl = wave_exclusive_sum(cnt, total);
		if (valid) {
			pt_local[c + g] = pbase + excl; // (one slot per command plus one end slot per glyph)
			if (in_lds)
				s_pt[c - c0] = pbase + excl;
		}
		pbase += total;
		total64 += total;
	}
	if (lane == 0) {
		pt_local[c1 + g] = pbase;
		if (in_lds)
			s_pt[c1 - c0] = pbase;
		if (total64 > (1ull << 28)) // (also keeps the 32-bit point offsets of a glyph exact)
			atomicOr(error_flag, 1u);
	}
	__syncthreads(); // pt_local is read back below (other lanes' entries)

	// ---- pass 2 ----
	auto ring_rec = [&](uint32_t ra, uint32_t rb) {
		RingRec r;
		const uint32_t a = pt_at(ra), b = pt_at(rb); // points [a, b) of the glyph
		r.pt_first = a;
		r.pt_count = b - a;
		r.append = 0;
		r.accepted = 0;
		r.seg_local = 0;
		r.glyph = g;
		if (r.pt_count >= 3) { // ring_builder.rs:35-38
			const double fx = (double)cmds[ra].x, fy = (double)cmds[ra].y, lx = (double)cmds[rb - 1].x, ly = (double)cmds[rb - 1].y;
			const double eps = 2.220446049250313e-16;
			r.append = (fabs(fx - lx) > eps || fabs(fy - ly) > eps) ? 1u : 0u; // ring.rs:60-62
			if (r.pt_count + r.append >= 4)                                      // ring_builder.rs:45-48
				r.accepted = 1;
		}
		return r;
	};
	uint32_t last_start = 0xFFFFFFFFu; // the command that opened the ring being collected (wave-uniform carry)
	bool open_after = false;          // ring non-empty behind the last command seen
	uint32_t seg_base = 0, n_rings_acc = 0;
	for (uint32_t base = c0; base < c1; base += 64) {
		const uint32_t c = base + lane;
		const bool valid = c < c1;
		const uint32_t k = valid ? cmds[c].kind : 0xFFu;
		const bool open = valid && (cmd_open[c] & 1);
		const bool starts = k == CMD_MOVE || (k == CMD_LINE && !open);
		const bool ends_cmd = open && (k == CMD_MOVE || k == CMD_CLOSE);
		const bool have_after = k == CMD_MOVE || k == CMD_LINE || (valid && k != CMD_CLOSE && open);
		// nearest ring start at or before this lane (strictly before for the ring this command ends)
		const unsigned long long sb = __ballot(starts);
		const unsigned long long below = sb & ((1ull << lane) - 1ull);
		const uint32_t start_before = below ? base + (uint32_t)(63 - __builtin_clzll(below)) : last_start;
		const uint32_t start_here = starts ? c : start_before;
		// (an "open" ring has a command that opened it — context bytes that say otherwise would index ring 0xFFFFFFFF)
		const bool ends = ends_cmd && start_before != 0xFFFFFFFFu;
		if (valid) {
			cmd_ring[c] = have_after ? start_here : 0xFFFFFFFFu;
			if (in_lds)
				s_ring[c - c0] = have_after ? start_here : 0xFFFFFFFFu;
		}
		RingRec rec;
		rec.accepted = 0;
		uint32_t segs = 0;
		if (ends) {
			rec = ring_rec(start_before, c);
			if (rec.accepted)
				segs = rec.pt_count + rec.append - 1;
		}
		uint32_t tot_segs, tot_acc;
		const uint32_t seg_excl = wave_exclusive_sum(segs, tot_segs);
		(void)wave_exclusive_sum(ends ? rec.accepted : 0u, tot_acc);
		if (ends) {
			rec.seg_local = seg_base + seg_excl;
			rings[start_before] = rec; // ring records live at the index of the command that opened them
			if (in_lds)
				s_acc[start_before - c0] = (uint8_t)rec.accepted;
		}
		seg_base += tot_segs;
		n_rings_acc += tot_acc;
		if (sb)
			last_start = base + (uint32_t)(63 - __builtin_clzll(sb));
		const unsigned long long ha = __ballot(have_after);
		const uint32_t n_valid = min(64u, c1 - base);
		open_after = (ha >> (n_valid - 1)) & 1ull;
	}
	if (open_after && lane == 0 && last_start != 0xFFFFFFFFu) { // into_rings: the ring still open at the end of the glyph
		RingRec rec = ring_rec(last_start, c1);
		rec.seg_local = seg_base;
		if (rec.accepted) {
			seg_base += rec.pt_count + rec.append - 1;
			n_rings_acc++;
		}
		rings[last_start] = rec;
		if (in_lds)
			s_acc[last_start - c0] = (uint8_t)rec.accepted;
	}
	seg_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_base);
	n_rings_acc = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_rings_acc);
	__syncthreads(); // cmd_ring / rings are read back below

	// ---- pass 3 ----
	double minx = inf, miny = inf, maxx = -inf, maxy = -inf;
	for (uint32_t c = c0 + lane; c < c1; c += 64) {
		const uint32_t rs = in_lds ? s_ring[c - c0] : cmd_ring[c];
		if (rs != 0xFFFFFFFFu && (in_lds ? s_acc[rs - c0] != 0 : rings[rs].accepted != 0)) {
			const double4 bx = cmd_box[c];
			minx = fmin(minx, bx.x);
			miny = fmin(miny, bx.y);
			maxx = fmax(maxx, bx.z);
			maxy = fmax(maxy, bx.w);
		}
	}
	if (raw_boxes) {
		minx *= sc, miny *= sc, maxx *= sc, maxy *= sc; // point.rs:96-99
		minx += dx, maxx += dx;                         // point.rs:83-86
		miny += 0.0, maxy += 0.0;
	}
	for (int sh = 32; sh > 0; sh >>= 1) { // fmin/fmax are exact selections: any order gives the same box
		minx = fmin(minx, __shfl_xor(minx, sh));
		miny = fmin(miny, __shfl_xor(miny, sh));
		maxx = fmax(maxx, __shfl_xor(maxx, sh));
		maxy = fmax(maxy, __shfl_xor(maxy, sh));
	}
	if (lane == 0) {
		const uint32_t nseg = seg_base, n_rings = n_rings_acc;
		OutlineRect rc;
		rc.x0 = rc.y0 = 0;
		rc.w = rc.h = 0;
		rc.n_segments = 0;
		rc.has_raster = 0;
		// renderer.rs:118-120 (no rings) and :133-137 / bbox.rs:56-58 (empty bbox) -> PbfGlyph::empty
		if (n_rings > 0 && !(maxx <= minx && maxy <= miny)) {
			const int32_t x0 = to_i32(floor(minx)) - 3, y0 = to_i32(floor(miny)) - 3; // renderer.rs:73-76, BUFFER = 3
			const int32_t x1 = to_i32(ceil(maxx)) + 3, y1 = to_i32(ceil(maxy)) + 3;
			rc.x0 = x0;
			rc.y0 = y0;
			rc.w = (uint32_t)(x1 - x0);
			rc.h = (uint32_t)(y1 - y0);
			rc.n_segments = nseg;
			rc.has_raster = 1;
		}
		rects[g] = rc;
	}
}

// ---------------------------------------------------------------------------------------
// plan: ONE workgroup.  From the rects: the glyph descriptors (segment and output offsets: exclusive sums),
// the raster's work list and the totals the host reads back together with the rects.  The list follows the
// host's policy for resident batches (vgsdf_device.cpp, build_descs_and_tiles): class (main kernel / brute
// force), span length T per glyph, heaviest workgroup first (counting sort over 512 logarithmic weight
// buckets; the order inside a bucket is whatever the atomics give: it only shapes the schedule).
// ---------------------------------------------------------------------------------------
constexpr int kPlanThreads = 1024;
constexpr int kPlanBuckets = 512;

__device__ __forceinline__ uint32_t plan_bucket(uint32_t wgt) // bucket 0 = heaviest
{
	if (wgt < 16u)
		return 511u - wgt;
	const uint32_t e = 31u - (uint32_t)__builtin_clz(wgt);      // 4..31
	return 511u - ((e - 3u) * 16u + ((wgt >> (e - 4u)) & 15u)); // 16..463 -> descending
}

// block-wide exclusive sum of one value per thread (1024 threads = 16 waves); returns the exclusive prefix, sets total
__device__ __forceinline__ unsigned long long block_exclusive_sum(unsigned long long v, unsigned long long *s_wave /*[17]*/,
                                                                  unsigned long long &total)
{
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long incl = v;
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = __shfl_up(incl, d);
		if ((int)lane >= d)
			incl += o;
	}
	__syncthreads(); // s_wave free again
	if (lane == 63)
		s_wave[wv] = incl;
	__syncthreads();
	if (wv == 0) { // the 16 wave totals: one shuffle scan in the first wave
		constexpr int kWaves = kPlanThreads / 64;
		const unsigned long long t = lane < (uint32_t)kWaves ? s_wave[lane] : 0ull;
		unsigned long long in = t;
		for (int d = 1; d < kWaves; d <<= 1) {
			const unsigned long long o = __shfl_up(in, d);
			if ((int)lane >= d)
				in += o;
		}
		if (lane < (uint32_t)kWaves)
			s_wave[lane] = in - t;
		if (lane == (uint32_t)kWaves - 1)
			s_wave[kWaves] = in;
	}
	__syncthreads();
	total = s_wave[kPlanThreads / 64];
	return s_wave[wv] + incl - v;
}

// bytes of a protobuf varint
__device__ __forceinline__ uint32_t varint_len(unsigned long long v)
{
	const uint32_t bits = 64u - (uint32_t)__builtin_clzll(v | 1ull); // 1 .. 64 significant bits, 7 per byte
	return (bits + 6u) / 7u;
}

// In-place PBF assembly (vgsdf.h, vgsdf_outlines_packed::pbf_pre / pbf_fix): what glyph `r` occupies in the arena of
// finished PBF blocks — `pre` reserved bytes in front of it (the block's file and fontstack header when it is the first
// glyph of its block), then its `glyphs` entry of the fontstack message (src/protobuf/glyph.rs:10-41, fontstack.rs:9-25):
//   0x1A varint(msg) | 0x08 varint(id) | [0x12 varint(w h) bitmap] | 0x18 width 0x20 height 0x28 left 0x30 top 0x38 advance
// with width = w - 6, height = h - 6, left = x0 + 3, top = y0 + h - 27 (src/render/result.rs:66-76 after renderer.rs:146).
// fix = (1 + varint_len(id)) | (1 + varint_len(advance)) << 4, from the host (it knows id and advance).
// Returns the bytes taken, sets `bitmap_at` to the bitmap's offset in them.  The host writes the headers with the same
// arithmetic (csrc/host/pbf.hpp, pbf_entry_layout) once the rects are back.
__device__ __forceinline__ unsigned long long pbf_place(const OutlineRect &r, uint32_t pre, uint32_t fix, unsigned long long &bitmap_at)
{
	const uint32_t idlen = fix & 15u, advlen = fix >> 4;
	unsigned long long msg = idlen + advlen;
	const unsigned long long px = r.has_raster ? (unsigned long long)r.w * r.h : 0ull;
	uint32_t bm_hdr = 0;
	if (r.has_raster) {
		bm_hdr = 1u + varint_len(px);
		msg += bm_hdr + px;
		const uint32_t left = (uint32_t)r.x0 + 3u, top = (uint32_t)r.y0 + r.h - 27u; // two's complement, as i32 arithmetic wraps
		const uint32_t zl = (left << 1) ^ (uint32_t)((int32_t)left >> 31), zt = (top << 1) ^ (uint32_t)((int32_t)top >> 31);
		msg += 4u + varint_len(r.w - 6u) + varint_len(r.h - 6u) + varint_len(zl) + varint_len(zt);
	} else {
		msg += 8u; // PbfGlyph::empty: width, height, left, top = 0 (glyph.rs:60-70), one byte each behind its tag
	}
	const uint32_t ent_hdr = 1u + varint_len(msg);
	bitmap_at = (unsigned long long)pre + ent_hdr + idlen + bm_hdr;
	return (unsigned long long)pre + ent_hdr + msg;
}

// KEEP: glyphs per thread whose rects (and classification) stay in registers for all three passes; PBF: in-place placement
// is part of the unrolled passes (instances: <8, false> for packed bitmaps, <4, true> for in-place batches of <= 4096 glyphs;
// eight inlined placements spill the 1024-thread workgroup to scratch: 21 -> 26 us)
template <uint32_t KEEP, bool PBF>
__global__ __launch_bounds__(kPlanThreads) void outline_plan(const OutlineRect *__restrict__ rects, uint32_t n_glyphs, int span_list,
                                                             uint32_t delta_cap, uint32_t span_max, uint32_t span_budget,
                                                             uint32_t tile_cap, GlyphDesc *__restrict__ descs,
                                                             uint2 *__restrict__ tiles, PlanHeader *__restrict__ hdr,
                                                             const uint32_t *__restrict__ error_flag, unsigned long long seg_cap,
                                                             unsigned long long out_cap, uint32_t launch_spans,
                                                             const uint32_t *__restrict__ pbf_pre, const uint8_t *__restrict__ pbf_fix,
                                                             unsigned long long *__restrict__ pbf_at, uint32_t *__restrict__ next_flag)
{
	__shared__ unsigned long long s_wave[kPlanThreads / 64 + 1];
	// the error word of the NEXT submission of this context (the two alternate): zeroed here, behind everything that could
	// still raise the previous use of that word, instead of a memset launch in front of every submission
	if (threadIdx.x == 0 && next_flag != nullptr)
		next_flag[0] = 0;
	__shared__ uint32_t s_hist[2][kPlanBuckets]; // spans per (class, bucket); then the bucket's write cursor
	__shared__ unsigned long long s_carry[2];
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = tid; i < 2 * kPlanBuckets; i += kPlanThreads)
		(&s_hist[0][0])[i] = 0;
	if (tid == 0)
		s_carry[0] = s_carry[1] = 0;
	__syncthreads();
	// rows touched by T consecutive tiles, times the (odd-padded) row stride, must fit the winding histogram
	auto fits = [&](uint32_t w, uint32_t T) { return (unsigned long long)((256u * T - 2u) / w + 2u) * ((unsigned long long)w + 2u) <= delta_cap; };
	// class (0 main / 1 brute), span length and span count of a glyph
	auto classify = [&](const OutlineRect &r, uint32_t &cls, uint32_t &T, uint32_t &nspans, uint32_t &weight) {
		const unsigned long long px = r.has_raster ? (unsigned long long)r.w * r.h : 0ull;
		cls = 0, T = 1, nspans = 0, weight = 0;
		if (px == 0 || px > 0xFFFFFFFFull - 256ull) // (a bitmap beyond 2^32 pixels is an error of the batch: no entries)
			return;
		const uint32_t nseg = r.n_segments;
		if (!fits(r.w, 1) || nseg >= (1u << 24)) {
			cls = 1;
		} else if (span_list) {
			const uint32_t chunks = (nseg + 255u) / 256u;
			const uint32_t t_hi = min(span_max, max(1u, span_budget / max(chunks, 1u)));
			for (T = t_hi; T > 1; T--)
				if (fits(r.w, T))
					break;
		}
		const unsigned long long t256 = (px + 255ull) >> 8;
		nspans = (uint32_t)((t256 + T - 1) / T);
		const unsigned long long wgt = (unsigned long long)nseg * (t256 < T ? t256 : T);
		weight = wgt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)wgt;
	};
	// pass A: descriptors (exclusive sums of segments and output bytes), span histogram.  Every thread takes a
	// contiguous run of glyphs: one block-wide scan of the runs' sums, then a running sum inside the run.
	bool bad = false;
	const uint32_t per = (n_glyphs + kPlanThreads - 1) / kPlanThreads;
	const uint32_t g_lo = min(tid * per, n_glyphs), g_hi = min(g_lo + per, n_glyphs);
	// a run of up to eight glyphs (batches of <= 8192 glyphs: every group the dispatcher forms) stays in registers with its classification, so
	// the rects are read once and classified once for both passes
	constexpr uint32_t kKeep = KEEP;
	const bool kept = per <= kKeep && (PBF || pbf_fix == nullptr);
	OutlineRect kr[kKeep];
	uint32_t kcls[kKeep], kT[kKeep], kn[kKeep], kw[kKeep];
#pragma unroll
	for (uint32_t j = 0; j < kKeep; j++) {
		kr[j].x0 = kr[j].y0 = 0;
		kr[j].w = kr[j].h = kr[j].n_segments = kr[j].has_raster = 0;
		kcls[j] = kT[j] = kn[j] = kw[j] = 0;
		if (kept && g_lo + j < g_hi) {
			kr[j] = rects[g_lo + j];
			classify(kr[j], kcls[j], kT[j], kn[j], kw[j]);
		}
	}
	{
		unsigned long long my_s = 0, my_p = 0;
		// output bytes of a glyph: its bitmap, or — in-place PBF assembly — everything it occupies in the arena
		auto sum_one = [&](uint32_t g, const OutlineRect &r) {
			if (r.has_raster) {
				my_s += r.n_segments;
				bad |= (unsigned long long)r.w * r.h > 0xFFFFFFFFull - 256ull;
			}
			if (pbf_fix != nullptr) {
				unsigned long long at;
				my_p += pbf_place(r, pbf_pre[g], pbf_fix[g], at);
			} else if (r.has_raster) {
				my_p += (unsigned long long)r.w * r.h;
			}
		};
		if (kept) {
#pragma unroll
			for (uint32_t j = 0; j < kKeep; j++) {
				if (kr[j].has_raster) { // (absent glyphs: zero rects)
					my_s += kr[j].n_segments;
					if (!(PBF && pbf_fix != nullptr))
						my_p += (unsigned long long)kr[j].w * kr[j].h;
					bad |= (unsigned long long)kr[j].w * kr[j].h > 0xFFFFFFFFull - 256ull;
				}
				if (PBF && pbf_fix != nullptr && g_lo + j < g_hi) {
					unsigned long long at;
					my_p += pbf_place(kr[j], pbf_pre[g_lo + j], pbf_fix[g_lo + j], at);
				}
			}
		} else {
			for (uint32_t g = g_lo; g < g_hi; g++)
				sum_one(g, rects[g]);
		}
		unsigned long long tot_s, tot_p;
		unsigned long long so = block_exclusive_sum(my_s, s_wave, tot_s);
		unsigned long long po = block_exclusive_sum(my_p, s_wave, tot_p);
		if (tid == 0) {
			s_carry[0] = tot_s;
			s_carry[1] = tot_p;
		}
		auto desc_one = [&](uint32_t g, const OutlineRect &r, uint32_t cls, uint32_t nspans, uint32_t weight, auto in_place) {
			const unsigned long long px = r.has_raster ? (unsigned long long)r.w * r.h : 0ull;
			const unsigned long long segs = r.has_raster ? r.n_segments : 0u;
			GlyphDesc d;
			d.seg_off = (uint32_t)so; // (the host rejects batches with more than 2^32 - 1 segments: header)
			d.n_seg = (uint32_t)segs;
			d.x0 = r.x0;
			d.y0 = r.y0;
			d.w = r.has_raster ? r.w : 0;
			d.h = r.has_raster ? r.h : 0;
			if (decltype(in_place)::value && pbf_fix != nullptr) {
				unsigned long long at;
				const unsigned long long took = pbf_place(r, pbf_pre[g], pbf_fix[g], at);
				d.out_off = po + at; // the raster stores the bitmap where the finished file has it
				pbf_at[g] = po + at; // (read back with the rects: the host writes the bytes around it)
				po += took;
			} else {
				d.out_off = po;
				po += px;
			}
			descs[g] = d;
			so += segs;
			if (nspans)
				atomicAdd(&s_hist[cls][plan_bucket(weight)], nspans);
		};
		if (kept) {
#pragma unroll
			for (uint32_t j = 0; j < kKeep; j++)
				if (g_lo + j < g_hi)
					desc_one(g_lo + j, kr[j], kcls[j], kn[j], kw[j], std::integral_constant<bool, PBF>{});
		} else {
			for (uint32_t g = g_lo; g < g_hi; g++) {
				const OutlineRect r = rects[g];
				uint32_t cls, T, nspans, weight;
				classify(r, cls, T, nspans, weight);
				desc_one(g, r, cls, nspans, weight, std::true_type{});
			}
		}
	}
	__syncthreads();
	// bucket bases: main class first (heaviest bucket first), then the brute-force class — one (class, bucket)
	// cell per thread
	__shared__ uint32_t s_nmain, s_nall;
	static_assert(2 * kPlanBuckets == kPlanThreads, "one histogram cell per thread");
	{
		const unsigned long long mine = (&s_hist[0][0])[tid];
		unsigned long long acc;
		const unsigned long long base_at = block_exclusive_sum(mine, s_wave, acc);
		(&s_hist[0][0])[tid] = (uint32_t)(base_at > 0xFFFFFFFFull ? 0xFFFFFFFFull : base_at);
		if (tid == kPlanBuckets) // first cell of the brute-force class: everything before it is main class
			s_nmain = (uint32_t)(base_at > 0x7FFFFFFFull ? 0x7FFFFFFFull : base_at);
		if (tid == 0)
			s_nall = (uint32_t)(acc > 0x7FFFFFFFull ? 0x80000000ull : acc);
	}
	__syncthreads();
	if (tid == 0) {
		const unsigned long long acc = s_nall;
		hdr->n_segments = s_carry[0];
		hdr->out_bytes = s_carry[1];
		hdr->n_spans = s_nall;
		hdr->n_main = s_nmain;
		hdr->error = error_flag[0] | (((s_carry[0] > 0xFFFFFFFFull) || (acc > 0x7FFFFFFFull)) ? 1u : 0u); // bit 0: sizes, bit 1: unknown command kind
		hdr->ok = 0;
	}
	__syncthreads();
	const bool any_bad = __syncthreads_or(bad);
	if (tid == 0 && any_bad)
		hdr->error |= 1u;
	// no work list for a batch in error (its sizes may be absurd), nor when the list does not fit: the host
	// grows it and runs the plan again
	if (any_bad || error_flag[0] != 0 || s_carry[0] > 0xFFFFFFFFull || s_nall > tile_cap || s_nall > 0x7FFFFFFFu)
		return;
	// (every glyph's entries are in place when the kernel ends: a launch enqueued behind it may read the list)
	if (tid == 0)
		hdr->ok = s_carry[0] <= seg_cap && s_carry[1] <= out_cap && s_nall == s_nmain && s_nmain <= launch_spans;
	// pass B: entries.  One per span of T tiles: (glyph, first pixel | T) in the span list's main class, else
	// (glyph, first pixel).
	auto entries = [&](uint32_t g, const OutlineRect &r, uint32_t cls, uint32_t T, uint32_t nspans, uint32_t weight) {
		if (!nspans)
			return;
		uint32_t at = atomicAdd(&s_hist[cls][plan_bucket(weight)], nspans);
		const unsigned long long px = (unsigned long long)r.w * r.h;
		for (unsigned long long p = 0; p < px; p += 256ull * T) {
			const uint32_t left = (uint32_t)((px - p + 255ull) >> 8);
			tiles[at++] = make_uint2(g, span_list && cls == 0 ? ((uint32_t)p | min(T, left)) : (uint32_t)p);
		}
	};
	if (kept) {
#pragma unroll
		for (uint32_t j = 0; j < kKeep; j++)
			if (g_lo + j < g_hi)
				entries(g_lo + j, kr[j], kcls[j], kT[j], kn[j], kw[j]);
	} else {
		for (uint32_t g = tid; g < n_glyphs; g += kPlanThreads) {
			const OutlineRect r = rects[g];
			uint32_t cls, T, nspans, weight;
			classify(r, cls, T, nspans, weight);
			entries(g, r, cls, T, nspans, weight);
		}
	}
}

// ---------------------------------------------------------------------------------------
// chunk boxes from the commands' boxes (one wave per glyph of more than two chunks).  The raster skips a chunk of 256
// segments whose box is far from a span (sdf_span_kernel.inc); sdf_chunk_boxes takes those boxes from the segments, i.e.
// behind the second flattening pass, on the path every bitmap waits for.  Everything needed for a box that CONTAINS the
// chunk's segments is known once the plan has run: the count pass stored the box of the points every command appends, the
// ring pass where those points sit in their ring, the plan where the glyph's segments start.  Point q of an accepted ring
// (ring-local) starts segment q and ends segment q - 1; with the appended copy of point 0 (Ring::close, ring.rs:53-63)
// point 0 also ends the ring's last segment.  So a command whose points are [qa, qb) touches the ring's segments
// max(qa - 1, 0) .. min(qb - 1, last), plus the last one if it holds point 0 of a closed-by-append ring: its box goes into
// the chunks of those segments.  Both end points of every segment are covered that way, so the union is a superset of the
// exact box (by a command's extent, a few dozen points): chunks are skipped a little less often, never wrongly.  Boxes are
// pushed through the glyph's transform exactly as the ring pass does (monotone for scale > 0; for any other scale the count
// pass took them on the transformed points), then made relative to (x0, y0) and rounded to f32 like sdf_chunk_boxes' — the
// raster's pad covers the rounding.  The first `n_box_glyphs` workgroups of outline_emit_segments' grid do this, one glyph each,
// beside the second flattening pass: no launch and no dependency of its own (a separate stream for it cost what it saved).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kBoxLdsChunks = 256; // glyphs of more chunks (> 65 536 segments) get boxes that contain everything

__device__ __forceinline__ void chunk_boxes_of_glyph(uint32_t g, uint32_t lane, uint32_t (*s_key)[kBoxLdsChunks] /* [4]: min x, min y, max x, max y
                                                     as order-preserving keys of the f32 values */,
                                                     const uint32_t *__restrict__ cmd_off, const double *__restrict__ scale,
                                                     const double *__restrict__ shift_x, const uint32_t *__restrict__ pt_local,
                                                     const RingRec *__restrict__ rings, const uint32_t *__restrict__ cmd_ring,
                                                     const double4 *__restrict__ cmd_box, const GlyphDesc *__restrict__ descs,
                                                     float4 *__restrict__ boxes)
{
	const GlyphDesc d = descs[g];
	if (d.n_seg <= 2u * 256u) // (the raster asks for boxes only when a glyph has more than two chunks)
		return;
	const uint32_t n_chunks = (d.n_seg + 255u) >> 8;
	const float inf = __builtin_inff();
	if (n_chunks > kBoxLdsChunks) {
		for (uint32_t c = lane; c < n_chunks; c += 64u)
			boxes[chunk_box_index(d.seg_off, g, c)] = make_float4(-inf, -inf, inf, inf);
		return;
	}
	auto key = [](float v) { // f32 -> u32 with the same order
		const uint32_t b = __float_as_uint(v);
		return (b >> 31) ? ~b : (b | 0x80000000u);
	};
	auto unkey = [](uint32_t k) { return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k); };
	for (uint32_t c = lane; c < n_chunks; c += 64u) {
		s_key[0][c] = s_key[1][c] = key(inf);
		s_key[2][c] = s_key[3][c] = key(-inf);
	}
	__syncthreads();
	const uint32_t c0 = cmd_off[g], c1 = cmd_off[g + 1];
	const double sc = scale[g], dx = shift_x[g];
	const bool raw_boxes = sc > 0.0 && sc < __builtin_huge_val(); // (as the count and ring passes decide)
	for (uint32_t c = c0 + lane; c < c1; c += 64u) {
		const uint32_t rs = cmd_ring[c];
		if (rs == 0xFFFFFFFFu)
			continue;
		const RingRec r = rings[rs];
		const uint32_t a = pt_local[c + g], b = pt_local[c + 1 + g]; // the command's points inside the glyph
		if (!r.accepted || b <= a || a < r.pt_first)
			continue;
		const uint32_t last = r.pt_count + r.append - 2u; // last segment of the ring (accepted: pt_count + append >= 4)
		const uint32_t qa = a - r.pt_first, qb = b - r.pt_first;
		const uint32_t s_lo = qa ? qa - 1u : 0u, s_hi = min(qb - 1u, last);
		double4 bx = cmd_box[c];
		if (raw_boxes) {
			bx.x *= sc, bx.y *= sc, bx.z *= sc, bx.w *= sc; // point.rs:96-99
			bx.x += dx, bx.z += dx;                         // point.rs:83-86
			bx.y += 0.0, bx.w += 0.0;
		}
		float x0 = (float)(bx.x - (double)d.x0), y0 = (float)(bx.y - (double)d.y0);
		float x1 = (float)(bx.z - (double)d.x0), y1 = (float)(bx.w - (double)d.y0);
		if (!(fabsf(x0) < 3.0e38f) || !(fabsf(y0) < 3.0e38f) || !(fabsf(x1) < 3.0e38f) || !(fabsf(y1) < 3.0e38f))
			x0 = y0 = -inf, x1 = y1 = inf; // non-finite: the chunks it touches are never skipped
		auto put = [&](uint32_t s_first, uint32_t s_last) {
			const uint32_t k_lo = (r.seg_local + s_first) >> 8, k_hi = min((r.seg_local + s_last) >> 8, n_chunks - 1u);
			for (uint32_t k = k_lo; k <= k_hi; k++) {
				atomicMin(&s_key[0][k], key(x0));
				atomicMin(&s_key[1][k], key(y0));
				atomicMax(&s_key[2][k], key(x1));
				atomicMax(&s_key[3][k], key(y1));
			}
		};
		if (s_lo <= s_hi)
			put(s_lo, s_hi);
		if (qa == 0u && r.append)
			put(last, last); // point 0 ends the segment Ring::close appended
	}
	__syncthreads();
	for (uint32_t c = lane; c < n_chunks; c += 64u) {
		float4 o = make_float4(unkey(s_key[0][c]), unkey(s_key[1][c]), unkey(s_key[2][c]), unkey(s_key[3][c]));
		if (!(o.x <= o.z) || !(o.y <= o.w)) // nothing arrived (cannot happen: every segment has end points): never skip
			o = make_float4(-inf, -inf, inf, inf);
		boxes[chunk_box_index(d.seg_off, g, c)] = o;
	}
}

// ---------------------------------------------------------------------------------------
// emit: thread per command — the second flattening pass.  Point i of a ring of n points is the start of segment i
// (i <= n - 2, or i == n - 1 when Ring::close appended the first point again) and the end of segment i - 1; with
// the appended point, point 0 also ends segment n - 1 (Rings::get_segments, rings.rs:75-81) — written straight
// into the segment records {sx, sy, ex, ey} after scale + translate (renderer.rs:122-131); the raster reads
// them with a stride of four doubles.  A command that belongs to a ring has the
// ring open in front of it unless it opened the ring itself (move_to / line_to, which do not care).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFlattenThreads) void outline_emit_segments(const OutlineCmd *__restrict__ cmds, uint32_t n_cmds,
                                                                         const uint8_t *__restrict__ cmd_open,
                                                                         const double *__restrict__ scale,
                                                                         const double *__restrict__ shift_x,
                                                                         const uint32_t *__restrict__ pt_local,
                                                                         const RingRec *__restrict__ rings,
                                                                         const uint32_t *__restrict__ cmd_ring,
                                                                         const GlyphDesc *__restrict__ descs,
                                                                         const PlanHeader *__restrict__ hdr,
                                                                         unsigned long long seg_cap, double2 *__restrict__ seg,
                                                                         const unsigned long long *__restrict__ cmd_mask,
                                                                         uint32_t n_box_glyphs, const uint32_t *__restrict__ cmd_off,
                                                                         const double4 *__restrict__ cmd_box, float4 *__restrict__ boxes)
{
	__shared__ double s_stack[kSerialStackDoubles];
	__shared__ WaveQuads s_w;
	__shared__ uint32_t s_key[4][kBoxLdsChunks];
	// per command, for the lanes that place its points: first record of its ring, index of its first point inside
	// the ring, the ring's point count and closing flag, the glyph's transform
	__shared__ unsigned long long s_seg0[64];
	__shared__ uint32_t s_idx0[64], s_n[64], s_append[64];
	__shared__ double s_sc[64], s_dx[64];
	if (hdr->error || hdr->n_segments > seg_cap) // nothing may be written: the host grows the arrays and launches again
		return;
	const uint32_t lane = threadIdx.x;
	// the first n_box_glyphs workgroups: chunk boxes of one glyph each (see chunk_boxes_of_glyph), first so that the longest
	// of them are under way when the flattening workgroups fill the chip
	if (blockIdx.x < n_box_glyphs) {
		chunk_boxes_of_glyph(blockIdx.x, lane, s_key, cmd_off, scale, shift_x, pt_local, rings, cmd_ring, cmd_box, descs, boxes);
		return;
	}
	const uint32_t c = (blockIdx.x - n_box_glyphs) * blockDim.x + threadIdx.x;
	bool active = c < n_cmds;
	uint32_t rcmd = 0xFFFFFFFFu;
	RingRec r;
	r.accepted = 0, r.append = 0, r.pt_count = 0, r.pt_first = 0, r.seg_local = 0, r.glyph = 0;
	GlyphDesc d;
	d.n_seg = 0, d.seg_off = 0;
	if (active) {
		rcmd = cmd_ring[c];
		active = rcmd != 0xFFFFFFFFu;
	}
	if (active) {
		r = rings[rcmd];
		active = r.accepted != 0;
	}
	if (active) {
		d = descs[r.glyph];
		active = d.n_seg != 0; // else PbfGlyph::empty (empty bbox): nothing is rasterised
	}
	OutlineCmd cmd;
	cmd.kind = CMD_CLOSE;
	cmd.x = cmd.y = cmd.x1 = cmd.y1 = cmd.x2 = cmd.y2 = 0.0f;
	double sc = 1.0, dx = 0.0;
	uint32_t idx = 0;
	bool monotone = false;
	if (active) {
		cmd = cmds[c];
		sc = scale[r.glyph], dx = shift_x[r.glyph];
		idx = pt_local[c + r.glyph] - r.pt_first; // position of the command's first point inside its ring
		monotone = (cmd_open[c] & 2) == 0;       // (the count pass took the sequential walk for the other glyphs)
	}
	const bool open = active && c != rcmd;
	const double lx = open ? (double)cmds[c - 1].x : 0.0, ly = open ? (double)cmds[c - 1].y : 0.0;
	const size_t seg0 = (size_t)d.seg_off + r.seg_local;
	// point `at` of a ring of n points (scaled + shifted): start of segment `at`, end of segment `at - 1`
	auto place = [&](size_t base, uint32_t at, uint32_t n, uint32_t append, double x, double y) {
		// records {sx, sy, ex, ey}: two 16-byte stores per point, side by side in memory
		if (at + 1 < n || append) // start of segment at
			seg[2 * (base + at)] = make_double2(x, y);
		if (at >= 1) // end of segment at - 1
			seg[2 * (base + at - 1) + 1] = make_double2(x, y);
		else if (append) // the closing segment (n - 1) returns to the first point
			seg[2 * (base + n - 1) + 1] = make_double2(x, y);
	};
	uint32_t n_items = 0, lev = 0;
	unsigned long long mask = 0;
	if (active && monotone) { // the same split as in the count pass
		if (cmd.kind == CMD_MOVE || cmd.kind == CMD_LINE) {
			n_items = 1;
		} else if (cmd.kind == CMD_QUAD && open) {
			if (quad_parallel_points(lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x, (double)cmd.y, lev))
				n_items = items_of_depth(lev);
		} else if (cmd.kind == CMD_CURVE && open) {
			const CubicNode root{lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x2, (double)cmd.y2, (double)cmd.x, (double)cmd.y};
			if (cubic_parallel_depth(root, lev)) {
				n_items = items_of_depth(lev);
				lev |= kCubicFlag;
				mask = cmd_mask[c]; // first candidates of the leaves, from the count pass
			}
		}
	}
	s_seg0[lane] = (unsigned long long)seg0;
	s_idx0[lane] = idx;
	s_n[lane] = r.pt_count;
	s_append[lane] = r.append;
	s_sc[lane] = sc;
	s_dx[lane] = dx;
	s_w.mask[lane] = mask;
	auto put = [&](uint32_t k, uint32_t j, double x, double y) {
		const double ksc = s_sc[k];
		x *= ksc; // point.rs:96-99
		y *= ksc;
		x += s_dx[k]; // point.rs:83-86
		y += 0.0;
		place((size_t)s_seg0[k], s_idx0[k] + j, s_n[k], s_append[k], x, y);
	};
	(void)wave_parallel_points(
	    s_w, lane, n_items, lev, lx, ly, (double)cmd.x1, (double)cmd.y1, (double)cmd.x2, (double)cmd.y2, (double)cmd.x, (double)cmd.y,
	    [&](uint32_t k, uint32_t j, double x, double y) { put(k, j, x, y); },
	    [&](uint32_t k, uint32_t cand, double x, double y) {
		    // the leaf's index inside its command: the leaves in front of it
		    put(k, (uint32_t)__builtin_popcountll(s_w.mask[k] & ((1ull << cand) - 1ull)), x, y);
	    },
	    [](uint32_t) {});
	// sequential walks, one lane at a time (they share the wave's work list in LDS)
	const bool walk = active && n_items == 0 && (cmd.kind == CMD_MOVE || cmd.kind == CMD_LINE || ((cmd.kind == CMD_QUAD || cmd.kind == CMD_CURVE) && open));
	for (unsigned long long todo = __ballot(walk); todo; todo &= todo - 1) {
		if (lane != (uint32_t)__builtin_ctzll(todo))
			continue;
		run_command(cmd, open, lx, ly, s_stack, [&](double x, double y) {
			x *= sc; // point.rs:96-99
			y *= sc;
			x += dx; // point.rs:83-86
			y += 0.0;
			place(seg0, idx, r.pt_count, r.append, x, y);
			idx++;
		});
	}
}


// ---------------------------------------------------------------------------------------
// glyf decode: thread per PART (one simple glyph of a — possibly composite — glyph, with the transform ttf-parser has
// accumulated for it).  The host only looks glyphs up and copies each simple glyph's `glyf` arrays (end points of the
// contours, then flags / x / y as they stand in the font, instructions left out); this pass replays ttf-parser's walk
// (glyf.rs: parse_simple_outline — flag runs, short / same-or-positive coordinates, wrapping i16 sums — and Builder:
// implied on-curve midpoints, the closing curve of a contour, close()) and writes the OutlineBuilder callbacks as the
// OutlineCmd records every later pass reads: what csrc/host/ttf_face.cpp records on the host, callback for callback
// (f32, one multiply-add pair per transformed coordinate, never fused).
// A part owns `cmd_cap` >= points + 2 * contours command slots (a contour of L points brings at most L + 2 callbacks: one per
// point, the closing line or curve, close(); the second closing curve of a contour that begins and ends off the curve comes
// instead of its first point's callback; end points that do not ascend cannot add contours — the lengths EndpointsIter
// hands out sum to at least the point count); the slots it does not need are filled with close():
// on the empty ring that follows a contour's own close() the RingBuilder does nothing (ring_builder.rs:33-38).
// error_flag bit 4: an entry whose arrays do not fit its bytes or its slots — ttf-parser would drop that glyph and, in a
// composite, the components after it; the host records such a batch itself.
// ---------------------------------------------------------------------------------------
struct GlyfPart {          // mirrors vgsdf_glyf_part (include/vgsdf.h)
	uint32_t byte_off;     // into `bytes`: endPtsOfContours[n_contours], then the flag / x / y arrays of the entry
	uint32_t byte_len;
	uint32_t cmd_at, cmd_cap;
	uint32_t n_contours;
	uint32_t plain;        // 1: identity transform
	float a, b, c, d, e, f;
};
static_assert(sizeof(GlyfPart) == 48, "GlyfPart layout");

// One wave per part, the points dealt out to the lanes (a lane that walks a 300-point glyph alone needs ~0.2 ms: ~700 dependent
// instructions per point on a machine that issues one every few cycles per wave):
//   A  flags: run-length decoding without a walk.  A byte of the flag stream is a repeat COUNT iff the byte in front of it
//      is a flag with REPEAT set — inside a run of bytes that all carry bit 3 flags and counts alternate, so a byte's role
//      follows from the length of the run of such bytes in front of it (one ballot per 64 bytes); runs are laid out by a
//      prefix sum, every flag lands on the points it covers (LDS), the sizes of the x / y arrays come out as sums.
//   B  coordinates: offset of a point's delta = prefix sum of the sizes its flags give, value = prefix sum of the deltas
//      (mod 2^16, as the i16 additions of the walk wrap).
//   C  contours and callbacks: the end points give every contour's first and last point (ttf-parser's EndpointsIter, also for
//      end points that do not ascend); what Builder::push_point emits for a point depends on its own flag, the flag of
//      the point in front of it and the first two points of its contour — no state is carried; the last point of a
//      contour also emits what Builder::finish adds; positions by prefix sum.
// A wave's time is a handful of dependent memory round trips (~1.5 us each), not arithmetic: the part's bytes are fetched
// ONCE (coalesced dwords into LDS; flags, coordinates and end points are then read there), and the LDS a workgroup takes is
// sized per launch from the batch's largest part, so that as a rule every part of a font is resident at once (3993 parts of
// Noto Sans Regular: 25.8 us with 21 KB of LDS per wave and the arrays read from global memory).
constexpr uint32_t kGlyfMaxPoints = 6144; // per part (LDS: 1 + 2 + 2 bytes per point + its bytes); beyond: the host's reader (error_flag bit 4)
constexpr uint32_t kGlyfMaxBytes = 30 * 1024;   // (together below the 64 KB a workgroup gets without asking for more)

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v)
{
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)v, d);
		if ((int)(threadIdx.x & 63u) >= d)
			v = max(v, o);
	}
	return v;
}

// max_points / max_bytes: what the launch's LDS was sized for (the batch's largest cmd_cap bounds its largest point count)
// cmd_open (may be NULL): the context pass's result for these commands, written here as well — whether the ring is non-empty in
// front of a command follows from the contour rules alone: every contour of a part begins on an empty ring (the contour
// in front of it, or the filler behind the previous part, ended with close()), its move_to opens the ring and everything
// behind it up to its close() finds it open.  (Bit 1 of the context byte — a glyph whose scale is not positive and finite —
// is the caller's to rule out: the host passes cmd_open only when every scale of the batch is.)
__global__ __launch_bounds__(64) void glyf_decode(const GlyfPart *__restrict__ parts, uint32_t n_parts, const uint8_t *__restrict__ bytes,
                                                  OutlineCmd *__restrict__ cmds, uint32_t *__restrict__ error_flag, uint32_t max_points,
                                                  uint32_t max_bytes, uint8_t *__restrict__ cmd_open)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
	uint8_t *const body = s_dyn;                                              // [max_bytes] the part's bytes (a multiple of 4)
	short *const s_x = reinterpret_cast<short *>(s_dyn + max_bytes);           // [max_points]
	short *const s_y = s_x + max_points;                                       // [max_points]
	uint32_t *const s_last = reinterpret_cast<uint32_t *>(s_y + max_points);   // bit p: point p is the last of its contour
	uint8_t *const s_flag = reinterpret_cast<uint8_t *>(s_last + (max_points + 31u) / 32u);
	if (blockIdx.x >= n_parts)
		return;
	const uint32_t lane = threadIdx.x;
	const GlyfPart pt = parts[blockIdx.x];
	const uint32_t len = pt.byte_len, nc = pt.n_contours, cap = pt.cmd_cap;
	const bool fits = len <= max_bytes;
	if (fits) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(bytes + pt.byte_off); // (4-aligned, padded: checked on the host)
		uint32_t *dst = reinterpret_cast<uint32_t *>(body);
		for (uint32_t w = lane; w < (len + 3u) / 4u; w += 64u)
			dst[w] = src[w];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
	}
	OutlineCmd *out = cmds + pt.cmd_at;
	uint8_t *out_open = cmd_open ? cmd_open + pt.cmd_at : nullptr;
	auto u16 = [&](uint32_t at) { return (uint32_t)((body[at] << 8) | body[at + 1]); };
	auto close_cmd = [] {
		OutlineCmd o;
		o.x1 = o.y1 = o.x2 = o.y2 = o.x = o.y = 0.0f;
		o.kind = CMD_CLOSE;
		return o;
	};
	// everything below is wave-uniform control flow
	bool ok = fits && nc != 0 && 2u * nc <= len;
	uint32_t n_points = 0;
	if (ok) {
		const uint32_t last_end = u16(2u * (nc - 1u));
		ok = last_end != 0xFFFFu;
		n_points = last_end + 1u;
	}
	uint32_t written = 0;
	bool open_end = false; // the part's last contour ran out of points (end points beyond the entry's points): no close() of its own
	if (ok && n_points > 1u)
		ok = n_points <= max_points && (unsigned long long)n_points + 2ull * nc <= cap;
	if (ok && n_points > 1u) {
		// ---- A: flags ----
		uint32_t covered = 0, xs = 0, ys = 0, x_at = 2u * nc;
		bool carry_count = false; // the first byte of the next 64 is a repeat count
		for (uint32_t base = 2u * nc; covered < n_points && ok; base += 64u) {
			const uint32_t j = base + lane;
			const bool inb = j < len;
			const uint32_t b = inb ? body[j] : 0u;
			unsigned long long ones = __ballot(inb && (b & 0x08u));
			if (carry_count)
				ones &= ~1ull;
			// role of byte j: the bytes [a, j) all carry bit 3 (a = the first such); flag, count, flag, ... from a on
			const unsigned long long below = (1ull << lane) - 1ull;
			const unsigned long long zeros_below = ~ones & below;
			const uint32_t a = zeros_below ? 64u - (uint32_t)__builtin_clzll(zeros_below) : (carry_count ? 1u : 0u);
			const bool is_count = lane == 0 ? carry_count : (((ones >> (lane - 1u)) & 1ull) != 0 && ((lane - 1u - a) & 1u) == 0);
			const bool rep = ((ones >> lane) & 1ull) != 0 && !is_count;
			// (the role of byte base + 64, computed the same way for a virtual lane 64)
			{
				const unsigned long long zb = ~ones;
				const uint32_t a64 = zb ? 64u - (uint32_t)__builtin_clzll(zb) : (carry_count ? 1u : 0u);
				carry_count = (ones >> 63) != 0 && ((63u - a64) & 1u) == 0;
			}
			const bool is_flag = inb && !is_count;
			uint32_t run = 0;
			bool bad = false;
			if (is_flag) {
				run = 1;
				if (rep) {
					if (j + 1u < len)
						run += body[j + 1u];
					else
						bad = true; // the count lies behind the entry
				}
			}
			uint32_t total_run;
			const uint32_t start = covered + wave_exclusive_sum(run, total_run);
			const bool needed = is_flag && start < n_points;
			// the stream must not end before the points are covered, a run must not cross their end
			const bool ends_here = !inb && start < n_points; // (a lane behind the entry while points are still open)
			bad = (needed && (bad || start + run > n_points)) || ends_here;
			if (__ballot(bad))
				ok = false;
			if (needed && !bad) {
				for (uint32_t r = 0; r < run; r++)
					s_flag[start + r] = (uint8_t)b;
			}
			const uint32_t cx = needed ? ((b & 0x02u) ? run : ((b & 0x10u) ? 0u : 2u * run)) : 0u;
			const uint32_t cy = needed ? ((b & 0x04u) ? run : ((b & 0x20u) ? 0u : 2u * run)) : 0u;
			uint32_t tx, ty;
			(void)wave_exclusive_sum(cx, tx);
			(void)wave_exclusive_sum(cy, ty);
			xs += tx;
			ys += ty;
			const uint32_t end_here = needed ? j + 1u + (rep ? 1u : 0u) : 0u;
			x_at = max(x_at, (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(end_here), 63));
			covered += total_run; // (runs of bytes that are not needed any more do not matter: the loop ends)
		}
		const uint32_t y_at = x_at + xs, y_end = y_at + ys;
		if (ok)
			ok = y_end <= len;
		if (ok) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			// ---- B: coordinates ----
			uint32_t offx = x_at, offy = y_at;
			int accx = 0, accy = 0;
			for (uint32_t base = 0; base < n_points; base += 64u) {
				const uint32_t p = base + lane;
				const bool in = p < n_points;
				const uint32_t fl = in ? s_flag[p] : 0x30u; // (0x30: no bytes, no delta)
				const uint32_t szx = (fl & 0x02u) ? 1u : ((fl & 0x10u) ? 0u : 2u);
				const uint32_t szy = (fl & 0x04u) ? 1u : ((fl & 0x20u) ? 0u : 2u);
				uint32_t tx, ty;
				const uint32_t ax = offx + wave_exclusive_sum(in ? szx : 0u, tx);
				const uint32_t ay = offy + wave_exclusive_sum(in ? szy : 0u, ty);
				offx += tx;
				offy += ty;
				int dx = 0, dy = 0;
				if (in) {
					if (fl & 0x02u) {
						const int v = body[ax];
						dx = (fl & 0x10u) ? v : -v;
					} else if (!(fl & 0x10u)) {
						dx = (int)(short)u16(ax);
					}
					if (fl & 0x04u) {
						const int v = body[ay];
						dy = (fl & 0x20u) ? v : -v;
					} else if (!(fl & 0x20u)) {
						dy = (int)(short)u16(ay);
					}
				}
				uint32_t sdx, sdy;
				const uint32_t ex = wave_exclusive_sum((uint32_t)dx, sdx), ey = wave_exclusive_sum((uint32_t)dy, sdy);
				if (in) {
					s_x[p] = (short)(unsigned short)((uint32_t)accx + ex + (uint32_t)dx); // wrapping i16 sums
					s_y[p] = (short)(unsigned short)((uint32_t)accy + ey + (uint32_t)dy);
				}
				accx = (int)((uint32_t)accx + sdx);
				accy = (int)((uint32_t)accy + sdy);
			}
			// ---- C: contours ----
			for (uint32_t w = lane; w < (n_points + 31u) / 32u; w += 64u)
				s_last[w] = 0;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			uint32_t laid = 0; // points of the contours laid out so far
			for (uint32_t base = 0; base < nc; base += 64u) {
				const uint32_t k = base + lane;
				uint32_t L = 0;
				if (k < nc) {
					const uint32_t end = u16(2u * k);
					if (k == 0) {
						L = end + 1u;
					} else {
						const uint32_t prev = u16(2u * (k - 1u));
						L = end > prev ? end - prev : 1u; // (a span of 0 still takes one point: EndpointsIter)
					}
				}
				uint32_t tl;
				const uint32_t first = laid + wave_exclusive_sum(L, tl);
				if (k < nc) {
					const unsigned long long last = (unsigned long long)first + L - 1ull;
					if (last < n_points)
						atomicOr(&s_last[last >> 5], 1u << (last & 31u));
				}
				laid += tl; // (sums beyond 2^32 cannot occur: nc * 65536 < 2^32)
			}
			// points behind the last contour: every one of them ends a contour of its own
			for (uint32_t p = laid + lane; p < n_points; p += 64u)
				atomicOr(&s_last[p >> 5], 1u << (p & 31u));
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			__builtin_amdgcn_wave_barrier();
			auto is_last = [&](uint32_t p) { return ((s_last[p >> 5] >> (p & 31u)) & 1u) != 0; };
			auto map = [&](float &x, float &y) {
				if (!pt.plain) {
					const float tx = x, ty = y;
					x = pt.a * tx + pt.c * ty + pt.e;
					y = pt.b * tx + pt.d * ty + pt.f;
				}
			};
			uint32_t start_carry = 0; // first point of the contour the previous 64 points ended in
			for (uint32_t base = 0; base < n_points; base += 64u) {
				const uint32_t p = base + lane;
				const bool in = p < n_points;
				// first point of p's contour: behind the nearest earlier last point
				const uint32_t mark = (in && p > 0 && is_last(p - 1u)) ? p : 0u;
				const uint32_t cstart = max(start_carry, wave_inclusive_max(mark));
				start_carry = (uint32_t)__builtin_amdgcn_readlane((int)cstart, 63);
				// how many callbacks point p brings (its own, and Builder::finish behind the last point of a contour), then where
				uint32_t i = 0, n = 0;
				bool on = false, on0 = false, on1 = false, pending = false, last_pt = false, has_start = false, has_lead = false, pend = false;
				if (in) {
					i = p - cstart;
					on = (s_flag[p] & 1u) != 0;
					on0 = (s_flag[cstart] & 1u) != 0;
					on1 = i >= 1u && (s_flag[cstart + 1u] & 1u) != 0;
					pending = i >= 1u && (s_flag[p - 1u] & 1u) == 0; // (read for i >= 2, or i == 1 behind an on-curve start)
					const bool own = i == 0u ? on0 : ((i == 1u && !on0) ? true : (pending || on));
					last_pt = is_last(p);
					has_start = on0 || i >= 1u;
					has_lead = !on0;
					pend = i == 0u ? false : ((i == 1u && !on0) ? !on1 : !on);
					n = (own ? 1u : 0u) + (last_pt ? ((has_lead && pend) ? 1u : 0u) + (has_start ? 1u : 0u) + 1u : 0u);
				}
				uint32_t tn;
				uint32_t at = written + wave_exclusive_sum(n, tn);
				bool ring_open = false; // in front of the command being emitted (set below, per command)
				auto emit = [&](uint32_t kind, float x1, float y1, float x, float y) {
					if (at < cap) {
						OutlineCmd o;
						o.x1 = x1, o.y1 = y1, o.x2 = 0.0f, o.y2 = 0.0f, o.x = x, o.y = y;
						o.kind = kind;
						out[at] = o;
						if (out_open)
							out_open[at] = ring_open ? 1 : 0;
					}
					at++;
				};
				auto move = [&](float mx, float my) {
					map(mx, my);
					emit(CMD_MOVE, 0.0f, 0.0f, mx, my);
				};
				auto line = [&](float mx, float my) {
					map(mx, my);
					emit(CMD_LINE, 0.0f, 0.0f, mx, my);
				};
				auto quad = [&](float cx, float cy, float ex, float ey) {
					map(cx, cy);
					map(ex, ey);
					emit(CMD_QUAD, cx, cy, ex, ey);
				};
				if (in && n) {
					const float fx = (float)s_x[p], fy = (float)s_y[p];
					const float x0 = (float)s_x[cstart], y0 = (float)s_y[cstart];
					float x1 = 0, y1 = 0;
					if (i >= 1u)
						x1 = (float)s_x[cstart + 1u], y1 = (float)s_y[cstart + 1u];
					// the contour's start point: its first point when that lies on the curve, else the second, else their middle
					const float stx = on0 ? x0 : (on1 ? x1 : x0 + 0.5f * (x1 - x0));
					const float sty = on0 ? y0 : (on1 ? y1 : y0 + 0.5f * (y1 - y0));
					// Builder::push_point (the contour's move_to finds the ring empty, whatever follows finds it open)
					ring_open = !(i == 0u || (i == 1u && !on0));
					if (i == 0u) {
						if (on0)
							move(fx, fy);
					} else if (i == 1u && !on0) {
						move(stx, sty);
					} else if (pending) {
						const float qx = (float)s_x[p - 1u], qy = (float)s_y[p - 1u];
						if (on)
							quad(qx, qy, fx, fy);
						else
							quad(qx, qy, qx + 0.5f * (fx - qx), qy + 0.5f * (fy - qy));
					} else if (on) {
						line(fx, fy);
					}
					if (last_pt) { // Builder::finish
						ring_open = has_start; // (a contour of one off-curve point emitted no move_to: its close() meets an empty ring)
						if (has_lead && pend) {
							quad(fx, fy, fx + 0.5f * (x0 - fx), fy + 0.5f * (y0 - fy));
							pend = false;
						}
						if (has_start && has_lead)
							quad(x0, y0, stx, sty);
						else if (has_start && pend)
							quad(fx, fy, stx, sty);
						else if (has_start)
							line(stx, sty);
						emit(CMD_CLOSE, 0.0f, 0.0f, 0.0f, 0.0f);
					}
				}
				written += tn;
				// the entry's very last point: when it does not end its contour (points that ran out inside a contour leave it
				// unclosed, ttf-parser: no finish()), the ring is still open behind the part's last callback
				if (base + 64u >= n_points)
					open_end = __ballot(in && p == n_points - 1u && !last_pt && has_start) != 0;
			}
		}
	}
	if (ok && open_end && out_open && written >= cap) {
		ok = false; // (no filler slot to close the ring in: the per-part context rule cannot say what the next part meets)
		open_end = false;
	}
	if (!ok) {
		if (lane == 0)
			atomicOr(error_flag, 16u);
		written = 0;
		open_end = false;
	}
	// filler: close() on the empty ring does nothing (ring_builder.rs:33-38) — except the first one behind an unclosed
	// contour, which finds the ring open and ends it, as the context pass would note
	for (uint32_t k = written + lane; k < cap; k += 64u) {
		out[k] = close_cmd();
		if (out_open)
			out_open[k] = (open_end && k == written) ? 1 : 0;
	}
}


// The upload of a submission whose input sits in ONE page-locked, device-mapped block: 16 bytes per thread, every load of
// the block in flight at once — one PCIe round trip plus the bytes, without the copy engine's hand-over in front of the
// first kernel (hipMemcpyAsync: 16.5 us for the 200 KB of a font + 8.7 us until the next kernel starts).
__global__ __launch_bounds__(256) void copy_in(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16, const uint8_t *__restrict__ src_tail,
                                               uint8_t *__restrict__ dst_tail, uint32_t n_tail)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n16)
		dst[i] = src[i];
	if (i < n_tail)
		dst_tail[i] = src_tail[i];
}

} // namespace vgsdf

using namespace vgsdf;

extern "C" int vgsdf_copy_in(const void *src_mapped, void *dst, size_t bytes, hipStream_t stream)
{
	if (bytes == 0)
		return 0;
	const uint32_t n16 = (uint32_t)(bytes / 16), n_tail = (uint32_t)(bytes % 16);
	hipLaunchKernelGGL(copy_in, dim3((std::max(n16, 1u) + 255u) / 256u), dim3(256), 0, stream, (const uint4 *)src_mapped, (uint4 *)dst, n16,
	                   (const uint8_t *)src_mapped + 16 * (size_t)n16, (uint8_t *)dst + 16 * (size_t)n16, n_tail);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_glyf_decode(const void *parts, uint32_t n_parts, const uint8_t *bytes, OutlineCmd *cmds, uint32_t *error_flag,
                                 uint32_t max_cmd_cap, uint32_t max_byte_len, uint8_t *cmd_open, hipStream_t stream)
{
	if (n_parts == 0)
		return 0;
	// LDS of a workgroup: the largest part's bytes + 5 bytes and a bit per point (a part has fewer points than command slots);
	// parts beyond the limits fail on the device (error_flag bit 4: the host's reader takes the batch)
	const uint32_t max_points = std::min(std::max(max_cmd_cap, 64u), kGlyfMaxPoints);
	const uint32_t max_bytes = std::min((std::max(max_byte_len, 64u) + 15u) & ~15u, kGlyfMaxBytes);
	const size_t lds = (size_t)max_bytes + 4 * (size_t)max_points + 4 * (size_t)((max_points + 31u) / 32u) + max_points;
	hipLaunchKernelGGL(glyf_decode, dim3(n_parts), dim3(64), lds, stream, (const GlyfPart *)parts, n_parts, bytes, cmds, error_flag, max_points,
	                   max_bytes, cmd_open);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_context(const OutlineCmd *cmds, const uint32_t *cmd_off, const double *scale, uint32_t n_glyphs,
                                     uint8_t *cmd_open, uint32_t *error_flag, hipStream_t stream)
{
	if (n_glyphs == 0)
		return 0;
	hipLaunchKernelGGL(outline_context<false>, dim3(n_glyphs), dim3(64), 0, stream, cmds, cmd_off, scale, n_glyphs, cmd_open, error_flag,
	                   (const uint8_t *)nullptr, (const float *)nullptr, (const uint32_t *)nullptr, (OutlineCmd *)nullptr);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_context_packed(const uint8_t *kinds, const float *coords, const uint32_t *dat_off, const uint32_t *cmd_off,
                                            const double *scale, uint32_t n_glyphs, OutlineCmd *cmds_out, uint8_t *cmd_open,
                                            uint32_t *error_flag, hipStream_t stream)
{
	if (n_glyphs == 0)
		return 0;
	hipLaunchKernelGGL(outline_context<true>, dim3(n_glyphs), dim3(64), 0, stream, (const OutlineCmd *)nullptr, cmd_off, scale, n_glyphs,
	                   cmd_open, error_flag, kinds, coords, dat_off, cmds_out);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_count(const OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, const uint32_t *cmd_off,
                                   uint32_t n_glyphs, const double *scale, const double *shift_x, uint32_t *counts, void *cmd_box,
                                   unsigned long long *cmd_mask, uint32_t *error_flag, hipStream_t stream)
{
	if (n_cmds == 0)
		return 0;
	hipLaunchKernelGGL(outline_count, dim3((n_cmds + kFlattenThreads - 1) / kFlattenThreads), dim3(kFlattenThreads), 0, stream, cmds,
	                   cmd_open, n_cmds, cmd_off, n_glyphs, scale, shift_x, counts, (double4 *)cmd_box, cmd_mask, error_flag);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_rings(const OutlineCmd *cmds, const uint32_t *cmd_off, const uint8_t *cmd_open, const double *scale,
                                   const double *shift_x, uint32_t n_glyphs, const uint32_t *counts, uint32_t *pt_local,
                                   const void *cmd_box, RingRec *rings, uint32_t *cmd_ring, OutlineRect *rects,
                                   uint32_t *error_flag, hipStream_t stream)
{
	if (n_glyphs == 0)
		return 0;
	hipLaunchKernelGGL(outline_rings, dim3(n_glyphs), dim3(64), 0, stream, cmds, cmd_off, cmd_open, scale, shift_x, n_glyphs, counts, pt_local,
	                   (const double4 *)cmd_box, rings, cmd_ring, rects, error_flag);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_plan(const OutlineRect *rects, uint32_t n_glyphs, int span_list, uint32_t delta_cap, uint32_t span_max,
                                  uint32_t span_budget, uint32_t tile_cap, GlyphDesc *descs, uint2 *tiles, PlanHeader *hdr,
                                  const uint32_t *error_flag, unsigned long long seg_cap, unsigned long long out_cap,
                                  uint32_t launch_spans, const uint32_t *pbf_pre, const uint8_t *pbf_fix,
                                  unsigned long long *pbf_at, uint32_t *next_flag, hipStream_t stream)
{
	if (pbf_fix != nullptr && n_glyphs <= 4u * kPlanThreads)
		hipLaunchKernelGGL((outline_plan<4, true>), dim3(1), dim3(kPlanThreads), 0, stream, rects, n_glyphs, span_list, delta_cap, span_max,
		                   span_budget, tile_cap, descs, tiles, hdr, error_flag, seg_cap, out_cap, launch_spans, pbf_pre, pbf_fix, pbf_at, next_flag);
	else if (pbf_fix != nullptr && n_glyphs <= 8u * kPlanThreads)
		// groups of 4097 .. 8192 glyphs (the dispatcher's groups of several fonts: ~7000): eight placements per thread spill
		// to scratch, and still beat the unkept path below, which reads and classifies every rect twice (21 fonts in two
		// groups: 48 / 90 us per plan)
		hipLaunchKernelGGL((outline_plan<8, true>), dim3(1), dim3(kPlanThreads), 0, stream, rects, n_glyphs, span_list, delta_cap, span_max,
		                   span_budget, tile_cap, descs, tiles, hdr, error_flag, seg_cap, out_cap, launch_spans, pbf_pre, pbf_fix, pbf_at, next_flag);
	else
	hipLaunchKernelGGL((outline_plan<8, false>), dim3(1), dim3(kPlanThreads), 0, stream, rects, n_glyphs, span_list, delta_cap, span_max,
	                   span_budget, tile_cap, descs, tiles, hdr, error_flag, seg_cap, out_cap, launch_spans, pbf_pre, pbf_fix, pbf_at, next_flag);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_emit_segments(const OutlineCmd *cmds, uint32_t n_cmds, const uint8_t *cmd_open, const double *scale,
                                           const double *shift_x,
                                           const uint32_t *pt_local, const RingRec *rings, const uint32_t *cmd_ring,
                                           const GlyphDesc *descs, const PlanHeader *hdr, unsigned long long seg_cap, double *seg,
                                           const unsigned long long *cmd_mask, uint32_t n_box_glyphs, const uint32_t *cmd_off,
                                           const void *cmd_box, void *boxes, hipStream_t stream)
{
	if (n_cmds == 0)
		return 0;
	hipLaunchKernelGGL(outline_emit_segments, dim3(n_box_glyphs + (n_cmds + kFlattenThreads - 1) / kFlattenThreads), dim3(kFlattenThreads), 0,
	                   stream, cmds, n_cmds, cmd_open, scale, shift_x, pt_local, rings, cmd_ring, descs, hdr, seg_cap, (double2 *)seg, cmd_mask,
	                   n_box_glyphs, cmd_off, (const double4 *)cmd_box, (float4 *)boxes);
	return (int)hipGetLastError();
}
