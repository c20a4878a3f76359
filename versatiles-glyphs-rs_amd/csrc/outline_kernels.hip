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
// Pipeline (one launch each, all tiny next to the SDF raster):
//   count   thread per command: number of points the command appends to its ring
//   scan    exclusive prefix sum -> point offsets
//   emit    thread per command: write the points (font units)
//   rings   thread per glyph: ring acceptance/closing rules, segment counts, bbox -> rect
//   scan    exclusive prefix sum of segments per glyph
//   segs    thread per point: write the scaled + shifted SoA segments
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "outline_kernels.h"

namespace vgsdf {

constexpr double kTolSq = 0.01; // ring_builder.rs:62 `precision`, passed as tolerance_sq (:91,:108)
constexpr int kMaxStack = 18;   // the work list holds depth + 1 entries; flatness shrinks 4x per level, so depth 16
                                // covers control polygons up to ~4e8 font units.  Deeper (non-finite / absurd
                                // input, where the reference would never finish) is cut off: a curve emits at
                                // most 65 536 points.  (A stack-free variant that re-derives each node from
                                // the root measured slower than the scratch stack: 160 vs 100 us per pass.)

// Iterative de Casteljau, explicit LIFO stack, right half pushed first (ring.rs:119-144).
// Calls emit(x, y) for every point appended to the ring, in order.  Returns the count.
//
// The reference pushes (m, m2, e) then (s, m1, m) and pops the left half at once; what stays on its
// stack are the right halves of the ancestors.  A right half's start point is the end point of
// its left sibling's subtree, i.e. the point emitted last (the same f64 value m, carried down
// unchanged), so a pending entry needs only (control, end): 4 doubles.  The first kLdsLevels
// levels of every lane live in LDS ([component][level][lane]: conflict-free); a private array in
// scratch memory took ~1 us per push/pop and made the two flattening passes the longest kernels
// of the front-end.  Deeper levels (control polygons > ~10^4 font units) use a scratch array.
constexpr int kLdsLevels = 8;
struct QuadStack {
	double *lds;                            // [4][kLdsLevels][64], this lane's column pre-offset
	double deep[kMaxStack - kLdsLevels][4]; // levels kLdsLevels.. (rare)
	__device__ __forceinline__ void push(int n, double cx, double cy, double ex, double ey)
	{
		if (n < kLdsLevels) {
			lds[(0 * kLdsLevels + n) * 64] = cx;
			lds[(1 * kLdsLevels + n) * 64] = cy;
			lds[(2 * kLdsLevels + n) * 64] = ex;
			lds[(3 * kLdsLevels + n) * 64] = ey;
		} else {
			deep[n - kLdsLevels][0] = cx, deep[n - kLdsLevels][1] = cy, deep[n - kLdsLevels][2] = ex, deep[n - kLdsLevels][3] = ey;
		}
	}
	__device__ __forceinline__ void pop(int n, double &cx, double &cy, double &ex, double &ey)
	{
		if (n < kLdsLevels) {
			cx = lds[(0 * kLdsLevels + n) * 64];
			cy = lds[(1 * kLdsLevels + n) * 64];
			ex = lds[(2 * kLdsLevels + n) * 64];
			ey = lds[(3 * kLdsLevels + n) * 64];
		} else {
			cx = deep[n - kLdsLevels][0], cy = deep[n - kLdsLevels][1], ex = deep[n - kLdsLevels][2], ey = deep[n - kLdsLevels][3];
		}
	}
};

template <class Emit> __device__ __forceinline__ uint32_t flatten_quad(double sx, double sy, double cx, double cy,
                                                                         double ex, double ey, double *lds_col, Emit emit)
{
	QuadStack st;
	st.lds = lds_col;
	int n = 0; // pending right halves = the reference's stack size after its pop
	uint32_t count = 0;
	double qsx = sx, qsy = sy, qcx = cx, qcy = cy, qex = ex, qey = ey;
	for (;;) {
		const double dx = qsx + qex - qcx * 2.0; // ring.rs:129
		const double dy = qsy + qey - qcy * 2.0;
		// (non-finite control points never become flat: the reference would not terminate;
		// here the work list is bounded and the end point is emitted)
		if (dx * dx + dy * dy <= kTolSq || n + 2 > kMaxStack) {
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

// ring.rs:159-187
template <class Emit>
__device__ __forceinline__ uint32_t flatten_cubic(double sx, double sy, double ax, double ay, double bx, double by,
                                                  double ex, double ey, Emit emit)
{
	double st[kMaxStack][8];
	int n = 0;
	uint32_t count = 0;
	st[n][0] = sx, st[n][1] = sy, st[n][2] = ax, st[n][3] = ay, st[n][4] = bx, st[n][5] = by, st[n][6] = ex, st[n][7] = ey;
	n++;
	while (n > 0) {
		n--;
		const double s0 = st[n][0], s1 = st[n][1], a0 = st[n][2], a1 = st[n][3], b0 = st[n][4], b1 = st[n][5], e0 = st[n][6],
		             e1 = st[n][7];
		const double dx = (b0 + a0) - (s0 + e0);
		const double dy = (b1 + a1) - (s1 + e1);
		if (dx * dx + dy * dy <= kTolSq || n + 2 > kMaxStack) {
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
		st[n][0] = mx, st[n][1] = my, st[n][2] = p123x, st[n][3] = p123y, st[n][4] = p23x, st[n][5] = p23y, st[n][6] = e0,
		st[n][7] = e1;
		n++;
		st[n][0] = s0, st[n][1] = s1, st[n][2] = p01x, st[n][3] = p01y, st[n][4] = p012x, st[n][5] = p012y, st[n][6] = mx,
		st[n][7] = my;
		n++;
	}
	return count;
}

// One command.  `ring_open` is false right after a CLOSE / at the glyph start, where the
// ring is empty and quad_to / curve_to are ignored (ring_builder.rs:83-85,99-101).
template <class Emit>
__device__ __forceinline__ uint32_t run_command(const OutlineCmd &c, bool ring_open, double lastx, double lasty, double *lds_col,
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
		return flatten_quad(lastx, lasty, (double)c.x1, (double)c.y1, (double)c.x, (double)c.y, lds_col, emit);
	case CMD_CURVE: // :98-110
		if (!ring_open)
			return 0;
		return flatten_cubic(lastx, lasty, (double)c.x1, (double)c.y1, (double)c.x2, (double)c.y2, (double)c.x, (double)c.y, emit);
	default: // CMD_CLOSE
		return 0;
	}
}

// One wave per glyph: is the ring non-empty when command c arrives?  (quad_to / curve_to are
// ignored on an empty ring, ring_builder.rs:83-85,99-101.)  The ring is empty at the glyph
// start and after close(); move_to and line_to make it non-empty.  64 commands per step: the
// state in front of a command is decided by the nearest earlier state-changing command,
// found with two ballots.
__global__ __launch_bounds__(64) void outline_context(const OutlineCmd *__restrict__ cmds,
                                                      const uint32_t *__restrict__ cmd_off, uint32_t n_glyphs,
                                                      uint8_t *__restrict__ cmd_open)
{
	const uint32_t g = blockIdx.x, lane = threadIdx.x;
	if (g >= n_glyphs)
		return;
	const uint32_t c0 = cmd_off[g], c1 = cmd_off[g + 1];
	bool carry = false; // ring state in front of the current 64-command window
	for (uint32_t base = c0; base < c1; base += 64) {
		const uint32_t c = base + lane;
		const uint32_t k = c < c1 ? cmds[c].kind : 0xFFu;
		const unsigned long long opens = __ballot(k == CMD_MOVE || k == CMD_LINE);
		const unsigned long long closes = __ballot(k == CMD_CLOSE);
		const unsigned long long before = (opens | closes) & ((1ull << lane) - 1ull);
		bool open = carry;
		if (before) {
			const int top = 63 - __builtin_clzll(before);
			open = (opens >> top) & 1ull;
		}
		if (c < c1)
			cmd_open[c] = open ? 1 : 0;
		const unsigned long long all = opens | closes;
		if (all) {
			const int top = 63 - __builtin_clzll(all);
			carry = (opens >> top) & 1ull;
		}
	}
}

// When the ring is open the previous command of the glyph emitted at least one point and
// ended on its own (x, y): that is the current point of the ring.
__device__ __forceinline__ void command_context(const OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t i,
                                                bool &ring_open, double &lastx, double &lasty)
{
	ring_open = cmd_open[i] != 0;
	lastx = ring_open ? (double)cmds[i - 1].x : 0.0;
	lasty = ring_open ? (double)cmds[i - 1].y : 0.0;
}

constexpr int kFlattenThreads = 64; // one wave per workgroup: 16 KiB of LDS stack each, 9 workgroups per CU

__global__ __launch_bounds__(kFlattenThreads) void outline_count(const OutlineCmd *__restrict__ cmds,
                                                                 const uint8_t *__restrict__ cmd_open, uint32_t n_cmds,
                                                                 uint32_t *__restrict__ counts)
{
	__shared__ double s_stack[4 * kLdsLevels * 64];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n_cmds)
		return;
	if (i == n_cmds) { // sentinel so the exclusive scan also yields the total
		counts[i] = 0;
		return;
	}
	bool open;
	double lx, ly;
	command_context(cmds, cmd_open, i, open, lx, ly);
	counts[i] = run_command(cmds[i], open, lx, ly, s_stack + threadIdx.x, [](double, double) {});
}

__global__ __launch_bounds__(kFlattenThreads) void outline_emit(const OutlineCmd *__restrict__ cmds,
                                                                const uint8_t *__restrict__ cmd_open, uint32_t n_cmds,
                                                                const uint32_t *__restrict__ pt_off, double *__restrict__ ptx,
                                                                double *__restrict__ pty, double4 *__restrict__ cmd_box)
{
	__shared__ double s_stack[4 * kLdsLevels * 64];
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_cmds)
		return;
	bool open;
	double lx, ly;
	command_context(cmds, cmd_open, i, open, lx, ly);
	uint32_t k = pt_off[i];
	// bounding box of the RAW points this command appends (font units): scale > 0 and the shift are
	// monotone, so the ring pass can take min / max per command and transform four numbers instead of
	// reading every point again (fmin / fmax skip NaN exactly as they do there)
	const double inf = __builtin_huge_val();
	double minx = inf, miny = inf, maxx = -inf, maxy = -inf;
	run_command(cmds[i], open, lx, ly, s_stack + threadIdx.x, [&](double x, double y) {
		ptx[k] = x;
		pty[k] = y;
		k++;
		minx = fmin(minx, x);
		miny = fmin(miny, y);
		maxx = fmax(maxx, x);
		maxy = fmax(maxy, y);
	});
	cmd_box[i] = make_double4(minx, miny, maxx, maxy);
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

// One wave per glyph: RingBuilder::save_ring + Ring::close (ring_builder.rs:33-54,
// ring.rs:53-63), rings.scale / translate / get_bbox and Renderer::prepare_glyph
// (renderer.rs:122-137, 64-91).  Lane 0 walks the (short) command list and applies the ring
// rules; all lanes then reduce the bounding box over the accepted rings' points.
constexpr int kRingCmdsLds = 1024; // commands cached in LDS per glyph (longer lists are read from global)
constexpr int kRingListLds = 256;  // accepted rings listed in LDS per glyph (more: lane 0 reduces them alone)

__global__ __launch_bounds__(64) void outline_rings(const OutlineCmd *__restrict__ cmds, const uint32_t *__restrict__ cmd_off,
                                                    const uint32_t *__restrict__ pt_off, const double *__restrict__ ptx,
                                                    const double *__restrict__ pty, const double *__restrict__ scale,
                                                    const double *__restrict__ shift_x, uint32_t n_glyphs,
                                                    RingRec *__restrict__ rings, uint32_t *__restrict__ cmd_ring,
                                                    OutlineRect *__restrict__ rects, uint32_t *__restrict__ seg_count,
                                                    const double4 *__restrict__ cmd_box)
{
	__shared__ uint8_t s_kind[kRingCmdsLds];
	__shared__ uint32_t s_poff[kRingCmdsLds + 1];
	__shared__ uint32_t s_ring_a[kRingListLds], s_ring_n[kRingListLds];
	__shared__ uint32_t s_nlist, s_nseg, s_nrings;
	__shared__ double s_box[4];

	const uint32_t g = blockIdx.x, lane = threadIdx.x;
	if (g == n_glyphs) { // sentinel so the exclusive scan also yields the total
		if (lane == 0)
			seg_count[g] = 0;
		return;
	}
	const uint32_t c0 = cmd_off[g], c1 = cmd_off[g + 1];
	const uint32_t nc = c1 - c0;
	const bool cached = nc <= (uint32_t)kRingCmdsLds;
	if (cached) {
		for (uint32_t i = lane; i < nc; i += 64)
			s_kind[i] = (uint8_t)cmds[c0 + i].kind;
		for (uint32_t i = lane; i <= nc; i += 64)
			s_poff[i] = pt_off[c0 + i];
	}
	__syncthreads();
	const double sc = scale[g], dx = shift_x[g];
	const double inf = __builtin_huge_val();
	double minx = inf, miny = inf, maxx = -inf, maxy = -inf;
	auto include = [&](uint32_t i) { // scale, translate, bbox of raw point i
		double x = ptx[i], y = pty[i];
		x *= sc; // point.rs:96-99
		y *= sc;
		x += dx; // point.rs:83-86
		y += 0.0;
		minx = fmin(minx, x); // bbox.rs:64-69
		miny = fmin(miny, y);
		maxx = fmax(maxx, x);
		maxy = fmax(maxy, y);
	};

	// Fast path (command list cached in LDS, at most kRingListLds rings): lane 0 only walks the command
	// kinds (LDS, no global latency) to cut the list into rings; the per-ring work (first / last point
	// loads, acceptance, ring record) is then done one ring per lane, segment offsets by a wave scan,
	// and the per-command ring index is written coalesced.  Everything else: the serial walk below.
	__shared__ uint32_t s_ra[kRingListLds], s_rb[kRingListLds];
	__shared__ uint32_t s_cring[kRingCmdsLds];
	__shared__ uint32_t s_found;
	__shared__ uint8_t s_accept[kRingCmdsLds]; // per command slot: 1 = this command opened an accepted ring
	bool fast = cached;
	if (cached) {
		if (lane == 0) {
			uint32_t found = 0;
			uint32_t ring_start = c0;
			bool have = false; // a ring is being collected
			auto cut = [&](uint32_t ra, uint32_t rb) {
				if (found < (uint32_t)kRingListLds) {
					s_ra[found] = ra;
					s_rb[found] = rb;
				}
				found++;
			};
			for (uint32_t c = c0; c < c1; c++) {
				const uint32_t k = s_kind[c - c0];
				if (k == CMD_MOVE) { // move_to: save_ring, then start a new ring with this point
					if (have)
						cut(ring_start, c);
					ring_start = c;
					have = true;
				} else if (k == CMD_CLOSE) { // close: save_ring
					if (have)
						cut(ring_start, c);
					have = false;
					ring_start = c + 1;
				} else if (!have && k == CMD_LINE) { // line_to on an empty ring starts one (ring_builder.rs:75-77)
					ring_start = c;
					have = true;
				}
				s_cring[c - c0] = have ? ring_start : 0xFFFFFFFFu;
			}
			if (have)
				cut(ring_start, c1); // into_rings (ring_builder.rs:26-29)
			s_found = found;
		}
		__syncthreads();
		fast = s_found <= (uint32_t)kRingListLds;
	}
	const bool boxed = fast && sc > 0.0 && sc < inf; // monotone transform: per-command boxes can be used
	if (fast) {
		const uint32_t found = s_found;
		for (uint32_t i = lane; i < nc; i += 64) {
			cmd_ring[c0 + i] = s_cring[i];
			s_accept[i] = 0;
		}
		__syncthreads();
		uint32_t seg_base = 0, n_rings_acc = 0, n_list_acc = 0; // wave-uniform running totals
		for (uint32_t base = 0; base < found; base += 64) {
			const uint32_t r = base + lane;
			const bool valid = r < found;
			RingRec rec;
			rec.pt_first = rec.pt_count = rec.append = rec.accepted = rec.seg_local = 0;
			rec.glyph = g;
			uint32_t segs = 0, ra = 0;
			if (valid) {
				ra = s_ra[r];
				const uint32_t a = s_poff[ra - c0], bb = s_poff[s_rb[r] - c0]; // raw points [a, bb)
				const uint32_t n = bb - a;
				rec.pt_first = a;
				rec.pt_count = n;
				if (n >= 3) { // ring_builder.rs:35-38
					const double fx = ptx[a], fy = pty[a], lx = ptx[bb - 1], ly = pty[bb - 1];
					const double eps = 2.220446049250313e-16;
					rec.append = (fabs(fx - lx) > eps || fabs(fy - ly) > eps) ? 1u : 0u; // ring.rs:60-62
					if (n + rec.append >= 4) { // ring_builder.rs:45-48
						rec.accepted = 1;
						segs = n + rec.append - 1;
					}
				}
			}
			// segment offsets / list slots of the accepted rings, in ring order: inclusive wave scans
			uint32_t incl = segs, lincl = rec.accepted;
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t o1 = __shfl_up(incl, d), o2 = __shfl_up(lincl, d);
				if ((int)lane >= d) {
					incl += o1;
					lincl += o2;
				}
			}
			if (valid) {
				rec.seg_local = seg_base + incl - segs;
				rings[ra] = rec; // ring records live at the index of the command that opened them
				if (rec.accepted) {
					const uint32_t slot = n_list_acc + lincl - 1; // < found <= kRingListLds
					s_ring_a[slot] = rec.pt_first;
					s_ring_n[slot] = rec.pt_count;
					s_accept[ra - c0] = 1;
				}
			}
			seg_base += __shfl(incl, 63);
			const uint32_t acc = __shfl(lincl, 63);
			n_rings_acc += acc;
			n_list_acc += acc;
		}
		if (lane == 0) {
			s_nlist = n_list_acc;
			s_nseg = seg_base;
			s_nrings = n_rings_acc;
		}
		__syncthreads();
	} else {
		if (lane == 0) {
			uint32_t nseg = 0, n_rings = 0, n_list = 0;
			auto kind_of = [&](uint32_t c) { return cached ? (uint32_t)s_kind[c - c0] : cmds[c].kind; };
			auto poff_of = [&](uint32_t c) { return cached ? s_poff[c - c0] : pt_off[c]; };
			// a ring = the points of the commands [ra, rb) that followed its MOVE (or a LINE on an empty ring)
			auto finish = [&](uint32_t ra, uint32_t rb) {
				const uint32_t a = poff_of(ra), b = poff_of(rb); // raw points [a, b)
				RingRec r;
				r.pt_first = a;
				r.pt_count = b - a;
				r.append = 0;
				r.accepted = 0;
				r.seg_local = nseg;
				r.glyph = g;
				const uint32_t n = b - a;
				if (n >= 3) { // ring_builder.rs:35-38
					const double fx = ptx[a], fy = pty[a], lx = ptx[b - 1], ly = pty[b - 1];
					const double eps = 2.220446049250313e-16;
					r.append = (fabs(fx - lx) > eps || fabs(fy - ly) > eps) ? 1u : 0u; // ring.rs:60-62
					if (n + r.append >= 4) { // ring_builder.rs:45-48
						r.accepted = 1;
						nseg += n + r.append - 1;
						n_rings++;
						if (n_list < (uint32_t)kRingListLds) {
							s_ring_a[n_list] = a;
							s_ring_n[n_list] = n;
							n_list++;
						} else {
							for (uint32_t i = a; i < b; i++) // overflow of the list: reduce here
								include(i);
						}
					}
				}
				rings[ra] = r; // ring records live at the index of the command that opened them
			};
			uint32_t ring_start = c0;
			bool have = false; // a ring is being collected
			for (uint32_t c = c0; c < c1; c++) {
				const uint32_t k = kind_of(c);
				if (k == CMD_MOVE) { // move_to: save_ring, then start a new ring with this point
					if (have)
						finish(ring_start, c);
					ring_start = c;
					have = true;
				} else if (k == CMD_CLOSE) { // close: save_ring
					if (have)
						finish(ring_start, c);
					have = false;
					ring_start = c + 1;
				} else if (!have && k == CMD_LINE) { // line_to on an empty ring starts one (ring_builder.rs:75-77)
					ring_start = c;
					have = true;
				}
				cmd_ring[c] = have ? ring_start : 0xFFFFFFFFu;
			}
			if (have)
				finish(ring_start, c1); // into_rings (ring_builder.rs:26-29)
			s_nlist = n_list;
			s_nseg = nseg;
			s_nrings = n_rings;
		}
		__syncthreads();

	}

	// all lanes: bbox over the accepted rings' points (the appended closing point repeats the first)
	if (boxed) {
		// per-command boxes of the commands that belong to an accepted ring, then the transform
		for (uint32_t i = lane; i < nc; i += 64) {
			const uint32_t rs = s_cring[i];
			if (rs != 0xFFFFFFFFu && s_accept[rs - c0]) {
				const double4 bx = cmd_box[c0 + i];
				minx = fmin(minx, bx.x);
				miny = fmin(miny, bx.y);
				maxx = fmax(maxx, bx.z);
				maxy = fmax(maxy, bx.w);
			}
		}
		minx *= sc, miny *= sc, maxx *= sc, maxy *= sc; // point.rs:96-99
		minx += dx, maxx += dx;                         // point.rs:83-86
		miny += 0.0, maxy += 0.0;
	} else {
		const uint32_t n_list = s_nlist;
		for (uint32_t r = 0; r < n_list; r++) {
			const uint32_t a = s_ring_a[r], n = s_ring_n[r];
			for (uint32_t i = lane; i < n; i += 64)
				include(a + i);
		}
	}
	for (int sh = 32; sh > 0; sh >>= 1) { // fmin/fmax are exact selections: any order gives the same box
		minx = fmin(minx, __shfl_xor(minx, sh));
		miny = fmin(miny, __shfl_xor(miny, sh));
		maxx = fmax(maxx, __shfl_xor(maxx, sh));
		maxy = fmax(maxy, __shfl_xor(maxy, sh));
	}
	if (lane == 0) {
		const uint32_t nseg = s_nseg, n_rings = s_nrings;
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
		seg_count[g] = rc.has_raster ? nseg : 0;
	}
	(void)s_box;
}

// Thread per raw point: Rings::get_segments (rings.rs:75-81) after scale + translate.
__global__ void outline_segments(const uint32_t *__restrict__ pt_off, uint32_t n_cmds, uint32_t n_points,
                                 const uint32_t *__restrict__ cmd_ring, const RingRec *__restrict__ rings,
                                 const OutlineRect *__restrict__ rects, const uint32_t *__restrict__ seg_off,
                                 const double *__restrict__ ptx, const double *__restrict__ pty,
                                 const double *__restrict__ scale, const double *__restrict__ shift_x,
                                 double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ ex,
                                 double *__restrict__ ey)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_points)
		return;
	// command that emitted point p: last c with pt_off[c] <= p (pt_off has n_cmds + 1 entries)
	uint32_t lo = 0, hi = n_cmds;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (pt_off[mid] <= p)
			lo = mid;
		else
			hi = mid;
	}
	const uint32_t rcmd = cmd_ring[lo];
	if (rcmd == 0xFFFFFFFFu)
		return;
	const RingRec r = rings[rcmd];
	if (!r.accepted || !rects[r.glyph].has_raster)
		return;
	const uint32_t idx = p - r.pt_first; // position inside the ring
	uint32_t q;                          // the segment's end point
	if (idx + 1 < r.pt_count)
		q = p + 1;
	else if (r.append)
		q = r.pt_first; // closing segment back to the first point
	else
		return; // last point of a ring that was already closed
	const double sc = scale[r.glyph], dx = shift_x[r.glyph];
	double ax = ptx[p], ay = pty[p], bx = ptx[q], by = pty[q];
	ax *= sc;
	ay *= sc;
	bx *= sc;
	by *= sc;
	ax += dx;
	ay += 0.0;
	bx += dx;
	by += 0.0;
	const uint32_t s = seg_off[r.glyph] + r.seg_local + idx;
	sx[s] = ax;
	sy[s] = ay;
	ex[s] = bx;
	ey[s] = by;
}

} // namespace vgsdf

using namespace vgsdf;

extern "C" size_t vgsdf_outline_scan_temp_bytes(uint32_t n)
{
	size_t bytes = 0;
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
	return bytes;
}

extern "C" int vgsdf_outline_scan(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, uint32_t n,
                                  hipStream_t stream)
{
	return (int)hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, stream);
}

extern "C" int vgsdf_outline_context(const OutlineCmd *cmds, const uint32_t *cmd_off, uint32_t n_glyphs, uint8_t *cmd_open,
                                     hipStream_t stream)
{
	if (n_glyphs == 0)
		return 0;
	hipLaunchKernelGGL(outline_context, dim3(n_glyphs), dim3(64), 0, stream, cmds, cmd_off, n_glyphs, cmd_open);
	return (int)hipGetLastError();
}

// counts has n_cmds + 1 entries (the last one is a 0 sentinel)
extern "C" int vgsdf_outline_count(const OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, uint32_t *counts,
                                   hipStream_t stream)
{
	hipLaunchKernelGGL(outline_count, dim3((n_cmds + 1 + kFlattenThreads - 1) / kFlattenThreads), dim3(kFlattenThreads), 0, stream, cmds, cmd_open, n_cmds, counts);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_emit(const OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, const uint32_t *pt_off,
                                  double *ptx, double *pty, void *cmd_box, hipStream_t stream)
{
	if (n_cmds == 0)
		return 0;
	hipLaunchKernelGGL(outline_emit, dim3((n_cmds + kFlattenThreads - 1) / kFlattenThreads), dim3(kFlattenThreads), 0, stream,
	                   cmds, cmd_open, n_cmds, pt_off, ptx, pty, (double4 *)cmd_box);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_rings(const OutlineCmd *cmds, const uint32_t *cmd_off, const uint32_t *pt_off, const double *ptx,
                                   const double *pty, const double *scale, const double *shift_x, uint32_t n_glyphs,
                                   RingRec *rings, uint32_t *cmd_ring, OutlineRect *rects, uint32_t *seg_count,
                                   const void *cmd_box, hipStream_t stream)
{
	// one wave per glyph; seg_count has n_glyphs + 1 entries (the last one is a 0 sentinel)
	hipLaunchKernelGGL(outline_rings, dim3(n_glyphs + 1), dim3(64), 0, stream, cmds, cmd_off, pt_off, ptx, pty, scale,
	                   shift_x, n_glyphs, rings, cmd_ring, rects, seg_count, (const double4 *)cmd_box);
	return (int)hipGetLastError();
}

extern "C" int vgsdf_outline_segments(const uint32_t *pt_off, uint32_t n_cmds, uint32_t n_points, const uint32_t *cmd_ring,
                                      const RingRec *rings, const OutlineRect *rects, const uint32_t *seg_off,
                                      const double *ptx, const double *pty, const double *scale, const double *shift_x,
                                      double *sx, double *sy, double *ex, double *ey, hipStream_t stream)
{
	if (n_points == 0)
		return 0;
	hipLaunchKernelGGL(outline_segments, dim3((n_points + 255) / 256), dim3(256), 0, stream, pt_off, n_cmds, n_points, cmd_ring,
	                   rings, rects, seg_off, ptx, pty, scale, shift_x, sx, sy, ex, ey);
	return (int)hipGetLastError();
}
