// geometry.hpp — host-side f64 outline geometry: the part of the reference's
// src/geometry that stays on the CPU (tessellation, scale/shift, bbox).  The distance
// routines of src/geometry/segment.rs live in the HIP kernel (sdf_kernels.hip).
//
//   Point            src/geometry/point.rs:6-112
//   BBox             src/geometry/bbox.rs:7-81
//   Ring / Rings     src/geometry/ring.rs:11-187, rings.rs:13-81
//
// Rings are stored flat (one point array + ring start offsets): that is the layout the
// GPU batch packer consumes, and it keeps a whole glyph in one allocation.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace vg {

struct Point {
	double x = 0, y = 0;
	// point.rs:29-31
	Point midpoint(const Point &o) const { return Point{(x + o.x) / 2.0, (y + o.y) / 2.0}; }
};

struct BBox {
	Point min{std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity()};
	Point max{-std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity()};
	// bbox.rs:64-69 (f64::min/max; inputs are never NaN)
	void include_point(const Point &p)
	{
		min.x = std::fmin(min.x, p.x);
		min.y = std::fmin(min.y, p.y);
		max.x = std::fmax(max.x, p.x);
		max.y = std::fmax(max.y, p.y);
	}
	// bbox.rs:56-58: empty only when it has no extent on BOTH axes
	bool is_empty() const { return max.x <= min.x && max.y <= min.y; }
};

// A set of closed rings, flat.  Ring r owns points [start[r], start[r+1]).
class Rings {
public:
	void clear()
	{
		pts_.clear();
		start_.assign(1, 0);
	}
	bool is_empty() const { return start_.size() <= 1; }
	size_t len() const { return start_.size() - 1; }
	const std::vector<Point> &points() const { return pts_; }
	std::vector<Point> &points() { return pts_; }
	const std::vector<uint32_t> &starts() const { return start_; }

	// Rings::add_ring
	void add_ring(const std::vector<Point> &ring)
	{
		pts_.insert(pts_.end(), ring.begin(), ring.end());
		start_.push_back((uint32_t)pts_.size());
	}
	// rings.rs:66-70 / point.rs:96-99
	void scale(double s)
	{
		for (Point &p : pts_) {
			p.x *= s;
			p.y *= s;
		}
	}
	// rings.rs:59-63 / point.rs:83-86
	void translate(const Point &o)
	{
		for (Point &p : pts_) {
			p.x += o.x;
			p.y += o.y;
		}
	}
	// rings.rs:50-56
	BBox get_bbox() const
	{
		BBox b;
		for (const Point &p : pts_)
			b.include_point(p);
		return b;
	}
	// number of segments Rings::get_segments() yields (rings.rs:75-81): points-1 per ring
	size_t segment_count() const { return pts_.size() - len(); }

private:
	std::vector<Point> pts_;
	std::vector<uint32_t> start_{0};
};

// ring.rs:119-144: de Casteljau subdivision with an explicit LIFO work list; the right
// half is pushed first so points come out in start->end order.  `tolerance_sq` is compared
// with |s + e - 2c|^2.
inline void flatten_quadratic(std::vector<Point> &ring, const Point &start, const Point &ctrl, const Point &end,
                              double tolerance_sq)
{
	struct Q { // no default initialisers: the work list must not be zero-filled per call
		double sx, sy, cx, cy, ex, ey;
	};
	constexpr int kInline = 48; // subdivision depth of real outlines stays below 16
	Q work[kInline];
	std::vector<Q> spill;
	int n = 0;
	auto push = [&](const Q &q) {
		if (n < kInline)
			work[n] = q;
		else
			spill.push_back(q);
		n++;
	};
	push(Q{start.x, start.y, ctrl.x, ctrl.y, end.x, end.y});
	while (n > 0) {
		Q q;
		if (n > kInline) {
			q = spill.back();
			spill.pop_back();
		} else {
			q = work[n - 1];
		}
		n--;
		const double dx = q.sx + q.ex - q.cx * 2.0;
		const double dy = q.sy + q.ey - q.cy * 2.0;
		if (dx * dx + dy * dy <= tolerance_sq) {
			ring.push_back(Point{q.ex, q.ey});
			continue;
		}
		// Point::midpoint (point.rs:29-31): (a + b) / 2.0
		const double m1x = (q.sx + q.cx) / 2.0, m1y = (q.sy + q.cy) / 2.0;
		const double m2x = (q.cx + q.ex) / 2.0, m2y = (q.cy + q.ey) / 2.0;
		const double mx = (m1x + m2x) / 2.0, my = (m1y + m2y) / 2.0;
		push(Q{mx, my, m2x, m2y, q.ex, q.ey});
		push(Q{q.sx, q.sy, m1x, m1y, mx, my});
	}
}

// ring.rs:159-187
inline void flatten_cubic(std::vector<Point> &ring, const Point &start, const Point &c1, const Point &c2,
                          const Point &end, double tolerance_sq)
{
	struct C {
		Point s, a, b, e;
	};
	std::vector<C> work;
	work.push_back(C{start, c1, c2, end});
	while (!work.empty()) {
		const C c = work.back();
		work.pop_back();
		const double dx = (c.b.x + c.a.x) - (c.s.x + c.e.x);
		const double dy = (c.b.y + c.a.y) - (c.s.y + c.e.y);
		if (dx * dx + dy * dy <= tolerance_sq) {
			ring.push_back(c.e);
			continue;
		}
		const Point p01 = c.s.midpoint(c.a), p12 = c.a.midpoint(c.b), p23 = c.b.midpoint(c.e);
		const Point p012 = p01.midpoint(p12), p123 = p12.midpoint(p23), m = p012.midpoint(p123);
		work.push_back(C{m, p123, p23, c.e});
		work.push_back(C{c.s, p01, p012, m});
	}
}

} // namespace vg
