// ring_builder.hpp — OutlineBuilder sink that turns ttf outline callbacks into closed,
// flattened rings.  Restates /root/reference/src/render/ring_builder.rs:8-117 and
// Ring::close (src/geometry/ring.rs:53-63).
#pragma once
#include <cmath>
#include <limits>
#include <vector>

#include "geometry.hpp"
#include "ttf_face.hpp"

namespace vg {

class RingBuilder final : public OutlineBuilder {
public:
	// ring_builder.rs:57-65 — `precision` is handed to the flatteners as tolerance_sq, in
	// unscaled font units (ring_builder.rs:91,108)
	static constexpr double kPrecision = 0.01;

	void reset()
	{
		rings_.clear();
		ring_.clear();
	}

	// ring_builder.rs:69-72
	void move_to(float x, float y) override
	{
		save_ring();
		ring_.push_back(Point{(double)x, (double)y}); // point.rs:108-112 exact widening
	}
	// :75-77
	void line_to(float x, float y) override { ring_.push_back(Point{(double)x, (double)y}); }
	// :82-93
	void quad_to(float x1, float y1, float x, float y) override
	{
		if (ring_.empty())
			return;
		const Point start = ring_.back();
		flatten_quadratic(ring_, start, Point{(double)x1, (double)y1}, Point{(double)x, (double)y}, kPrecision);
	}
	// :98-110
	void curve_to(float x1, float y1, float x2, float y2, float x, float y) override
	{
		if (ring_.empty())
			return;
		const Point start = ring_.back();
		flatten_cubic(ring_, start, Point{(double)x1, (double)y1}, Point{(double)x2, (double)y2},
		              Point{(double)x, (double)y}, kPrecision);
	}
	// :114-116
	void close() override { save_ring(); }

	// :26-29 — saves the trailing ring; the builder is then spent until reset()
	Rings &into_rings()
	{
		save_ring();
		return rings_;
	}

private:
	// :33-54 with Ring::close (ring.rs:53-63)
	void save_ring()
	{
		if (ring_.size() >= 3) {
			const Point first = ring_.front(), last = ring_.back();
			constexpr double eps = std::numeric_limits<double>::epsilon();
			if (std::fabs(first.x - last.x) > eps || std::fabs(first.y - last.y) > eps)
				ring_.push_back(first);
			if (ring_.size() >= 4)
				rings_.add_ring(ring_);
		}
		ring_.clear();
	}

	Rings rings_;
	std::vector<Point> ring_;
};

} // namespace vg
