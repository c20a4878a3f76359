#include "renderer.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace vg {

namespace {

// Rust `as i32` / `as u32` on f64: truncate, saturate, NaN -> 0
inline int32_t to_i32(double v)
{
	if (std::isnan(v))
		return 0;
	if (v >= 2147483647.0)
		return INT32_MAX;
	if (v <= -2147483648.0)
		return INT32_MIN;
	return (int32_t)v;
}
inline uint32_t to_u32(double v)
{
	if (std::isnan(v) || v <= 0.0)
		return 0;
	if (v >= 4294967295.0)
		return UINT32_MAX;
	return (uint32_t)v;
}

} // namespace

void GlyphBatch::clear()
{
	jobs.clear();
	raster_job.clear();
	seg_off.assign(1, 0);
	sx.clear();
	sy.clear();
	ex.clear();
	ey.clear();
	x0.clear();
	y0.clear();
	w.clear();
	h.clear();
	out_off.assign(1, 0);
}

vgsdf_batch GlyphBatch::view() const
{
	vgsdf_batch b;
	b.n_glyphs = (uint32_t)w.size();
	b.seg_off = seg_off.data();
	b.seg_sx = sx.data();
	b.seg_sy = sy.data();
	b.seg_ex = ex.data();
	b.seg_ey = ey.data();
	b.x0 = x0.data();
	b.y0 = y0.data();
	b.w = w.data();
	b.h = h.data();
	b.out_off = out_off.data();
	return b;
}

void GlyphBatch::append(const GlyphBatch &o)
{
	const uint32_t job_base = (uint32_t)jobs.size();
	const uint32_t seg_base = seg_off.back();
	const uint64_t out_base = out_off.back();
	jobs.insert(jobs.end(), o.jobs.begin(), o.jobs.end());
	for (uint32_t j : o.raster_job)
		raster_job.push_back(job_base + j);
	for (size_t i = 1; i < o.seg_off.size(); i++)
		seg_off.push_back(seg_base + o.seg_off[i]);
	sx.insert(sx.end(), o.sx.begin(), o.sx.end());
	sy.insert(sy.end(), o.sy.begin(), o.sy.end());
	ex.insert(ex.end(), o.ex.begin(), o.ex.end());
	ey.insert(ey.end(), o.ey.begin(), o.ey.end());
	x0.insert(x0.end(), o.x0.begin(), o.x0.end());
	y0.insert(y0.end(), o.y0.begin(), o.y0.end());
	w.insert(w.end(), o.w.begin(), o.w.end());
	h.insert(h.end(), o.h.begin(), o.h.end());
	for (size_t i = 1; i < o.out_off.size(); i++)
		out_off.push_back(out_base + o.out_off[i]);
}

void PackedBatch::reserve(uint32_t rasters, uint64_t segs, uint64_t pixels)
{
	seg_off.ensure((size_t)rasters + 1);
	out_off.ensure((size_t)rasters + 1);
	x0.ensure(rasters);
	y0.ensure(rasters);
	w.ensure(rasters);
	h.ensure(rasters);
	sx.ensure(segs);
	sy.ensure(segs);
	ex.ensure(segs);
	ey.ensure(segs);
	out.ensure(pixels);
	seg_off[0] = 0;
	out_off[0] = 0;
}

vgsdf_batch PackedBatch::view() const
{
	vgsdf_batch b;
	b.n_glyphs = n_raster;
	b.seg_off = seg_off.data();
	b.seg_sx = sx.data();
	b.seg_sy = sy.data();
	b.seg_ex = ex.data();
	b.seg_ey = ey.data();
	b.x0 = x0.data();
	b.y0 = y0.data();
	b.w = w.data();
	b.h = h.data();
	b.out_off = out_off.data();
	return b;
}

std::shared_ptr<Renderer> Renderer::create(bool dummy, int device, std::string *err)
{
	return dummy ? new_dummy() : new_precise(device, err);
}

std::shared_ptr<Renderer> Renderer::new_dummy()
{
	std::shared_ptr<Renderer> r(new Renderer());
	r->mode_ = Mode::Dummy;
	return r;
}

std::shared_ptr<Renderer> Renderer::new_precise(int device, std::string *err)
{
	std::shared_ptr<Renderer> r(new Renderer());
	r->mode_ = Mode::Hip;
	r->device_ = device;
	const int rc = vgsdf_create(device, &r->ctx_);
	if (rc != VGSDF_OK) {
		if (err)
			*err = vgsdf_last_error(nullptr);
		return nullptr; // no CPU fallback: the caller must fail
	}
	return r;
}

std::shared_ptr<Renderer> Renderer::new_multi(const std::vector<int> &devices, std::string *err)
{
	if (devices.empty() || devices.size() > 254) {
		if (err)
			*err = "Renderer::new_multi: 1 .. 254 devices";
		return nullptr;
	}
	std::shared_ptr<Renderer> r = new_precise(devices[0], err);
	if (!r)
		return nullptr;
	for (size_t i = 1; i < devices.size(); i++) {
		std::shared_ptr<Renderer> p = new_precise(devices[i], err);
		if (!p)
			return nullptr;
		r->peers_.push_back(std::move(p));
	}
	return r;
}

void Renderer::add_counters(uint64_t blocks, uint64_t glyphs, uint64_t pixels) const
{
	if (mode_ != Mode::Hip)
		return;
	std::lock_guard<std::mutex> lock(mu_);
	vgsdf_add_counters(ctx_, blocks, glyphs, pixels);
}

void Renderer::reset_counters() const
{
	for (size_t i = 0; i < n_devices(); i++) {
		const Renderer &l = device_lane(i);
		if (l.mode_ == Mode::Hip) {
			std::lock_guard<std::mutex> lock(l.mu_);
			vgsdf_reset_counters(l.ctx_);
		}
	}
}

void Renderer::reduce_counters(uint64_t out[3]) const
{
	if (mode_ != Mode::Hip)
		throw std::runtime_error("reduce_counters needs the HIP renderer");
	std::vector<vgsdf_ctx *> ctxs;
	for (size_t i = 0; i < n_devices(); i++)
		ctxs.push_back(device_lane(i).ctx_);
	if (vgsdf_reduce_counters(ctxs.data(), (int)ctxs.size(), out) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_reduce_counters: ") + vgsdf_last_error(ctx_));
}

std::string Renderer::reduce_path() const
{
	if (mode_ != Mode::Hip)
		return "";
	return vgsdf_reduce_path(device_lane(0).ctx_);
}

Renderer::~Renderer()
{
	if (ctx2_)
		vgsdf_destroy(ctx2_);
	if (ctx_)
		vgsdf_destroy(ctx_);
}

bool Renderer::prepare(const Face &face, uint32_t index, TessScratch &scratch, GlyphBatch &batch)
{
	// renderer.rs:104 char::from_u32
	if (index > 0x10FFFF || (index >= 0xD800 && index <= 0xDFFF))
		return false;
	const auto glyph_id = face.glyph_index(index); // :106
	if (!glyph_id)
		return false;
	const double scale = (double)GLYPH_SIZE / (double)face.units_per_em(); // :107

	scratch.builder.reset(); // :109-111
	face.outline_glyph(*glyph_id, scratch.builder);
	Rings &rings = scratch.builder.into_rings();

	const double advance_float = (double)face.glyph_hor_advance(*glyph_id).value_or(0) * scale * 0.95; // :115
	const uint32_t advance = to_u32(std::round(advance_float));                                          // :116

	GlyphJob job;
	job.id = index;
	job.advance = advance;
	if (rings.is_empty()) { // :118-120
		batch.jobs.push_back(job);
		return true;
	}
	rings.scale(scale);                                           // :122
	const double dx = ((double)advance - advance_float) / 2.0;   // :130
	rings.translate(Point{dx, 0.0});                              // :131

	const BBox bbox = rings.get_bbox(); // prepare_glyph :64-91
	if (bbox.is_empty()) {
		batch.jobs.push_back(job);
		return true;
	}
	job.x0 = to_i32(std::floor(bbox.min.x)) - BUFFER;
	job.y0 = to_i32(std::floor(bbox.min.y)) - BUFFER;
	job.x1 = to_i32(std::ceil(bbox.max.x)) + BUFFER;
	job.y1 = to_i32(std::ceil(bbox.max.y)) + BUFFER;
	job.width = (uint32_t)(job.x1 - job.x0);
	job.height = (uint32_t)(job.y1 - job.y0);
	job.has_raster = true;

	// Rings::get_segments (rings.rs:75-81): consecutive point pairs, ring by ring
	const std::vector<Point> &pts = rings.points();
	const std::vector<uint32_t> &st = rings.starts();
	const size_t nseg = rings.segment_count();
	const size_t base = batch.sx.size();
	batch.sx.resize(base + nseg);
	batch.sy.resize(base + nseg);
	batch.ex.resize(base + nseg);
	batch.ey.resize(base + nseg);
	size_t k = base;
	for (size_t r = 0; r + 1 < st.size(); r++)
		for (uint32_t i = st[r]; i + 1 < st[r + 1]; i++, k++) {
			batch.sx[k] = pts[i].x;
			batch.sy[k] = pts[i].y;
			batch.ex[k] = pts[i + 1].x;
			batch.ey[k] = pts[i + 1].y;
		}
	job.n_segments = (uint32_t)nseg;
	batch.raster_job.push_back((uint32_t)batch.jobs.size());
	batch.jobs.push_back(job);
	batch.seg_off.push_back((uint32_t)(base + nseg));
	batch.x0.push_back(job.x0);
	batch.y0.push_back(job.y0);
	batch.w.push_back(job.width);
	batch.h.push_back(job.height);
	batch.out_off.push_back(batch.out_off.back() + (uint64_t)job.width * job.height);
	return true;
}

bool Renderer::record(const Face &face, uint32_t index, OutlineBatch &batch)
{
	if (index > 0x10FFFF || (index >= 0xD800 && index <= 0xDFFF)) // renderer.rs:104
		return false;
	const auto glyph_id = face.glyph_index(index); // :106
	if (!glyph_id)
		return false;
	const double scale = (double)GLYPH_SIZE / (double)face.units_per_em(); // :107
	CommandRecorder rec(batch.cmds);
	face.outline_glyph(*glyph_id, rec); // :109-111, callbacks only
	const double advance_float = (double)face.glyph_hor_advance(*glyph_id).value_or(0) * scale * 0.95; // :115
	const uint32_t advance = to_u32(std::round(advance_float));                                          // :116
	GlyphJob job;
	job.id = index;
	job.advance = advance;
	batch.jobs.push_back(job);
	batch.cmd_off.push_back((uint32_t)batch.cmds.size());
	batch.dat_off.push_back(batch.dat_off.back() + rec.n_floats());
	batch.scale.push_back(scale);
	batch.shift_x.push_back(((double)advance - advance_float) / 2.0); // :130
	return true;
}

bool Renderer::record(const Face &face, uint32_t index, PackedOutlineBatch &batch)
{
	if (index > 0x10FFFF || (index >= 0xD800 && index <= 0xDFFF)) // renderer.rs:104
		return false;
	const auto glyph_id = face.glyph_index(index); // :106
	if (!glyph_id)
		return false;
	const double scale = (double)GLYPH_SIZE / (double)face.units_per_em(); // :107
	face.outline_glyph_packed(*glyph_id, batch.kinds, batch.coords); // :109-111, callbacks only
	const double advance_float = (double)face.glyph_hor_advance(*glyph_id).value_or(0) * scale * 0.95; // :115
	const uint32_t advance = to_u32(std::round(advance_float));                                          // :116
	GlyphJob job;
	job.id = index;
	job.advance = advance;
	batch.jobs.push_back(job);
	batch.cmd_off.push_back((uint32_t)batch.kinds.size());
	batch.dat_off.push_back((uint32_t)batch.coords.size());
	batch.scale.push_back(scale);
	batch.shift_x.push_back(((double)advance - advance_float) / 2.0); // :130
	return true;
}

bool Renderer::record_parts(const Face &face, uint32_t index, GlyfPartsBatch &batch)
{
	static_assert(sizeof(GlyfPart) == sizeof(vgsdf_glyf_part), "the host's part record is the ABI's, field for field");
	if (index > 0x10FFFF || (index >= 0xD800 && index <= 0xDFFF)) // renderer.rs:104
		return false;
	const auto glyph_id = face.glyph_index(index); // :106
	if (!glyph_id)
		return false;
	const double scale = (double)GLYPH_SIZE / (double)face.units_per_em(); // :107
	(void)face.glyph_parts(*glyph_id, batch.parts, batch.bytes, batch.slots, &batch.overflow); // :109-111: the outline, undecoded
	const double advance_float = (double)face.glyph_hor_advance(*glyph_id).value_or(0) * scale * 0.95; // :115
	const uint32_t advance = to_u32(std::round(advance_float));                                          // :116
	GlyphJob job;
	job.id = index;
	job.advance = advance;
	batch.jobs.push_back(job);
	batch.slot_off.push_back(batch.slots);
	batch.part_off.push_back((uint32_t)batch.parts.size());
	batch.scale.push_back(scale);
	batch.shift_x.push_back(((double)advance - advance_float) / 2.0); // :130
	return true;
}

vgsdf_ctx *Renderer::lane_ctx(int lane) const
{
	if (lane == 0)
		return ctx_;
	std::lock_guard<std::mutex> lock(mu_);
	if (!ctx2_ && vgsdf_create(device_, &ctx2_) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_create (second lane): ") + vgsdf_last_error(nullptr));
	return ctx2_;
}

void Renderer::submit_outlines(int lane, const vgsdf_outlines_packed &v, HostBuffer<uint8_t> &out) const
{
	if (mode_ != Mode::Hip)
		throw std::runtime_error("render_outlines needs the HIP renderer (the device front-end has no CPU form)");
	lane &= 1;
	vgsdf_ctx *c = lane_ctx(lane);
	lane_mu_[lane].lock();
	// one submission: the raster writes into `out` as it stands (capacity kept from earlier groups; first guess
	// 448 bytes per glyph, the average of the fixture fonts) — a second step in wait only when that was too small
	try {
		if (out.capacity() == 0)
			out.ensure((size_t)v.n_glyphs * 480 + 16384);
		std::lock_guard<std::mutex> lock(mu_);
		if (vgsdf_outlines_submit_packed(c, &v, out.data(), out.capacity()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_submit_packed: ") + vgsdf_last_error(c));
	} catch (...) {
		lane_mu_[lane].unlock();
		throw;
	}
}

void Renderer::submit_outlines(int lane, const vgsdf_outlines_glyf &v, HostBuffer<uint8_t> &out) const
{
	if (mode_ != Mode::Hip)
		throw std::runtime_error("render_outlines needs the HIP renderer (the device front-end has no CPU form)");
	lane &= 1;
	vgsdf_ctx *c = lane_ctx(lane);
	lane_mu_[lane].lock();
	// one submission: the raster writes into `out` as it stands (capacity kept from earlier groups; first guess
	// 448 bytes per glyph, the average of the fixture fonts) — a second step in wait only when that was too small
	try {
		if (out.capacity() == 0)
			out.ensure((size_t)v.n_glyphs * 480 + 16384);
		std::lock_guard<std::mutex> lock(mu_);
		if (vgsdf_outlines_submit_glyf(c, &v, out.data(), out.capacity()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_submit_glyf: ") + vgsdf_last_error(c));
	} catch (...) {
		lane_mu_[lane].unlock();
		throw;
	}
}

bool Renderer::peek_outlines(int lane, std::vector<vgsdf_rect> &rects, uint64_t &out_bytes, uint32_t n_glyphs,
                             std::vector<uint64_t> *pbf_at) const
{
	lane &= 1;
	vgsdf_ctx *c = lane == 0 ? ctx_ : ctx2_;
	rects.assign(n_glyphs, vgsdf_rect{});
	out_bytes = 0;
	int in_place = 0;
	std::unique_lock<std::mutex> lock(mu_, std::defer_lock);
	if (lane == 0)
		lock.lock();
	if (vgsdf_outlines_peek(c, rects.data(), &out_bytes, &in_place) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_outlines_peek: ") + vgsdf_last_error(c));
	if (in_place && pbf_at) {
		pbf_at->assign(n_glyphs, 0);
		if (n_glyphs && vgsdf_outlines_pbf_positions(c, pbf_at->data()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_pbf_positions: ") + vgsdf_last_error(c));
	}
	return in_place != 0;
}

void Renderer::wait_outlines(int lane, std::vector<vgsdf_rect> &rects, HostBuffer<uint8_t> &out, uint64_t &out_bytes,
                             uint64_t &n_segments, uint32_t n_glyphs, std::vector<uint64_t> *pbf_at) const
{
	lane &= 1;
	vgsdf_ctx *c = lane == 0 ? ctx_ : ctx2_;
	out_bytes = n_segments = 0;
	struct Unlock {
		std::mutex &m;
		~Unlock() { m.unlock(); }
	} unlock{lane_mu_[lane]};
	rects.assign(n_glyphs, vgsdf_rect{});
	int rendered = 0;
	// lane 0 shares its context with the one-call entry points (render_batch, ...): keep them out while it is in use
	std::unique_lock<std::mutex> lock(mu_, std::defer_lock);
	if (lane == 0)
		lock.lock();
	if (const int rc = vgsdf_outlines_wait(c, rects.data(), &out_bytes, &n_segments, &rendered); rc != VGSDF_OK) {
		if (rc == VGSDF_E_GLYF)
			throw GlyfEntryError(std::string("vgsdf_outlines_wait: ") + vgsdf_last_error(c));
		throw std::runtime_error(std::string("vgsdf_outlines_wait: ") + vgsdf_last_error(c));
	}
	if (!rendered && out_bytes) {
		out.ensure((size_t)out_bytes + 1);
		if (vgsdf_outlines_render(c, out.data()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_render: ") + vgsdf_last_error(c));
	}
	if (pbf_at) { // in-place PBF assembly: where the device placed the bitmaps in the arena `out`
		pbf_at->assign(n_glyphs, 0);
		if (n_glyphs && vgsdf_outlines_pbf_positions(c, pbf_at->data()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_pbf_positions: ") + vgsdf_last_error(c));
	}
}

void Renderer::render_outlines(const vgsdf_outlines &v, std::vector<vgsdf_rect> &rects, HostBuffer<uint8_t> &out,
                               uint64_t &out_bytes, uint64_t &n_segments) const
{
	rects.assign(v.n_glyphs, vgsdf_rect{});
	out_bytes = n_segments = 0;
	if (v.n_glyphs == 0)
		return;
	if (mode_ != Mode::Hip)
		throw std::runtime_error("render_outlines needs the HIP renderer (the device front-end has no CPU form)");
	std::lock_guard<std::mutex> lane(lane_mu_[0]);
	std::lock_guard<std::mutex> lock(mu_);
	if (out.capacity() == 0)
		out.ensure((size_t)v.n_glyphs * 448 + 4096);
	int rendered = 0;
	if (vgsdf_outlines_render_into(ctx_, &v, rects.data(), out.data(), out.capacity(), &out_bytes, &n_segments, &rendered) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_outlines_render_into: ") + vgsdf_last_error(ctx_));
	if (!rendered && out_bytes) {
		out.ensure((size_t)out_bytes + 1);
		if (vgsdf_outlines_render(ctx_, out.data()) != VGSDF_OK)
			throw std::runtime_error(std::string("vgsdf_outlines_render: ") + vgsdf_last_error(ctx_));
	}
}

void Renderer::render_batch(const GlyphBatch &batch, uint8_t *out) const
{
	if (batch.n_raster() == 0)
		return;
	if (mode_ == Mode::Dummy) { // renderer_dummy.rs:3-5
		std::memset(out, 0, (size_t)batch.out_bytes());
		return;
	}
	std::lock_guard<std::mutex> lock(mu_);
	const vgsdf_batch v = batch.view();
	if (vgsdf_render_batch(ctx_, &v, out) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_render_batch: ") + vgsdf_last_error(ctx_));
}

void Renderer::render_packed(PackedBatch &batch) const
{
	if (batch.n_raster == 0)
		return;
	if (mode_ == Mode::Dummy) { // renderer_dummy.rs:3-5
		std::memset(batch.out.data(), 0, (size_t)batch.out_bytes);
		return;
	}
	std::lock_guard<std::mutex> lock(mu_);
	const vgsdf_batch v = batch.view();
	if (vgsdf_render_batch(ctx_, &v, batch.out.data()) != VGSDF_OK)
		throw std::runtime_error(std::string("vgsdf_render_batch: ") + vgsdf_last_error(ctx_));
}

std::optional<PbfGlyph> Renderer::render_glyph(const Face &face, uint32_t index) const
{
	TessScratch scratch;
	GlyphBatch batch;
	if (!prepare(face, index, scratch, batch))
		return std::nullopt;
	const GlyphJob &job = batch.jobs.front();
	if (!job.has_raster)
		return PbfGlyph::empty(job.id, job.advance);
	std::vector<uint8_t> bitmap((size_t)batch.out_bytes());
	render_batch(batch, bitmap.data());
	const PbfGlyphRef r = job.to_pbf(bitmap.data());
	PbfGlyph g;
	g.id = r.id;
	g.bitmap = std::move(bitmap);
	g.width = r.width;
	g.height = r.height;
	g.left = r.left;
	g.top = r.top;
	g.advance = r.advance;
	return g;
}

} // namespace vg
