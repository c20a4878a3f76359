// capi.cpp — flat C view (include/vgfont.h) of the C++ host façade.
#include <cstring>
#include <stdexcept>

#include <unistd.h>

#include "../../../include/vgfont.h"
#include "font_manager.hpp"
#include "index_files.hpp"
#include "writers.hpp"

namespace {
thread_local std::string g_err;
int fail(const std::string &m)
{
	g_err = m;
	return -1;
}
} // namespace

struct vg_renderer {
	std::shared_ptr<vg::Renderer> r;
};
struct vg_manager {
	vg::FontManager m;
	explicit vg_manager(bool p) : m(p) {}
};
struct vg_writer {
	std::unique_ptr<vg::Writer> w;
};
struct vg_outline_batch {
	vg::OutlineBatch b;
	std::vector<uint32_t> ids, advances;
};
struct vg_glyf_batch {
	vg::GlyfPartsBatch b;
	std::vector<uint32_t> ids, advances;
};
struct vg_glyph_batch {
	vg::PackedBatch b;
	std::vector<uint32_t> ids;
	uint32_t n_jobs = 0;
};

namespace {
struct CallbackWriter final : vg::Writer {
	vg_write_cb cb;
	void *user;
	CallbackWriter(vg_write_cb c, void *u) : cb(c), user(u) {}
	void write_directory(const std::string &p) override
	{
		if (cb && cb(user, p.c_str(), nullptr, 0, 1) != 0)
			throw std::runtime_error("writer callback failed for " + p);
	}
	void write_file(const std::string &p, const std::vector<uint8_t> &d) override { write_bytes(p, d.data(), d.size()); }
	void write_bytes(const std::string &p, const uint8_t *d, size_t n) override
	{
		if (cb && cb(user, p.c_str(), d, n, 0) != 0)
			throw std::runtime_error("writer callback failed for " + p);
	}
	void write_gather(const std::string &p, const Piece *pieces, size_t n) override
	{
		if (cb) // (the callback takes one buffer: joined by the default; the NULL sink has nothing to join for)
			vg::Writer::write_gather(p, pieces, n);
	}
};
} // namespace

extern "C" {

const char *vg_last_error(void) { return g_err.c_str(); }

vg_renderer *vg_renderer_new(int mode, int device)
{
	std::string err;
	auto r = mode == VG_MODE_DUMMY ? vg::Renderer::new_dummy() : vg::Renderer::new_precise(device, &err);
	if (!r) {
		g_err = err;
		return nullptr;
	}
	return new vg_renderer{std::move(r)};
}
vg_renderer *vg_renderer_new_multi(const int *devices, int n)
{
	if (!devices || n <= 0) {
		g_err = "vg_renderer_new_multi: no devices";
		return nullptr;
	}
	std::string err;
	auto r = vg::Renderer::new_multi(std::vector<int>(devices, devices + n), &err);
	if (!r) {
		g_err = err;
		return nullptr;
	}
	return new vg_renderer{std::move(r)};
}
int vg_renderer_device_count(const vg_renderer *r) { return (int)r->r->n_devices(); }
int vg_renderer_reduce_counters(const vg_renderer *r, uint64_t counters[3])
{
	try {
		r->r->reduce_counters(counters);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}
const char *vg_renderer_reduce_path(const vg_renderer *r)
{
	static thread_local std::string keep;
	try {
		keep = r->r->reduce_path();
	} catch (const std::exception &) {
		keep.clear();
	}
	return keep.c_str();
}
void vg_renderer_add_counters(const vg_renderer *r, int lane, uint64_t blocks, uint64_t glyphs, uint64_t pixels)
{
	if (lane >= 0 && (size_t)lane < r->r->n_devices())
		r->r->device_lane((size_t)lane).add_counters(blocks, glyphs, pixels);
}
void vg_renderer_reset_counters(const vg_renderer *r) { r->r->reset_counters(); }
void vg_renderer_free(vg_renderer *r) { delete r; }
void vg_manager_reduced_counters(const vg_manager *m, uint64_t counters[3]) { std::memcpy(counters, m->m.last_reduced_counters(), 3 * sizeof(uint64_t)); }

vg_manager *vg_manager_new(int parallel) { return new vg_manager(parallel != 0); }
void vg_manager_free(vg_manager *m) { delete m; }
void vg_manager_set_device_front_end(vg_manager *m, int on) { m->m.set_device_front_end(on != 0); }
void vg_manager_set_in_place_pbf(vg_manager *m, int on) { m->m.set_in_place_pbf(on != 0); }
void vg_manager_set_glyf_on_device(vg_manager *m, int on) { m->m.set_glyf_on_device(on != 0); }
void vg_manager_set_lane_form(vg_manager *m, int form) { m->m.set_lane_form(form < 0 || form > 2 ? -1 : form); }
void vg_manager_set_threads(vg_manager *m, unsigned threads, unsigned blocks_per_batch)
{
	m->m.set_threads(threads);
	if (blocks_per_batch)
		m->m.set_batch_blocks(blocks_per_batch);
}

int vg_manager_add_font_with_name(vg_manager *m, const char *name, const char *const *paths, int n)
{
	std::vector<std::string> ps(paths, paths + n);
	std::string err;
	return m->m.add_font_with_name(name, ps, &err) ? 0 : fail(err);
}
int vg_manager_add_font_data(vg_manager *m, const char *name, const uint8_t *data, size_t len)
{
	std::string err;
	return m->m.add_font_data(name, std::vector<uint8_t>(data, data + len), &err) ? 0 : fail(err);
}
int vg_manager_add_path(vg_manager *m, const char *path)
{
	std::string err;
	return m->m.add_path(path, &err) ? 0 : fail(err);
}
int vg_manager_shard_glyphs(const vg_manager *m, const char *font_id, uint32_t world, uint8_t *owner, double *cost)
{
	vg::GlyphShard sh;
	std::string err;
	if (!m->m.shard_glyphs(font_id, world, sh, &err))
		return fail(err);
	if (owner)
		std::memcpy(owner, sh.owner.data(), sh.owner.size());
	if (cost)
		std::memcpy(cost, sh.cost.data(), sh.cost.size() * sizeof(double));
	return 0;
}

int vg_manager_plan_lanes(vg_manager *m, const char *font_id, uint32_t world, uint8_t *owner, uint32_t *n_split_blocks, double *est_max_over_mean)
{
	try {
		std::vector<uint8_t> own;
		uint32_t n_split = 0;
		std::string err;
		if (!m->m.plan_lanes(font_id, world, own, n_split, est_max_over_mean, &err))
			return fail(err);
		if (owner)
			std::memcpy(owner, own.data(), own.size());
		if (n_split_blocks)
			*n_split_blocks = n_split;
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_manager_set_glyph_shard(vg_manager *m, uint32_t rank, uint32_t world)
{
	try {
		m->m.set_glyph_shard(rank, world);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

long vg_pbf_merge(const uint8_t *const *parts, const size_t *lens, int n, uint8_t *out, size_t cap)
{
	try {
		std::vector<std::pair<const uint8_t *, size_t>> ps;
		for (int i = 0; i < n; i++)
			ps.emplace_back(parts[i], lens[i]);
		const std::vector<uint8_t> v = vg::merge_pbf_partials(ps);
		if (out && cap >= v.size())
			std::memcpy(out, v.data(), v.size());
		return (long)v.size();
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

long vg_pbf_concat(const uint8_t *const *parts, const size_t *lens, int n, uint8_t *out, size_t cap)
{
	try {
		std::vector<std::pair<const uint8_t *, size_t>> ps;
		for (int i = 0; i < n; i++)
			ps.emplace_back(parts[i], lens[i]);
		const std::vector<uint8_t> v = vg::concat_pbf_partials(ps);
		if (out && cap >= v.size())
			std::memcpy(out, v.data(), v.size());
		return (long)v.size();
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_manager_scan(vg_manager *m, const char *path)
{
	std::string err;
	return m->m.scan(path, &err) ? 0 : fail(err);
}

static long copy_out(const std::string &s, char *out, size_t cap, bool nul)
{
	if (out && cap) {
		const size_t n = std::min(nul ? cap - 1 : cap, s.size());
		std::memcpy(out, s.data(), n);
		if (nul)
			out[n] = 0;
	}
	return (long)(s.size() + (nul ? 1 : 0));
}

long vg_manager_font_ids(const vg_manager *m, char *out, size_t cap)
{
	std::string s;
	for (const auto &kv : m->m.fonts()) {
		if (!s.empty())
			s.push_back('\n');
		s += kv.first;
	}
	return copy_out(s, out, cap, true);
}

long vg_manager_font_file_names(const vg_manager *m, const char *font_id, char *out, size_t cap)
{
	auto it = m->m.fonts().find(font_id);
	if (it == m->m.fonts().end())
		return fail(std::string("unknown font id ") + font_id);
	std::string s;
	for (const auto &f : it->second.files()) {
		if (!s.empty())
			s.push_back('\n');
		s += f->metadata().name;
	}
	return copy_out(s, out, cap, true);
}

int vg_parse_font_name(const char *family, const char *ps_name, char *family_out, size_t cap, char *style_out,
                       uint16_t *weight_out, char *width_out)
{
	const vg::ParsedFontName p = vg::parse_font_name(family ? family : "", ps_name ? ps_name : "");
	copy_out(p.family, family_out, cap, true);
	if (style_out)
		copy_out(p.style, style_out, 16, true);
	if (width_out)
		copy_out(p.width, width_out, 16, true);
	if (weight_out)
		*weight_out = p.weight;
	return (int)p.family.size();
}

long vg_manager_generate_name(const vg_manager *m, const char *font_id, int file_index, char *out, size_t cap)
{
	auto it = m->m.fonts().find(font_id);
	if (it == m->m.fonts().end() || file_index < 0 || (size_t)file_index >= it->second.files().size())
		return fail("unknown font id / file index");
	return copy_out(it->second.files()[(size_t)file_index]->metadata().generate_name(), out, cap, true);
}

long vg_encode_codeblocks(const uint32_t *codepoints, size_t n, char *out, size_t cap)
{
	return copy_out(vg::encode_codeblocks(std::vector<uint32_t>(codepoints, codepoints + n)), out, cap, true);
}

long vg_manager_index_json(const vg_manager *m, uint8_t *out, size_t cap)
{
	const auto v = vg::build_index_json(m->m);
	return copy_out(std::string(v.begin(), v.end()), (char *)out, cap, false);
}

long vg_manager_families_json(const vg_manager *m, uint8_t *out, size_t cap)
{
	try {
		const auto v = vg::build_font_families_json(m->m);
		return copy_out(std::string(v.begin(), v.end()), (char *)out, cap, false);
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

vg_writer *vg_writer_new_tar_path(const char *path, int64_t mtime)
{
	std::FILE *f = std::fopen(path, "wb");
	if (!f) {
		g_err = std::string("cannot create ") + path;
		return nullptr;
	}
	return new vg_writer{std::unique_ptr<vg::Writer>(new vg::TarWriter(f, true, mtime))};
}

vg_writer *vg_writer_new_tar_fd(int fd, int64_t mtime)
{
	const int dupfd = ::dup(fd); // fclose() of our stream must not close the caller's descriptor
	std::FILE *f = dupfd >= 0 ? ::fdopen(dupfd, "wb") : nullptr;
	if (!f) {
		if (dupfd >= 0)
			::close(dupfd);
		g_err = "cannot open descriptor " + std::to_string(fd) + " for writing";
		return nullptr;
	}
	return new vg_writer{std::unique_ptr<vg::Writer>(new vg::TarWriter(f, true, mtime))};
}

vg_writer *vg_writer_new_dir(const char *folder) { return new vg_writer{std::unique_ptr<vg::Writer>(new vg::FileWriter(folder))}; }

int vg_writer_write_file(vg_writer *w, const char *path, const uint8_t *data, size_t len)
{
	try {
		w->w->write_file(path, std::vector<uint8_t>(data, data + len));
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_writer_write_directory(vg_writer *w, const char *path)
{
	try {
		w->w->write_directory(path);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_writer_finish(vg_writer *w)
{
	try {
		w->w->finish();
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

void vg_writer_free(vg_writer *w) { delete w; }

int vg_manager_render_glyphs_to(vg_manager *m, vg_renderer *r, vg_writer *w)
{
	try {
		m->m.render_glyphs(*w->w, *r->r);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_manager_write_index_json(const vg_manager *m, vg_writer *w)
{
	try {
		m->m.write_index_json(*w->w);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_manager_write_families_json(const vg_manager *m, vg_writer *w)
{
	try {
		m->m.write_families_json(*w->w);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_name_to_id(const char *name, char *out, size_t cap)
{
	const std::string id = vg::name_to_id(name);
	if (out && cap) {
		const size_t n = std::min(cap - 1, id.size());
		std::memcpy(out, id.data(), n);
		out[n] = 0;
	}
	return (int)id.size();
}
int vg_manager_block_counts(const vg_manager *m, const char *font_id, uint32_t counts[256])
{
	auto it = m->m.fonts().find(font_id);
	if (it == m->m.fonts().end())
		return fail(std::string("unknown font id ") + font_id);
	const auto blocks = it->second.get_blocks();
	for (size_t i = 0; i < 256; i++)
		counts[i] = (uint32_t)blocks[i].len();
	return 0;
}

int vg_manager_render_glyphs(vg_manager *m, vg_renderer *r, vg_write_cb cb, void *user)
{
	try {
		CallbackWriter w(cb, user);
		m->m.render_glyphs(w, *r->r);
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}
int vg_manager_render_blocks(vg_manager *m, vg_renderer *r, const char *font_id, const uint32_t *starts, int n,
                             vg_write_cb cb, void *user)
{
	try {
		CallbackWriter w(cb, user);
		m->m.render_blocks(w, *r->r, font_id, std::vector<uint32_t>(starts, starts + n));
		return 0;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}
int vg_manager_timings(const vg_manager *m, vg_timings *out)
{
	const vg::RenderTimings &t = m->m.last_timings();
	*out = vg_timings{t.tessellate_s, t.pack_s, t.device_s, t.encode_s, t.write_s, t.total_s, t.blocks,
	                  t.glyphs,       t.rasters,  t.pixels,   t.segments, t.pbf_bytes,  t.glyf_groups, t.glyf_fallbacks};
	return 0;
}
long vg_manager_render_block(vg_manager *m, vg_renderer *r, const char *font_id, uint32_t start, uint8_t *out,
                             size_t cap)
{
	try {
		auto it = m->m.fonts().find(font_id);
		if (it == m->m.fonts().end())
			return fail(std::string("unknown font id ") + font_id);
		if (start % 256 || start > 0xFF00)
			return fail("block start must be a multiple of 256 below 65536");
		const auto blocks = it->second.get_blocks();
		const std::vector<uint8_t> pbf = blocks[start / 256].render(font_id, *r->r);
		if (out && pbf.size() <= cap)
			std::memcpy(out, pbf.data(), pbf.size());
		return (long)pbf.size();
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

int vg_render_glyph(vg_renderer *r, const vg_manager *m, const char *font_id, int file_index, uint32_t index,
                    vg_pbf_glyph *out, uint8_t *bitmap, size_t cap)
{
	try {
		auto it = m->m.fonts().find(font_id);
		if (it == m->m.fonts().end())
			return fail(std::string("unknown font id ") + font_id);
		const auto &files = it->second.files();
		if (file_index < 0 || (size_t)file_index >= files.size())
			return fail("file index out of range");
		const auto g = r->r->render_glyph(files[(size_t)file_index]->face(), index);
		if (!g)
			return 0;
		out->id = g->id;
		out->has_bitmap = g->bitmap ? 1 : 0;
		out->width = g->width;
		out->height = g->height;
		out->left = g->left;
		out->top = g->top;
		out->advance = g->advance;
		out->bitmap_len = g->bitmap ? (uint32_t)g->bitmap->size() : 0;
		if (g->bitmap) {
			if (g->bitmap->size() > cap)
				return fail("bitmap buffer too small");
			std::memcpy(bitmap, g->bitmap->data(), g->bitmap->size());
		}
		return 1;
	} catch (const std::exception &e) {
		return fail(e.what());
	}
}

vg_glyph_batch *vg_manager_build_batch(vg_manager *m, const char *font_id)
{
	try {
		auto *b = new vg_glyph_batch();
		std::string err;
		if (!m->m.build_batch(font_id, b->b, b->ids, b->n_jobs, &err)) {
			delete b;
			g_err = err;
			return nullptr;
		}
		return b;
	} catch (const std::exception &e) {
		g_err = e.what();
		return nullptr;
	}
}
int vg_glyph_batch_view(const vg_glyph_batch *b, vgsdf_batch *view, const uint32_t **ids, uint32_t *n_jobs)
{
	*view = b->b.view();
	if (ids)
		*ids = b->ids.data();
	if (n_jobs)
		*n_jobs = b->n_jobs;
	return 0;
}
void vg_glyph_batch_free(vg_glyph_batch *b) { delete b; }

vg_outline_batch *vg_manager_record_outlines(const vg_manager *m, const char *font_id)
{
	try {
		auto *b = new vg_outline_batch();
		std::string err;
		if (!m->m.record_outlines(font_id, b->b, &err)) {
			delete b;
			g_err = err;
			return nullptr;
		}
		for (const vg::GlyphJob &j : b->b.jobs) {
			b->ids.push_back(j.id);
			b->advances.push_back(j.advance);
		}
		return b;
	} catch (const std::exception &e) {
		g_err = e.what();
		return nullptr;
	}
}
int vg_outline_batch_view(const vg_outline_batch *b, vgsdf_outlines *view, const uint32_t **ids, const uint32_t **advances)
{
	*view = b->b.view();
	if (ids)
		*ids = b->ids.data();
	if (advances)
		*advances = b->advances.data();
	return 0;
}
void vg_outline_batch_free(vg_outline_batch *b) { delete b; }

vg_glyf_batch *vg_manager_record_glyf_parts(const vg_manager *m, const char *font_id)
{
	try {
		auto *b = new vg_glyf_batch();
		std::string err;
		if (!m->m.record_glyf_parts(font_id, b->b, &err)) {
			delete b;
			g_err = err;
			return nullptr;
		}
		for (const vg::GlyphJob &j : b->b.jobs) {
			b->ids.push_back(j.id);
			b->advances.push_back(j.advance);
		}
		return b;
	} catch (const std::exception &e) {
		g_err = e.what();
		return nullptr;
	}
}
int vg_glyf_batch_view(const vg_glyf_batch *b, vgsdf_outlines_glyf *view, const uint32_t **ids, const uint32_t **advances)
{
	static_assert(sizeof(vg::GlyfPart) == sizeof(vgsdf_glyf_part), "same record");
	const vg::GlyfPartsBatch &g = b->b;
	view->n_glyphs = (uint32_t)g.jobs.size();
	view->n_parts = (uint32_t)g.parts.size();
	view->n_bytes = (uint32_t)g.bytes.size();
	view->cmd_off = g.slot_off.data();
	view->parts = reinterpret_cast<const vgsdf_glyf_part *>(g.parts.data());
	view->bytes = g.bytes.data();
	view->scale = g.scale.data();
	view->shift_x = g.shift_x.data();
	view->pbf_pre = nullptr;
	view->pbf_fix = nullptr;
	if (ids)
		*ids = b->ids.data();
	if (advances)
		*advances = b->advances.data();
	return 0;
}
void vg_glyf_batch_free(vg_glyf_batch *b) { delete b; }

long vg_pbf_encode(const char *name, const char *range, const vg_pbf_glyph *glyphs, const uint8_t *const *bitmaps,
                   int n, uint8_t *out, size_t cap)
{
	std::vector<vg::PbfGlyphRef> refs((size_t)n);
	static const uint8_t kEmpty = 0;
	for (int i = 0; i < n; i++) {
		vg::PbfGlyphRef &r = refs[(size_t)i];
		r.id = glyphs[i].id;
		if (glyphs[i].has_bitmap) {
			r.bitmap = bitmaps && bitmaps[i] ? bitmaps[i] : &kEmpty;
			r.bitmap_len = glyphs[i].bitmap_len;
		}
		r.width = glyphs[i].width;
		r.height = glyphs[i].height;
		r.left = glyphs[i].left;
		r.top = glyphs[i].top;
		r.advance = glyphs[i].advance;
	}
	const std::vector<uint8_t> pbf = vg::PbfGlyphs::encode(name, range, std::move(refs));
	if (out && pbf.size() <= cap)
		std::memcpy(out, pbf.data(), pbf.size());
	return (long)pbf.size();
}

} // extern "C"
