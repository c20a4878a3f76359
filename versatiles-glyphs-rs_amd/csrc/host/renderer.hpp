// renderer.hpp — host mirror of the reference's Renderer (src/render/renderer.rs) with
// the SDF raster moved to the GPU.
//
//   Renderer::{new,new_precise,new_dummy}   renderer.rs:25-43
//   Renderer::prepare_glyph                 renderer.rs:64-91
//   Renderer::render_glyph                  renderer.rs:103-149
//   RenderResult / into_pbf_glyph           src/render/result.rs:7-29,66-76
//   renderer_dummy                          src/render/renderer_dummy.rs:3-5
//
// The reference dispatches `match self.mode { Precise => renderer_precise(..), Dummy => .. }`
// (renderer.rs:140-143).  Here Precise IS the HIP back-end: there is no CPU raster in this
// library.  render_glyph keeps the per-glyph API; the throughput path is the batched form
// (prepare -> GlyphBatch -> render_batch) that FontManager::render_glyphs drives.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/vgsdf.h"
#include "geometry.hpp"
#include "pbf.hpp"
#include "ring_builder.hpp"
#include "ttf_face.hpp"

namespace vg {

constexpr int32_t GLYPH_SIZE = 24; // src/render/mod.rs:52
constexpr int32_t BUFFER = 3;      // mod.rs:58

// RenderResult (result.rs:7-29) + what into_pbf_glyph needs, minus the pixels
struct GlyphJob {
	uint32_t id = 0;
	uint32_t advance = 0;
	bool has_raster = false; // false => PbfGlyph::empty(id, advance)
	int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
	uint32_t width = 0, height = 0; // including 2*BUFFER
	uint32_t n_segments = 0;

	// result.rs:66-76, with the `glyph.y1 -= GLYPH_SIZE` of renderer.rs:146 applied
	PbfGlyphRef to_pbf(const uint8_t *bitmap) const
	{
		PbfGlyphRef g;
		g.id = id;
		g.advance = advance;
		if (has_raster) {
			g.bitmap = bitmap;
			g.bitmap_len = (size_t)width * height;
			g.width = width - 2 * BUFFER;
			g.height = height - 2 * BUFFER;
			g.left = x0 + BUFFER;
			g.top = (y1 - GLYPH_SIZE) - BUFFER;
		}
		return g;
	}
};

// Host SoA batch in exactly the layout vgsdf_batch wants (include/vgsdf.h).
struct GlyphBatch {
	std::vector<GlyphJob> jobs;       // ALL glyphs, rasterised or empty, in submission order
	std::vector<uint32_t> raster_job; // index into jobs for each rasterised glyph
	std::vector<uint32_t> seg_off{0};
	std::vector<double> sx, sy, ex, ey;
	std::vector<int32_t> x0, y0;
	std::vector<uint32_t> w, h;
	std::vector<uint64_t> out_off{0};

	void clear();
	size_t n_raster() const { return w.size(); }
	uint64_t out_bytes() const { return out_off.back(); }
	vgsdf_batch view() const;
	// appends another batch (jobs keep their relative order)
	void append(const GlyphBatch &o);
};

// Grow-only host array; page-locked (vgsdf_host_alloc) when HIP is available so the device
// layer can DMA it without a staging copy, plain malloc otherwise (dummy renderer on CPU).
template <class T> class HostBuffer {
public:
	explicit HostBuffer(bool want_pinned = false) : want_pinned_(want_pinned) {}
	HostBuffer(const HostBuffer &) = delete;
	HostBuffer &operator=(const HostBuffer &) = delete;
	~HostBuffer() { release(); }
	// contents are NOT preserved across a growth
	void ensure(size_t n)
	{
		if (n <= cap_)
			return;
		release();
		const size_t want = n + n / 4 + 64;
		if (want_pinned_) {
			p_ = static_cast<T *>(vgsdf_host_alloc(want * sizeof(T)));
			pinned_ = p_ != nullptr;
		}
		if (!p_) {
			p_ = static_cast<T *>(std::malloc(want * sizeof(T)));
			if (!p_)
				throw std::bad_alloc();
		}
		cap_ = want;
	}
	T *data() { return p_; }
	const T *data() const { return p_; }
	size_t capacity() const { return cap_; }
	T &operator[](size_t i) { return p_[i]; }
	const T &operator[](size_t i) const { return p_[i]; }

private:
	void release()
	{
		if (p_) {
			if (pinned_)
				vgsdf_host_free(p_);
			else
				std::free(p_);
		}
		p_ = nullptr;
		cap_ = 0;
		pinned_ = false;
	}
	T *p_ = nullptr;
	size_t cap_ = 0;
	bool want_pinned_, pinned_ = false;
};

// The batch as the device boundary wants it (include/vgsdf.h), in reusable buffers; the big
// arrays (segments, output pixels) are page-locked.
struct PackedBatch {
	HostBuffer<uint32_t> seg_off;
	HostBuffer<double> sx{true}, sy{true}, ex{true}, ey{true};
	HostBuffer<int32_t> x0, y0;
	HostBuffer<uint32_t> w, h;
	HostBuffer<uint64_t> out_off;
	HostBuffer<uint8_t> out{true};
	uint32_t n_raster = 0;
	uint64_t n_seg = 0, out_bytes = 0;

	void reserve(uint32_t rasters, uint64_t segs, uint64_t pixels);
	vgsdf_batch view() const;
};

// Outline commands of a set of glyphs for the DEVICE front-end (vgsdf_outlines): the host
// records what ttf-parser's OutlineBuilder receives; flattening, ring rules, scale/shift and
// bbox then run on the GPU (csrc/outline_kernels.hip).
struct OutlineBatch {
	std::vector<GlyphJob> jobs;        // one per glyph the reference returns Some(..) for
	std::vector<uint32_t> cmd_off{0};  // [jobs + 1]
	std::vector<uint32_t> dat_off{0};  // [jobs + 1]: coordinates the commands carry (2 / 4 / 6 / 0 per move or line / quad / curve / close)
	std::vector<vgsdf_outline_cmd> cmds;
	std::vector<double> scale, shift_x;

	void clear()
	{
		jobs.clear();
		cmd_off.assign(1, 0);
		dat_off.assign(1, 0);
		cmds.clear();
		scale.clear();
		shift_x.clear();
	}
	vgsdf_outlines view() const
	{
		vgsdf_outlines o;
		o.n_glyphs = (uint32_t)jobs.size();
		o.cmd_off = cmd_off.data();
		o.cmds = cmds.data();
		o.scale = scale.data();
		o.shift_x = shift_x.data();
		return o;
	}
};

// The same, recorded straight in the compact upload form of vgsdf_outlines_packed (one kind byte per command plus only
// the coordinates its kind carries: ~12 bytes per command of a TrueType font instead of a 28-byte record): what the
// workers of FontManager write, so that merging their batches is a copy.
struct PackedOutlineBatch {
	std::vector<GlyphJob> jobs;
	std::vector<uint32_t> cmd_off{0}, dat_off{0}; // [jobs + 1] into kinds / coords
	std::vector<uint8_t> kinds;
	std::vector<float> coords;
	std::vector<double> scale, shift_x;
	void clear()
	{
		jobs.clear();
		cmd_off.assign(1, 0);
		dat_off.assign(1, 0);
		kinds.clear();
		coords.clear();
		scale.clear();
		shift_x.clear();
	}
};

// wait_outlines on a batch submitted in the glyf form: one of its `glyf` entries is malformed (VGSDF_E_GLYF) — the batch
// has to be recorded on the host, where ttf-parser's rules for such glyphs are applied
struct GlyfEntryError : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// What a worker records for the device's glyf decoder (vgsdf_outlines_glyf): per glyph its metrics and the parts — the
// simple glyphs it is drawn from, their `glyf` arrays copied as they stand.  Offsets are relative to this batch.
struct GlyfPartsBatch {
	std::vector<GlyphJob> jobs;
	std::vector<uint32_t> slot_off{0}, part_off{0}; // [jobs + 1] command slots / parts
	std::vector<GlyfPart> parts;
	std::vector<uint8_t> bytes;
	std::vector<double> scale, shift_x;
	uint32_t slots = 0;
	bool overflow = false; // a glyph's parts passed what 32-bit offsets address (Face::glyph_parts): not a batch for the device
	void clear()
	{
		overflow = false;
		jobs.clear();
		slot_off.assign(1, 0);
		part_off.assign(1, 0);
		parts.clear();
		bytes.clear();
		scale.clear();
		shift_x.clear();
		slots = 0;
	}
};

// The merged batch handed to the device, in the compact upload form (vgsdf_outlines_packed: one kind byte per
// command plus the coordinates its kind carries).  All arrays live back to back in ONE page-locked block, in the
// order vgsdf.h names for a single-copy upload: scale | shift_x | cmd_off | dat_off | (pad to 8) | coords | kinds.
struct MergedOutlines {
	std::vector<GlyphJob> jobs;
	HostBuffer<uint8_t> blob{true};
	uint32_t n_jobs = 0;
	double *scale = nullptr, *shift_x = nullptr;
	uint32_t *cmd_off = nullptr, *dat_off = nullptr;
	float *coords = nullptr;
	uint8_t *kinds = nullptr;
	uint32_t *pbf_pre = nullptr; // in-place PBF assembly (vgsdf.h): bytes reserved in front of a glyph's entry
	uint8_t *pbf_fix = nullptr;  // ... and the lengths of its id / advance fields; NULL when `with_pbf` was false
	// the glyf form (vgsdf_outlines_glyf) in the same block: scale | shift_x | cmd_off | (pad to 8) | parts | bytes [| pbf_pre | pbf_fix]
	bool glyf = false;
	vgsdf_glyf_part *parts = nullptr;
	uint8_t *glyf_bytes = nullptr;
	uint32_t n_parts = 0, n_glyf_bytes = 0;
	void layout_glyf(uint32_t jobs_n, uint32_t parts_n, uint32_t bytes_n, bool with_pbf)
	{
		n_jobs = jobs_n;
		n_parts = parts_n;
		n_glyf_bytes = bytes_n; // (a multiple of 4: every part's bytes are padded)
		glyf = true;
		const size_t n = jobs_n;
		const size_t o_shift = 8 * n, o_cmd = 16 * n;
		const size_t o_parts = (o_cmd + 4 * (n + 1) + 7) & ~(size_t)7, o_bytes = o_parts + sizeof(vgsdf_glyf_part) * (size_t)parts_n;
		const size_t o_pre = o_bytes + bytes_n, o_fix = o_pre + 4 * n;
		blob.ensure((with_pbf ? o_fix + n : o_pre) + 16);
		uint8_t *b = blob.data();
		pbf_pre = with_pbf ? reinterpret_cast<uint32_t *>(b + o_pre) : nullptr;
		pbf_fix = with_pbf ? b + o_fix : nullptr;
		scale = reinterpret_cast<double *>(b);
		shift_x = reinterpret_cast<double *>(b + o_shift);
		cmd_off = reinterpret_cast<uint32_t *>(b + o_cmd);
		dat_off = nullptr;
		coords = nullptr;
		kinds = nullptr;
		parts = reinterpret_cast<vgsdf_glyf_part *>(b + o_parts);
		glyf_bytes = b + o_bytes;
	}
	vgsdf_outlines_glyf view_glyf() const
	{
		vgsdf_outlines_glyf o;
		o.n_glyphs = n_jobs;
		o.n_parts = n_parts;
		o.n_bytes = n_glyf_bytes;
		o.cmd_off = cmd_off;
		o.parts = parts;
		o.bytes = glyf_bytes;
		o.scale = scale;
		o.shift_x = shift_x;
		o.pbf_pre = pbf_pre;
		o.pbf_fix = pbf_fix;
		return o;
	}
	void layout(uint32_t jobs_n, uint32_t n_cmds, uint32_t n_floats, bool with_pbf = false)
	{
		n_jobs = jobs_n;
		glyf = false;
		const size_t n = jobs_n;
		const size_t o_shift = 8 * n, o_cmd = 16 * n, o_dat = o_cmd + 4 * (n + 1);
		const size_t o_coords = (o_dat + 4 * (n + 1) + 7) & ~(size_t)7, o_kinds = o_coords + 4 * (size_t)n_floats;
		const size_t o_pre = (o_kinds + n_cmds + 3) & ~(size_t)3, o_fix = o_pre + 4 * n;
		blob.ensure((with_pbf ? o_fix + n : o_kinds + n_cmds) + 16);
		uint8_t *b = blob.data();
		pbf_pre = with_pbf ? reinterpret_cast<uint32_t *>(b + o_pre) : nullptr;
		pbf_fix = with_pbf ? b + o_fix : nullptr;
		scale = reinterpret_cast<double *>(b);
		shift_x = reinterpret_cast<double *>(b + o_shift);
		cmd_off = reinterpret_cast<uint32_t *>(b + o_cmd);
		dat_off = reinterpret_cast<uint32_t *>(b + o_dat);
		coords = reinterpret_cast<float *>(b + o_coords);
		kinds = b + o_kinds;
	}
	vgsdf_outlines_packed view() const
	{
		vgsdf_outlines_packed o;
		o.n_glyphs = n_jobs;
		o.cmd_off = cmd_off;
		o.dat_off = dat_off;
		o.kinds = kinds;
		o.coords = coords;
		o.scale = scale;
		o.shift_x = shift_x;
		o.pbf_pre = pbf_pre;
		o.pbf_fix = pbf_fix;
		return o;
	}
};

// OutlineBuilder sink that records the callbacks verbatim.
class CommandRecorder final : public OutlineBuilder {
public:
	explicit CommandRecorder(std::vector<vgsdf_outline_cmd> &out) : out_(out) {}
	void move_to(float x, float y) override { n_floats_ += 2, push(0, 0, 0, 0, 0, x, y); }
	void line_to(float x, float y) override { n_floats_ += 2, push(1, 0, 0, 0, 0, x, y); }
	void quad_to(float x1, float y1, float x, float y) override { n_floats_ += 4, push(2, x1, y1, 0, 0, x, y); }
	void curve_to(float x1, float y1, float x2, float y2, float x, float y) override { n_floats_ += 6, push(3, x1, y1, x2, y2, x, y); }
	void close() override { push(4, 0, 0, 0, 0, 0, 0); }
	uint32_t n_floats() const { return n_floats_; } // coordinates the recorded commands carry

private:
	uint32_t n_floats_ = 0;
	void push(uint32_t kind, float x1, float y1, float x2, float y2, float x, float y)
	{
		vgsdf_outline_cmd c;
		c.x1 = x1;
		c.y1 = y1;
		c.x2 = x2;
		c.y2 = y2;
		c.x = x;
		c.y = y;
		c.kind = kind;
		out_.push_back(c);
	}
	std::vector<vgsdf_outline_cmd> &out_;
};

// Per-thread scratch so tessellation allocates nothing in steady state.
struct TessScratch {
	RingBuilder builder;
};

class Renderer {
public:
	enum class Mode { Hip, Dummy };

	// renderer.rs:25-31
	static std::shared_ptr<Renderer> create(bool dummy, int device = 0, std::string *err = nullptr);
	static std::shared_ptr<Renderer> new_precise(int device = 0, std::string *err = nullptr); // HIP back-end
	static std::shared_ptr<Renderer> new_dummy();
	// ONE process, N devices (the reference is one process too, manager.rs:81-125): a HIP renderer on
	// devices[0] plus one peer per further entry, each with its own device contexts and streams.  An entry may repeat a
	// device (N lanes rehearsed on one GPU).  FontManager::render_glyphs deals a font's glyph shards to the lanes, one host
	// thread each; every single-renderer call on the object goes to devices[0].
	static std::shared_ptr<Renderer> new_multi(const std::vector<int> &devices, std::string *err = nullptr);
	size_t n_devices() const { return 1 + peers_.size(); }
	const Renderer &device_lane(size_t i) const { return i == 0 ? *this : *peers_[i - 1]; }
	// run counters {blocks, glyphs, pixels}: credited to the lane that did the work, summed over the lanes by
	// vgsdf_reduce_counters (an RCCL all-reduce when the lanes sit on distinct devices).  Hip mode only.
	void add_counters(uint64_t blocks, uint64_t glyphs, uint64_t pixels) const;
	void reset_counters() const;
	void reduce_counters(uint64_t out[3]) const;
	// how the last reduce_counters took its sum (vgsdf_reduce_path): "rccl", "host: ...", "host: RCCL fallback: <reason>"
	std::string reduce_path() const;
	~Renderer();

	Mode mode() const { return mode_; }
	int device() const { return device_; }

	// Host half of render_glyph, renderer.rs:103-137: cmap lookup, outline, flatten, scale,
	// sub-pixel shift, bbox + buffer.  nullopt = the reference returns None (glyph skipped).
	// Rasterised glyphs get their segments appended to `batch`.
	static bool prepare(const Face &face, uint32_t index, TessScratch &scratch, GlyphBatch &batch);

	// Host half for the DEVICE front-end: cmap lookup, advance, scale / shift, and the raw
	// outline commands (renderer.rs:104-116,130); everything else happens on the GPU.
	static bool record(const Face &face, uint32_t index, OutlineBatch &batch);
	static bool record(const Face &face, uint32_t index, PackedOutlineBatch &batch); // the same into the compact form
	// ... and for the device's glyf decoder: nothing is decoded, the glyph's simple glyphs are appended as parts
	// (glyf fonts only: face.has_glyf_outlines())
	static bool record_parts(const Face &face, uint32_t index, GlyfPartsBatch &batch);
	// Device front-end + raster for a recorded batch: fills rects (one per job) and `out` with
	// the bitmaps of the glyphs that have a raster, packed in job order.  Hip mode only.
	void render_outlines(const vgsdf_outlines &batch, std::vector<vgsdf_rect> &rects, HostBuffer<uint8_t> &out,
	                     uint64_t &out_bytes, uint64_t &n_segments) const;
	// The same in two halves on one of two lanes (each lane = its own device context and stream): submit enqueues
	// upload, front-end and raster and returns; wait collects.  A caller that alternates the lanes keeps the GPU
	// busy while it records the next batch and encodes the previous one.  `batch` (its command array) and `out`
	// must stay untouched between the two calls; a lane is held from submit to wait.
	void submit_outlines(int lane, const vgsdf_outlines_packed &batch, HostBuffer<uint8_t> &out) const;
	void submit_outlines(int lane, const vgsdf_outlines_glyf &batch, HostBuffer<uint8_t> &out) const;
	void wait_outlines(int lane, std::vector<vgsdf_rect> &rects, HostBuffer<uint8_t> &out, uint64_t &out_bytes,
	                   uint64_t &n_segments, uint32_t n_glyphs, std::vector<uint64_t> *pbf_at = nullptr) const;
	// Between the two: the front-end's results as soon as they are on the host, while the raster is still running
	// (vgsdf_outlines_peek).  Returns true when the raster is storing the bitmaps straight into `out` as it stands, so
	// that the caller may write the bytes between them (in-place PBF assembly) right away.
	bool peek_outlines(int lane, std::vector<vgsdf_rect> &rects, uint64_t &out_bytes, uint32_t n_glyphs,
	                   std::vector<uint64_t> *pbf_at = nullptr) const;

	// Device half for a packed batch: fills out[batch.out_bytes()].  Hip: one
	// vgsdf_render_batch call; Dummy: zeros (renderer_dummy.rs).  Throws std::runtime_error.
	void render_batch(const GlyphBatch &batch, uint8_t *out) const;
	// Same for a packed (page-locked) batch; pixels land in batch.out.
	void render_packed(PackedBatch &batch) const;

	// renderer.rs:103 — per-glyph API kept for drop-in parity (a batch of one).
	std::optional<PbfGlyph> render_glyph(const Face &face, uint32_t index) const;

	vgsdf_ctx *ctx() const { return ctx_; }
	std::mutex &ctx_mutex() const { return mu_; }

private:
	Renderer() = default;
	Mode mode_ = Mode::Dummy;
	int device_ = 0;
	vgsdf_ctx *ctx_ = nullptr;
	mutable std::mutex mu_; // vgsdf_ctx is single-threaded
	mutable vgsdf_ctx *ctx2_ = nullptr; // lane 1 of the two-deep pipeline (created on first use)
	mutable std::mutex lane_mu_[2];     // held from submit_outlines to wait_outlines
	vgsdf_ctx *lane_ctx(int lane) const;
	std::vector<std::shared_ptr<Renderer>> peers_; // device lanes 1 .. N-1 of a multi-device renderer
};

} // namespace vg
