// font_manager.hpp — host mirror of the reference's orchestration layer for the render
// path, with the rayon block loop replaced by a GPU batch dispatcher.
//
//   FontFileEntry                       /root/reference/src/font/file_entry.rs:16-56
//   FontWrapper::{add_file,get_blocks}  src/font/wrapper.rs:21-76
//   GlyphBlock::{set_glyph_font,render,filename}   src/font/glyph_block.rs:10-90
//   FontManager::{new,add_font_with_name,render_glyphs}  src/font/manager.rs:18-125
//   name_to_id                          manager.rs:141-147
//   WriterTrait                         src/writer/mod.rs:10-19
//
//   FontManager::{add_path,add_paths}  manager.rs:39-61;  scan  src/commands/recurse.rs:104-133
//   write_index_json / write_families_json   manager.rs:128-138 (index_files.hpp)
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <deque>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "font_name.hpp"
#include "renderer.hpp"
#include "thread_pool.hpp"

namespace vg {

constexpr uint32_t GLYPH_BLOCK_SIZE = 256; // glyph_block.rs:7

// Sink for rendered files (writer/mod.rs:10-19).  write_file may be called from the
// dispatcher thread only (the reference serialises writers behind a Mutex, manager.rs:102).
struct Writer {
	virtual ~Writer() = default;
	virtual void write_directory(const std::string &path) = 0;
	virtual void write_file(const std::string &path, const std::vector<uint8_t> &data) = 0;
	// the same for bytes that live in someone else's buffer (a block assembled in place in the device's output arena):
	// sinks that can take a pointer override it and save the copy
	virtual void write_bytes(const std::string &path, const uint8_t *data, size_t len) { write_file(path, std::vector<uint8_t>(data, data + len)); }
	// ... and for a file whose bytes lie in several places (a block whose glyphs were rendered by several device lanes: its
	// header and the lanes' runs of entries): sinks that stream override it; the default joins the pieces first
	using Piece = std::pair<const uint8_t *, size_t>;
	virtual void write_gather(const std::string &path, const Piece *pieces, size_t n)
	{
		std::vector<uint8_t> all;
		for (size_t i = 0; i < n; i++)
			all.insert(all.end(), pieces[i].first, pieces[i].first + pieces[i].second);
		write_bytes(path, all.data(), all.size());
	}
	virtual void finish() {} // writer/mod.rs:71-77
};

// file_entry.rs: owns the bytes, the parsed face and the cmap coverage
class FontFileEntry {
public:
	static std::unique_ptr<FontFileEntry> create(std::vector<uint8_t> data, std::string *err);
	const Face &face() const { return face_; }
	const std::vector<uint32_t> &codepoints() const { return codepoints_; } // metadata.rs:105-119
	const FontMetadata &metadata() const { return metadata_; }                // metadata.rs:89-103

private:
	std::vector<uint8_t> data_;
	Face face_;
	std::vector<uint32_t> codepoints_;
	FontMetadata metadata_;
};

// glyph_block.rs:10-16 — which file renders each of the 256 code points of a range
struct GlyphBlock {
	uint32_t start_index = 0;
	std::array<const FontFileEntry *, GLYPH_BLOCK_SIZE> glyphs{}; // nullptr = unmapped
	uint32_t count = 0;
	bool all_glyf = true; // every glyph of the block comes from a file with `glyf` outlines (the device can decode them)

	// glyph_block.rs:34-36: first provider wins
	void set_glyph_font(uint8_t char_index, const FontFileEntry *font)
	{
		if (!glyphs[char_index]) {
			glyphs[char_index] = font;
			count++;
			all_glyf = all_glyf && font->face().has_glyf_outlines();
		}
	}
	size_t len() const { return count; }
	bool is_empty() const { return count == 0; }
	const std::string &range() const; // glyph_block.rs:53-59 (the 256 strings of the BMP blocks are built once)
	std::string filename() const { return range() + ".pbf"; } // :85-87
	// "<font>/<range>.pbf" into a caller-owned string (its capacity is reused: one task of a run after the other)
	void path_into(const std::string &font, std::string &out) const
	{
		out.assign(font);
		out.push_back('/');
		out += range();
		out += ".pbf";
	}

	// Host half of GlyphBlock::render (glyph_block.rs:72-77), glyphs in ascending id;
	// [ci0, ci1) restricts it to a sub-range of the block's 256 code points.
	void prepare(TessScratch &scratch, GlyphBatch &batch, uint32_t ci0 = 0, uint32_t ci1 = GLYPH_BLOCK_SIZE) const;
	// glyph_block.rs:69-80 complete (one block through the renderer)
	std::vector<uint8_t> render(const std::string &font_name, const Renderer &renderer) const;
};

// wrapper.rs:15-19
class FontWrapper {
public:
	void add_file(std::unique_ptr<FontFileEntry> f)
	{
		files_.push_back(std::move(f));
		blocks_valid_ = false;
	}
	bool add_paths(const std::vector<std::string> &paths, std::string *err); // wrapper.rs:31-39
	const std::vector<std::unique_ptr<FontFileEntry>> &files() const { return files_; }
	std::vector<GlyphBlock> get_blocks() const; // wrapper.rs:53-76: always 256 blocks
	// the same table, built once per set of files (0.5 MiB per font: rebuilding it for every
	// render_glyphs call was 7 % of an end-to-end run); valid until the next add_file / add_paths
	const std::vector<GlyphBlock> &blocks() const
	{
		if (!blocks_valid_) {
			blocks_ = get_blocks();
			blocks_valid_ = true;
		}
		return blocks_;
	}

private:
	std::vector<std::unique_ptr<FontFileEntry>> files_;
	mutable std::vector<GlyphBlock> blocks_;
	mutable bool blocks_valid_ = false;
};

// Glyph-level shard of one font over `world` ranks: every mapped code point <= 0xFFFF has an
// owner; costs are estimates of w*h*N (bitmap area x segment count) computed from the recorded outline
// commands, identical on every rank, so all ranks derive the same assignment without talking to each other.
struct GlyphShard {
	uint32_t world = 1;
	std::vector<uint8_t> owner; // [65536]: rank that renders the code point, 0xFF = not mapped by the font
	std::vector<double> cost;   // [65536]: estimated cost (0 for unmapped)
	std::vector<double> load;   // [world]: sum of estimated costs per rank
};

struct RenderTimings {
	double tessellate_s = 0, pack_s = 0, device_s = 0, encode_s = 0, write_s = 0, total_s = 0;
	uint64_t blocks = 0, glyphs = 0, rasters = 0, pixels = 0, segments = 0, pbf_bytes = 0;
	uint64_t glyf_groups = 0, glyf_fallbacks = 0; // device front-end: groups decoded from `glyf` arrays / re-recorded on the host
};

class FontManager {
public:
	explicit FontManager(bool parallel) : parallel_(parallel) {} // manager.rs:28-33
	~FontManager();

	// manager.rs:66-75
	bool add_font_with_name(const std::string &name, const std::vector<std::string> &sources, std::string *err);
	bool add_font_data(const std::string &name, std::vector<uint8_t> data, std::string *err);
	// manager.rs:39-53: the file's own name table decides the font id (family + width + weight + style
	// through parse_font_name / generate_name / name_to_id); files with the same id merge, in call order
	bool add_path(const std::string &path, std::string *err);
	bool add_paths(const std::vector<std::string> &paths, std::string *err); // manager.rs:56-61
	// recurse.rs:104-133: a file named *.ttf / *.otf is added; a directory with a fonts.json is read
	// through it ([{name, sources[]}], sources relative to that directory) and NOT descended into; any
	// other directory is walked.  The reference walks in fs::read_dir order (unspecified, so the
	// first-provider-wins merge depends on the file system); here entries are visited in ascending
	// byte order of their names, the canonical order of SURVEY.md §8a.
	bool scan(const std::string &path, std::string *err);
	// manager.rs:128-138
	void write_index_json(Writer &writer) const;
	void write_families_json(Writer &writer) const;

	// manager.rs:81-125.  Host threads tessellate blocks into SoA batches, the GPU renders
	// them, blocks are PBF-encoded and handed to the writer in task order.  Throws
	// std::runtime_error on failure (first error aborts, like try_for_each).
	void render_glyphs(Writer &writer, const Renderer &renderer);
	// Same, restricted to the given block start indices of one font (a rank's shard of the
	// (font, block) task list; blocks are independent: manager.rs:86-97).
	void render_blocks(Writer &writer, const Renderer &renderer, const std::string &font_id,
	                   const std::vector<uint32_t> &block_starts);

	// ---- glyph-level sharding (the reference's unit is the (font, block) task, manager.rs:86-97; 45 non-empty,
	// very unequal blocks do not balance over 8 GPUs, single glyphs do) ----
	// Longest-processing-time-first assignment of the font's glyphs to `world` ranks by estimated cost.
	bool shard_glyphs(const std::string &font_id, uint32_t world, GlyphShard &out, std::string *err) const;
	// From now on render_glyphs / render_blocks / build_batch / record_outlines see only the glyphs rank `rank`
	// owns (every block is still emitted: its PBF then holds this rank's glyphs only, a "partial" that
	// merge_pbf_partials() combines with the other ranks').  world <= 1 switches sharding off.
	void set_glyph_shard(uint32_t rank, uint32_t world);

	// Host stage only: every rasterised glyph of one font, blocks in ascending order (the
	// batch a bench/test keeps resident in HBM).  ids[i] = code point of rasterised glyph i.
	bool build_batch(const std::string &font_id, PackedBatch &out, std::vector<uint32_t> &ids, uint32_t &n_jobs,
	                 std::string *err);

	// Host half of the device front-end for every glyph of one font (ascending code point,
	// first provider wins): what vgsdf_outlines_prepare is fed.
	bool record_outlines(const std::string &font_id, OutlineBatch &out, std::string *err) const;

	const std::map<std::string, FontWrapper> &fonts() const { return parent_ ? parent_->fonts_ : fonts_; }
	const RenderTimings &last_timings() const { return timings_; }
	// {blocks, glyphs, pixels} of the last multi-device render_glyphs, as vgsdf_reduce_counters summed them over the
	// device lanes (all zero after a single-device run: there is nothing to reduce)
	const uint64_t *last_reduced_counters() const { return reduced_; }
	void set_threads(unsigned n) { threads_ = n; }
	void set_batch_blocks(unsigned n)
	{
		batch_blocks_ = n ? n : 1;
		batch_blocks_set_ = true;
	}
	// true: flatten / close / scale / bbox on the GPU (device front-end, HIP renderer only);
	// false: host tessellation.  Both produce identical bytes.
	void set_device_front_end(bool on) { device_front_end_ = on; }
	bool device_front_end() const { return device_front_end_; }
	// true (default): with the device front-end the blocks are assembled IN PLACE — the raster stores every bitmap where
	// the finished PBF has it and the host writes the ~20 bytes around it; false: bitmaps packed back to back, blocks
	// encoded afterwards (one more copy of every bitmap).  Same bytes either way.
	void set_in_place_pbf(bool on) { in_place_pbf_ = on; }
	// glyf fonts through the device front-end: true (default; VG_GLYF_ON_DEVICE=0 in the environment changes it) = the
	// device decodes the glyphs' `glyf` arrays, false = the host's reader records the callbacks.  Same bytes either way.
	void set_glyf_on_device(bool on) { glyf_on_device_ = on; }
	// a renderer with several device lanes: -1 (default) = whole (font, block) tasks per lane unless there are fewer than four
	// non-empty blocks per lane, 0 = always glyph-level shards of every font (+ merge), 1 = always whole tasks.  Same bytes.
	void set_lane_form(int form) { lane_form_ = form; }
	// every glyph of a font id in the form the device's glyf decoder takes (glyf fonts only; tests, inspection)
	bool record_glyf_parts(const std::string &font_id, GlyfPartsBatch &out, std::string *err) const;

private:
	struct Todo {
		const std::string *name;
		const GlyphBlock &block; // lives in its FontWrapper's table
	};
	// One unit of host work: a 64-code-point slice of a task's block, tessellated into the
	// worker's local batch; the ranges say where.
	struct alignas(64) Slice {
		uint32_t task = 0, ci0 = 0, ci1 = 0;
		unsigned worker = 0;
		uint32_t job0 = 0, job1 = 0;       // jobs [job0, job1) of the worker-local batch
		uint32_t raster0 = 0, raster1 = 0; // rasterised glyphs [raster0, raster1) of it
		uint32_t g_raster = 0;             // first raster index in the packed batch
		uint64_t g_seg = 0, g_out = 0;     // first segment / output byte in the packed batch
	};
	struct OSlice { // device front-end: a slice's glyph entries in its worker's OutlineBatch
		uint32_t task = 0;
		unsigned worker = 0;
		uint32_t job0 = 0, job1 = 0;
		uint32_t g_job = 0; // first glyph index in the merged batch
		// what the slice added to its worker's batch, noted by the worker while that batch is hot in its cache (the merge's
		// serial pass then reads this array only): commands / command slots, coordinates / parts, bytes
		uint32_t n_cmd = 0, n_dat = 0, n_byte = 0;
	};
	struct alignas(128) Worker { // own cache lines: the vector headers inside are written per glyph
		TessScratch scratch;
		GlyphBatch local;
		PackedOutlineBatch olocal;
		GlyfPartsBatch plocal;
		char pad[128];
	};
	// One process, N devices (renderer.n_devices() > 1): the glyphs of every font are dealt to the device lanes by
	// estimated cost (shard_glyphs), lane i renders its shard on its own host thread with its own pool, buffers and
	// device contexts, the partial PBFs of a block land side by side in this process's memory and are merged
	// (merge_pbf_partials) — no exchange step; the run counters are summed over the lanes by vgsdf_reduce_counters.
	void render_glyphs_multi(Writer &writer, const Renderer &renderer);
	// a lane of render_glyphs_multi: shares the parent's fonts, renders the glyphs rank `rank` of `world` owns
	FontManager(const FontManager *parent, uint32_t rank, uint32_t world);
	const FontManager *parent_ = nullptr;
public:
	struct CaptureFile { // a file a lane has produced: [at, at + len) of its capture store
		size_t at = 0, len = 0;
	};

private:
	std::vector<uint8_t> capture_store_;     // (a lane's files of the current run: render_glyphs_multi / render_tasks_multi)
	std::vector<CaptureFile> capture_files_;
	std::vector<std::unique_ptr<FontManager>> children_; // lanes, kept between runs (their buffers are grow-only)
	// render_tasks_multi: which lane takes which (font, block) task; kept until a font is added (invalidate_shards)
	struct LanePlan {
		uint32_t world = 0;
		int form = -1;                             // 1: whole tasks only, 2: hybrid (the heaviest blocks split between lanes)
		std::vector<Todo> all;                     // the (font, block) tasks in output order
		std::vector<const std::string *> names;
		static constexpr uint32_t kSplit = 0xFFFFFFFFu;
		std::vector<uint32_t> owner, slot;         // whole task i: its lane and its position among the lane's files;
		                                           // split task: owner = kSplit, slot = index into `splits`
		struct Split {
			uint32_t task, first_part, n_parts;    // parts [first_part, first_part + n_parts) of part_owner / part_slot / part_blocks
		};
		std::vector<Split> splits;
		std::vector<uint32_t> part_owner, part_slot;
		std::deque<GlyphBlock> part_blocks;        // the parts' glyph subsets (a Todo refers to its block)
		std::vector<std::vector<Todo>> lane_tasks;
		std::vector<uint32_t> lane_blocks;         // blocks credited to a lane's counters: its whole tasks + the first parts it holds
		bool accurate = false;                     // weights are the estimated w*h*N per glyph (else: outline sizes)
		double est_max_over_mean = 1.0;            // of the lanes' summed weights
	};
	void build_lane_plan(uint32_t world, int form);
public:
	// Which lane renders which code point of `font_id` on `world` lanes (owner[65536], 0xFF = unmapped) under the plan
	// render_glyphs would use with the present lane form, and how many of the font's blocks are split between lanes.
	// Needs no device: the plan is host arithmetic over the fonts' outlines.
	bool plan_lanes(const std::string &font_id, uint32_t world, std::vector<uint8_t> &owner, uint32_t &n_split_blocks, double *est_max_over_mean, std::string *err);
private:
	LanePlan lane_plan_;
	// shard tables, built once per (font, world, number of files) — on the pool — and shared with the lanes
	struct ShardEntry {
		uint32_t world = 0;
		size_t n_files = 0;
		GlyphShard shard;
	};
	mutable std::map<std::string, ShardEntry> shard_cache_;
	mutable std::mutex shard_mu_;
	const GlyphShard &cached_shard(const std::string &font_id, const FontWrapper &font, uint32_t world) const;
	void invalidate_shards();
	uint64_t reduced_[3] = {0, 0, 0};
	void run_tasks(std::vector<Todo> &tasks, Writer &writer, const Renderer &renderer);
	void render_tasks_multi(Writer &writer, const Renderer &renderer, int form); // N device lanes: whole (font, block) tasks, the heaviest split (form 2)
	void run_tasks_device_front_end(std::vector<Todo> &tasks, Writer &writer, const Renderer &renderer);
	// tessellate tasks [t0, t1) on the pool and pack them (task order, ascending id) into `out`
	void tessellate_and_pack(const std::vector<Todo> &tasks, size_t t0, size_t t1, std::vector<Slice> &slices,
	                         PackedBatch &out);
	ThreadPool &pool();
	unsigned worker_count() const;
	// the block table tasks are built from: the font's own, or its copy filtered to this rank's glyphs
	const std::vector<GlyphBlock> &task_blocks(const std::string &font_id, const FontWrapper &font) const;
	uint32_t shard_rank_ = 0, shard_world_ = 1;
	mutable std::map<std::string, std::vector<GlyphBlock>> shard_blocks_; // per font id, for (shard_rank_, shard_world_)
	std::unique_ptr<ThreadPool> pool_;
	std::vector<Worker> workers_;
	PackedBatch packed_;
	// device front-end: the buffers of one group of tasks (<= batch_blocks_ blocks, normally one font)
	struct FeGroup {
		size_t g0 = 0, g1 = 0;
		std::vector<OSlice> slices;
		std::vector<uint32_t> slice_ci, slice_cmd, slice_dat;
		std::vector<uint32_t> slice_part, slice_byte; // glyf form: first part / first byte of every slice in the merged batch
		MergedOutlines m;
		std::vector<vgsdf_rect> rects;
		std::vector<uint64_t> pbf_at;       // in-place assembly: position of every job's bitmap in `out`
		std::vector<uint32_t> task_g0;      // first job of every task of the group (+ one past the last)
		bool in_place = false;
		struct Piece { // a finished block file: inside `out` (assembled in place) or inside `small`
			const uint8_t *p = nullptr;
			size_t n = 0;
		};
		std::vector<Piece> piece;
		std::vector<uint8_t> small; // blocks without a glyph of the group: name + range only
		std::vector<uint32_t> busy; // the other blocks (indices into the group's tasks): assembled around their bitmaps
		uint64_t n_raster = 0, n_pixels = 0;
		HostBuffer<uint8_t> out{true};
		uint64_t out_bytes = 0, n_segs = 0;
		uint32_t n_jobs = 0;
	};
	FeGroup fe_group_[2]; // two groups in flight: one on the GPU, one being recorded / encoded
	void fe_record(const std::vector<Todo> &tasks, FeGroup &G, bool allow_glyf = true);
	bool fe_record_glyf(const std::vector<Todo> &tasks, FeGroup &G); // false: not a batch for the device's decoder (fan-out past 32-bit offsets)
	// fonts (by id) one of whose groups the device's glyf decoder refused (VGSDF_E_GLYF) or whose parts passed the batch bounds:
	// later groups and runs record them with the host's reader at once instead of paying the double path again; cleared
	// with the shard tables when a font is added
	std::set<const std::string *> glyf_refused_;
	void fe_make_slices(const std::vector<Todo> &tasks, FeGroup &G, uint32_t per_slice);
	void fe_layout_common(const std::vector<Todo> &tasks, FeGroup &G); // task_g0, pbf_pre of the merged batch
	void fe_encode_write(const std::vector<Todo> &tasks, FeGroup &G, Writer &writer);
	void fe_prepare_pieces(const std::vector<Todo> &tasks, FeGroup &G);
	void fe_assemble(const std::vector<Todo> &tasks, FeGroup &G);
	void fe_write_pieces(const std::vector<Todo> &tasks, FeGroup &G, Writer &writer);
	static bool glyf_on_device_default()
	{
		const char *e = std::getenv("VG_GLYF_ON_DEVICE");
		return !(e && e[0] == '0');
	}
	bool in_place_pbf_ = true;
	int lane_form_ = -1;
	bool glyf_on_device_ = glyf_on_device_default(); // glyf fonts: the device decodes the glyphs' arrays (VG_GLYF_ON_DEVICE=0 / set_glyf_on_device(false): the host does)
	bool device_front_end_ = true; // HIP renderer: flatten on the GPU unless switched off
	std::map<std::string, FontWrapper> fonts_; // reference: HashMap (arbitrary order); sorted here
	bool parallel_;
	unsigned threads_ = 0;       // 0 = hardware_concurrency
	unsigned batch_blocks_ = 256; // blocks per GPU submission (256 = one whole font)
	bool batch_blocks_set_ = false; // false: the device front-end groups by glyph count instead (kFeGlyphBudget)
	RenderTimings timings_;
};

std::string name_to_id(const std::string &name); // manager.rs:141-147

// Combines partial PBFs of ONE block (same font name and range; each holds a disjoint subset of the block's
// glyphs, e.g. one per rank of a glyph-level shard) into the block's PBF: glyph messages are taken as they
// are and written in ascending id, so the result equals the PBF a single process encodes.  Throws
// std::runtime_error on malformed input or when names / ranges differ.
std::vector<uint8_t> merge_pbf_partials(const std::vector<std::pair<const uint8_t *, size_t>> &parts);
// the same for parts that hold consecutive runs of the block's code points, in order: header + the parts' entries as they are
std::vector<uint8_t> concat_pbf_partials(const std::vector<std::pair<const uint8_t *, size_t>> &parts);
// the same without the copy: `head` receives the file's own bytes (length prefix, name and range), `pieces` head + the parts'
// entry regions where they lie — for Writer::write_gather.  false: the parts are not in that form (use merge_pbf_partials)
bool plan_pbf_concat(const std::vector<std::pair<const uint8_t *, size_t>> &parts, std::vector<uint8_t> &head,
                     std::vector<std::pair<const uint8_t *, size_t>> &pieces);

} // namespace vg
