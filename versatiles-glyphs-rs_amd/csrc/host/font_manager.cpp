#include "font_manager.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <thread>

#include "index_files.hpp"

namespace vg {

namespace {

double now_s()
{
	using namespace std::chrono;
	return duration<double>(steady_clock::now().time_since_epoch()).count();
}

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string *err)
{
	std::ifstream f(path, std::ios::binary | std::ios::ate);
	if (!f) {
		if (err)
			*err = "reading font file \"" + path + "\""; // wrapper.rs:35 context string
		return false;
	}
	const std::streamsize n = f.tellg();
	f.seekg(0);
	out.resize((size_t)n);
	if (n && !f.read((char *)out.data(), n)) {
		if (err)
			*err = "reading font file \"" + path + "\"";
		return false;
	}
	return true;
}

} // namespace

// manager.rs:141-147: lower-case, runs of [-_\s] -> one separator, trim, ' ' -> '_'
std::string name_to_id(const std::string &name)
{
	std::string out;
	bool pending_sep = false;
	for (unsigned char c : name) {
		const bool sep = c == '-' || c == '_' || c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
		if (sep) {
			pending_sep = !out.empty();
			continue;
		}
		if (pending_sep) {
			out.push_back('_');
			pending_sep = false;
		}
		out.push_back((c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c);
	}
	return out;
}

std::unique_ptr<FontFileEntry> FontFileEntry::create(std::vector<uint8_t> data, std::string *err)
{
	std::unique_ptr<FontFileEntry> e(new FontFileEntry());
	e->data_ = std::move(data);
	auto face = Face::parse(e->data_.data(), e->data_.size());
	if (!face) {
		if (err)
			*err = "font parse failed";
		return nullptr;
	}
	e->face_ = *face;
	if (!e->face_.has_cmap()) { // metadata.rs:104-107
		if (err)
			*err = "Font has no cmap table";
		return nullptr;
	}
	// The reference renders `glyf`, `CFF ` and `CFF2` outlines through ttf-parser, and so does this reader.  A font
	// whose only outline table it cannot open (a malformed `CFF ` / `CFF2`: the crate drops such a table and renders
	// empty glyphs) is refused loudly instead of writing PBFs whose glyphs are all empty.  (A font with no outline
	// table at all renders empty glyphs in the reference too: outline_glyph -> None.)
	if (e->face_.has_unsupported_outlines()) {
		if (err)
			*err = "the font's CFF / CFF2 table cannot be read and it has no glyf outlines";
		return nullptr;
	}
	e->codepoints_ = e->face_.unicode_codepoints();
	{
		// metadata.rs:92-103: HashMap::from_iter over the name records — a later record of the same id
		// replaces an earlier one, and a record ttf-parser cannot decode counts as an empty string
		std::map<uint16_t, std::string> names;
		for (auto &kv : e->face_.names())
			names[kv.first] = std::move(kv.second);
		e->metadata_.name = names.count(1) ? names[1] : std::string(); // name_id::FAMILY
		const ParsedFontName pn = parse_font_name(e->metadata_.name, names.count(6) ? names[6] : std::string()); // POST_SCRIPT_NAME
		e->metadata_.family = pn.family;
		e->metadata_.style = pn.style;
		e->metadata_.weight = pn.weight;
		e->metadata_.width = pn.width;
	}
	return e;
}

const std::string &GlyphBlock::range() const
{
	// "{start}-{start + 255}" (glyph_block.rs:53-59); start is a multiple of 256 below 65536 (wrapper.rs:55-60)
	static const std::vector<std::string> table = [] {
		std::vector<std::string> t;
		for (uint32_t s = 0; s < 0x10000; s += GLYPH_BLOCK_SIZE)
			t.push_back(std::to_string(s) + "-" + std::to_string(s + GLYPH_BLOCK_SIZE - 1));
		return t;
	}();
	static thread_local std::string other;
	if (start_index % GLYPH_BLOCK_SIZE == 0 && start_index < 0x10000)
		return table[start_index / GLYPH_BLOCK_SIZE];
	other = std::to_string(start_index) + "-" + std::to_string(start_index + GLYPH_BLOCK_SIZE - 1);
	return other;
}

void GlyphBlock::prepare(TessScratch &scratch, GlyphBatch &batch, uint32_t ci0, uint32_t ci1) const
{
	for (uint32_t ci = ci0; ci < ci1 && ci < GLYPH_BLOCK_SIZE; ci++)
		if (const FontFileEntry *f = glyphs[ci])
			Renderer::prepare(f->face(), start_index + ci, scratch, batch);
}

std::vector<uint8_t> GlyphBlock::render(const std::string &font_name, const Renderer &renderer) const
{
	TessScratch scratch;
	GlyphBatch batch;
	prepare(scratch, batch);
	std::vector<uint8_t> pixels((size_t)batch.out_bytes());
	renderer.render_batch(batch, pixels.data());
	std::vector<PbfGlyphRef> refs;
	refs.reserve(batch.jobs.size());
	size_t r = 0;
	for (const GlyphJob &j : batch.jobs) {
		const uint8_t *bm = j.has_raster ? pixels.data() + batch.out_off[r++] : nullptr;
		refs.push_back(j.to_pbf(bm));
	}
	return PbfGlyphs::encode(font_name, range(), std::move(refs));
}

bool FontWrapper::add_paths(const std::vector<std::string> &paths, std::string *err)
{
	for (const std::string &p : paths) {
		std::vector<uint8_t> data;
		if (!read_file(p, data, err))
			return false;
		auto e = FontFileEntry::create(std::move(data), err);
		if (!e)
			return false;
		files_.push_back(std::move(e));
		blocks_valid_ = false;
	}
	return true;
}

std::vector<GlyphBlock> FontWrapper::get_blocks() const
{
	constexpr uint32_t kBmpBlocks = 0x10000 / GLYPH_BLOCK_SIZE; // wrapper.rs:55
	std::vector<GlyphBlock> blocks(kBmpBlocks);
	for (uint32_t i = 0; i < kBmpBlocks; i++)
		blocks[i].start_index = i * GLYPH_BLOCK_SIZE;
	for (const auto &file : files_)
		for (uint32_t cp : file->codepoints()) {
			if (cp > 0xFFFF) // wrapper.rs:66-68
				continue;
			blocks[cp / GLYPH_BLOCK_SIZE].set_glyph_font((uint8_t)(cp % GLYPH_BLOCK_SIZE), file.get());
		}
	return blocks;
}

FontManager::~FontManager() = default;

FontManager::FontManager(const FontManager *parent, uint32_t rank, uint32_t world)
    : parent_(parent), parallel_(parent->parallel_)
{
	shard_rank_ = rank;
	shard_world_ = world;
	device_front_end_ = parent->device_front_end_;
	batch_blocks_ = parent->batch_blocks_;
	batch_blocks_set_ = parent->batch_blocks_set_;
}

// a font changed: shard tables, the ranks' filtered block tables and the lanes that hold them are stale
void FontManager::invalidate_shards()
{
	std::lock_guard<std::mutex> lock(shard_mu_);
	shard_cache_.clear();
	shard_blocks_.clear();
	children_.clear();
	lane_plan_ = LanePlan{};
	glyf_refused_.clear();
}

bool FontManager::add_font_with_name(const std::string &name, const std::vector<std::string> &sources, std::string *err)
{
	invalidate_shards();
	return fonts_[name_to_id(name)].add_paths(sources, err);
}

bool FontManager::add_font_data(const std::string &name, std::vector<uint8_t> data, std::string *err)
{
	auto e = FontFileEntry::create(std::move(data), err);
	if (!e)
		return false;
	invalidate_shards();
	fonts_[name_to_id(name)].add_file(std::move(e));
	return true;
}

bool FontManager::add_path(const std::string &path, std::string *err)
{
	std::vector<uint8_t> data;
	if (!read_file(path, data, err))
		return false;
	auto e = FontFileEntry::create(std::move(data), err);
	if (!e)
		return false;
	invalidate_shards();
	fonts_[name_to_id(e->metadata().generate_name())].add_file(std::move(e));
	return true;
}

bool FontManager::add_paths(const std::vector<std::string> &paths, std::string *err)
{
	for (const std::string &p : paths)
		if (!add_path(p, err))
			return false;
	return true;
}

bool FontManager::scan(const std::string &path, std::string *err)
{
	struct stat st;
	if (::stat(path.c_str(), &st) != 0) // is_file() / is_dir() are both false: nothing to do (recurse.rs:105-111)
		return true;
	if (S_ISREG(st.st_mode)) {
		// Path::extension(): what follows the last '.' of the file name (none for ".ttf" itself)
		const size_t slash = path.find_last_of('/');
		const std::string fname = slash == std::string::npos ? path : path.substr(slash + 1);
		const size_t dot = fname.find_last_of('.');
		const std::string ext = (dot == std::string::npos || dot == 0) ? std::string() : fname.substr(dot + 1);
		if (ext == "ttf" || ext == "otf")
			return add_path(path, err);
		return true;
	}
	if (!S_ISDIR(st.st_mode))
		return true;
	const std::string base = (!path.empty() && path.back() == '/') ? path : path + "/";
	const std::string cfg = base + "fonts.json";
	if (::stat(cfg.c_str(), &st) == 0) {
		std::vector<uint8_t> data;
		if (!read_file(cfg, data, err)) {
			if (err)
				*err = "Failed to read \"" + cfg + "\": " + *err;
			return false;
		}
		std::vector<FontConfig> configs;
		if (!parse_fonts_json(std::string(data.begin(), data.end()), configs, err))
			return false;
		for (const FontConfig &c : configs) {
			std::vector<std::string> sources;
			for (const std::string &src : c.sources)
				sources.push_back((!src.empty() && src[0] == '/') ? src : base + src); // Path::join
			if (!add_font_with_name(c.name, sources, err))
				return false;
		}
		return true;
	}
	DIR *d = ::opendir(path.c_str());
	if (!d) {
		if (err)
			*err = "reading directory \"" + path + "\": " + std::strerror(errno);
		return false;
	}
	std::vector<std::string> names;
	while (struct dirent *ent = ::readdir(d)) {
		const std::string n = ent->d_name;
		if (n != "." && n != "..")
			names.push_back(n);
	}
	::closedir(d);
	std::sort(names.begin(), names.end());
	for (const std::string &n : names)
		if (!scan(base + n, err))
			return false;
	return true;
}

void FontManager::write_index_json(Writer &writer) const { writer.write_file("index.json", build_index_json(*this)); }

void FontManager::write_families_json(Writer &writer) const
{
	writer.write_file("font_families.json", build_font_families_json(*this));
}

// CPUs this process may actually run on at once: the cgroup's CPU quota when there is one (containers: cpu.max =
// "quota period"), else the scheduler affinity / hardware_concurrency.
static unsigned cpu_budget()
{
	unsigned n = std::thread::hardware_concurrency();
	if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[32] = {0};
		long long period = 0;
		if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
			const long long quota = std::atoll(q);
			if (quota > 0)
				n = (unsigned)std::min<long long>(n ? n : 1u << 20, (quota + period - 1) / period);
		}
		std::fclose(f);
	}
	return n ? n : 1;
}

unsigned FontManager::worker_count() const
{
	if (!parallel_)
		return 1;
	if (threads_)
		return threads_;
	// VG_THREADS or set_threads() override.  Default: 1.5 x the CPUs the process may use, at most 64 (beyond that the
	// fork / join of a phase costs more than the phase; rounds 1-2 capped the pool at 16).  The phases are a few hundred
	// microseconds long and some threads beyond one per CPU hide the wake-up of the others — but a container's CPU quota
	// counts CPU TIME per 100 ms period, spinning included, and a run longer than one period is throttled when pool +
	// spinners exceed it (tools/sustained_e2e.py on a 16-CPU quota, 2 s runs, glyphs/s of the 21 fixture fonts / of Noto
	// Sans Regular alone: 32 threads that all spin 100 us after a fork 5.0-5.6 M / 3.2-3.6 M with 29 of 30 periods
	// throttled — the setting round 3 first chose from runs of a few milliseconds, where it is the fastest —; 32 threads,
	// no spinning 7.1-8.1 / 4.4-5.2; 16 threads, all spinning 5.9-7.1 / 5.2-5.8; 24 threads of which at most 2 spin
	// 8.4 / 6.0, nothing throttled).  The spinner limit is the pool's (thread_pool.hpp).
	if (const char *e = std::getenv("VG_THREADS"))
		if (int v = std::atoi(e); v > 0)
			return (unsigned)v;
	static const unsigned budget = cpu_budget();
	return std::max(1u, std::min(budget + budget / 2, 64u));
}

ThreadPool &FontManager::pool()
{
	const unsigned want = worker_count();
	if (!pool_ || pool_->size() != want) {
		pool_.reset(new ThreadPool(want));
		workers_.clear();
		workers_.resize(want);
	}
	return *pool_;
}

void FontManager::tessellate_and_pack(const std::vector<Todo> &tasks, size_t t0, size_t t1,
                                      std::vector<Slice> &slices, PackedBatch &out)
{
	constexpr uint32_t kSlice = 64; // code points per unit of host work
	ThreadPool &tp = pool();
	slices.clear();
	for (size_t t = t0; t < t1; t++) {
		if (tasks[t].block.is_empty())
			continue;
		for (uint32_t c = 0; c < GLYPH_BLOCK_SIZE; c += kSlice) {
			Slice s;
			s.task = (uint32_t)t;
			s.ci0 = c;
			s.ci1 = c + kSlice;
			slices.push_back(s);
		}
	}
	for (Worker &w : workers_)
		w.local.clear();
	const double t_begin = now_s();

	// T: tessellate every slice into its worker's local batch (capacity is retained run to run)
	tp.run(slices.size(), [&](size_t i, unsigned wid) {
		Slice &s = slices[i];
		Worker &w = workers_[wid];
		s.worker = wid;
		s.job0 = (uint32_t)w.local.jobs.size();
		s.raster0 = (uint32_t)w.local.n_raster();
		tasks[s.task].block.prepare(w.scratch, w.local, s.ci0, s.ci1);
		s.job1 = (uint32_t)w.local.jobs.size();
		s.raster1 = (uint32_t)w.local.n_raster();
	});

	const double t_tess_done = now_s();
	// P: positions in the packed batch, in task order (deterministic: ascending id per font)
	uint32_t rasters = 0;
	uint64_t segs = 0, pixels = 0;
	for (Slice &s : slices) {
		const GlyphBatch &l = workers_[s.worker].local;
		s.g_raster = rasters;
		s.g_seg = segs;
		s.g_out = pixels;
		rasters += s.raster1 - s.raster0;
		segs += l.seg_off[s.raster1] - l.seg_off[s.raster0];
		pixels += l.out_off[s.raster1] - l.out_off[s.raster0];
	}
	if (segs > 0xFFFFFFFFull)
		throw std::runtime_error("batch exceeds 2^32 segments; lower set_batch_blocks()");
	out.reserve(rasters, segs, pixels);
	out.n_raster = rasters;
	out.n_seg = segs;
	out.out_bytes = pixels;

	// C: copy the slices into the page-locked SoA arrays
	tp.run(slices.size(), [&](size_t i, unsigned) {
		const Slice &s = slices[i];
		const GlyphBatch &l = workers_[s.worker].local;
		const uint32_t ls0 = l.seg_off[s.raster0];
		const size_t n = l.seg_off[s.raster1] - ls0;
		if (n) {
			std::memcpy(out.sx.data() + s.g_seg, l.sx.data() + ls0, n * sizeof(double));
			std::memcpy(out.sy.data() + s.g_seg, l.sy.data() + ls0, n * sizeof(double));
			std::memcpy(out.ex.data() + s.g_seg, l.ex.data() + ls0, n * sizeof(double));
			std::memcpy(out.ey.data() + s.g_seg, l.ey.data() + ls0, n * sizeof(double));
		}
		const uint64_t lo0 = l.out_off[s.raster0];
		for (uint32_t r = s.raster0; r < s.raster1; r++) {
			const uint32_t g = s.g_raster + (r - s.raster0);
			out.x0[g] = l.x0[r];
			out.y0[g] = l.y0[r];
			out.w[g] = l.w[r];
			out.h[g] = l.h[r];
			out.seg_off[g + 1] = (uint32_t)(s.g_seg + (l.seg_off[r + 1] - ls0));
			out.out_off[g + 1] = s.g_out + (l.out_off[r + 1] - lo0);
		}
	});
	timings_.tessellate_s += t_tess_done - t_begin;
	timings_.pack_s += now_s() - t_tess_done;
}

bool FontManager::build_batch(const std::string &font_id, PackedBatch &out, std::vector<uint32_t> &ids,
                              uint32_t &n_jobs, std::string *err)
{
	auto it = fonts().find(font_id);
	if (it == fonts().end()) {
		if (err)
			*err = "unknown font id " + font_id;
		return false;
	}
	std::vector<Todo> tasks;
	for (const GlyphBlock &b : task_blocks(it->first, it->second))
		tasks.push_back(Todo{&it->first, b});
	std::vector<Slice> slices;
	tessellate_and_pack(tasks, 0, tasks.size(), slices, out);
	ids.assign(out.n_raster, 0);
	n_jobs = 0;
	for (const Slice &s : slices) {
		const GlyphBatch &l = workers_[s.worker].local;
		n_jobs += s.job1 - s.job0;
		for (uint32_t r = s.raster0; r < s.raster1; r++)
			ids[s.g_raster + (r - s.raster0)] = l.jobs[l.raster_job[r]].id;
	}
	return true;
}

bool FontManager::record_outlines(const std::string &font_id, OutlineBatch &out, std::string *err) const
{
	auto it = fonts().find(font_id);
	if (it == fonts().end()) {
		if (err)
			*err = "unknown font id " + font_id;
		return false;
	}
	out.clear();
	for (const GlyphBlock &b : task_blocks(it->first, it->second))
		for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
			if (const FontFileEntry *f = b.glyphs[ci])
				Renderer::record(f->face(), b.start_index + ci, out);
	return true;
}

bool FontManager::record_glyf_parts(const std::string &font_id, GlyfPartsBatch &out, std::string *err) const
{
	auto it = fonts().find(font_id);
	if (it == fonts().end()) {
		if (err)
			*err = "unknown font id " + font_id;
		return false;
	}
	out.clear();
	for (const GlyphBlock &b : task_blocks(it->first, it->second)) {
		if (!b.all_glyf) {
			if (err)
				*err = "font " + font_id + " has glyphs without glyf outlines (CFF / CFF2): the device decodes glyf entries only";
			return false;
		}
		for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
			if (const FontFileEntry *f = b.glyphs[ci])
				Renderer::record_parts(f->face(), b.start_index + ci, out);
		if (out.overflow) {
			if (err)
				*err = "font " + font_id + ": the glyf parts of a glyph pass the batch bounds (composite fan-out): not a batch for the device's decoder";
			return false;
		}
	}
	return true;
}

// ---- glyph-level sharding ---------------------------------------------------------------
namespace {

// Estimated raster cost of one recorded glyph: (bitmap area) x (segment count), the w*h*N of SURVEY.md §8e.
// Segments: a quadratic Bezier is halved until |s + e - 2c|^2 <= 0.01 font units^2 (ring.rs:129-131); both
// halves of a quadratic have exactly 1/4 of the parent's deviation, so the depth is uniform and the point
// count is a power of two that follows from the control polygon alone.  Cubics (no fixture font has any) are
// estimated the same way from their larger control deviation.  Area: bounding box of the control points,
// scaled, plus the 3 px buffer on every side (renderer.rs:64-91).
double estimate_cost(const vgsdf_outline_cmd *c, uint32_t n, double scale)
{
	if (n == 0)
		return 1.0;
	double segs = 0, minx = 1e300, miny = 1e300, maxx = -1e300, maxy = -1e300;
	double lx = 0, ly = 0;
	auto grow = [&](double x, double y) {
		minx = std::min(minx, x), maxx = std::max(maxx, x);
		miny = std::min(miny, y), maxy = std::max(maxy, y);
	};
	auto pieces = [](double dev2) {
		double k = 1;
		while (dev2 > 0.01 && k < 65536) {
			dev2 /= 16;
			k *= 2;
		}
		return k;
	};
	for (uint32_t i = 0; i < n; i++) {
		const vgsdf_outline_cmd &q = c[i];
		switch (q.kind) {
		case 0: // move_to
			break;
		case 1: // line_to
			segs += 1;
			break;
		case 2: { // quad_to
			const double dx = lx + q.x - 2.0 * q.x1, dy = ly + q.y - 2.0 * q.y1;
			segs += pieces(dx * dx + dy * dy);
			grow(q.x1, q.y1);
			break;
		}
		case 3: { // curve_to
			const double dx = (double)q.x2 + q.x1 - (lx + q.x), dy = (double)q.y2 + q.y1 - (ly + q.y);
			segs += pieces(dx * dx + dy * dy);
			grow(q.x1, q.y1);
			grow(q.x2, q.y2);
			break;
		}
		default: // close: the closing segment
			segs += 1;
			continue;
		}
		grow(q.x, q.y);
		lx = q.x, ly = q.y;
	}
	if (!(maxx >= minx))
		return 1.0;
	const double w = std::ceil((maxx - minx) * scale) + 2 * BUFFER + 1, h = std::ceil((maxy - miny) * scale) + 2 * BUFFER + 1;
	return std::max(1.0, segs) * w * h;
}

} // namespace

namespace {

// longest processing time first; ties by code point, so every rank computes the same assignment
void assign_lpt(GlyphShard &out)
{
	std::vector<std::pair<double, uint32_t>> order; // (cost, code point)
	for (uint32_t cp = 0; cp < 0x10000; cp++)
		if (out.cost[cp] > 0.0)
			order.emplace_back(out.cost[cp], cp);
	std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
	out.owner.assign(0x10000, 0xFF);
	out.load.assign(out.world, 0.0);
	for (const auto &e : order) {
		uint32_t best = 0;
		for (uint32_t r = 1; r < out.world; r++)
			if (out.load[r] < out.load[best])
				best = r;
		out.owner[e.second] = (uint8_t)best;
		out.load[best] += e.first;
	}
}

// estimated costs of the mapped glyphs of one block (cheap: table walks, no flattening), first provider wins
void block_costs(const GlyphBlock &b, OutlineBatch &rec, std::vector<double> &cost)
{
	for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
		if (const FontFileEntry *f = b.glyphs[ci]) {
			const uint32_t cp = b.start_index + ci;
			rec.clear();
			double c = 1.0;
			if (Renderer::record(f->face(), cp, rec))
				c = estimate_cost(rec.cmds.data(), (uint32_t)rec.cmds.size(), rec.scale[0]);
			cost[cp] = c; // >= 1 for every mapped code point
		}
}

} // namespace

bool FontManager::shard_glyphs(const std::string &font_id, uint32_t world, GlyphShard &out, std::string *err) const
{
	auto it = fonts().find(font_id);
	if (it == fonts().end() || world == 0 || world > 254) {
		if (err)
			*err = it == fonts().end() ? "unknown font id " + font_id : "shard_glyphs: world must be 1..254";
		return false;
	}
	out.world = world;
	out.cost.assign(0x10000, 0.0);
	OutlineBatch rec;
	for (const GlyphBlock &b : it->second.blocks())
		block_costs(b, rec, out.cost);
	assign_lpt(out);
	return true;
}

// The same table, kept per (font, world, number of files) and built with the blocks spread over the calling thread's
// pool when there is one (a lane asks its parent; the parent fills the cache before it starts the lanes).
const GlyphShard &FontManager::cached_shard(const std::string &font_id, const FontWrapper &font, uint32_t world) const
{
	if (parent_)
		return parent_->cached_shard(font_id, font, world);
	std::lock_guard<std::mutex> lock(shard_mu_);
	ShardEntry &e = shard_cache_[font_id];
	if (e.world == world && e.n_files == font.files().size() && !e.shard.owner.empty())
		return e.shard;
	if (world == 0 || world > 254)
		throw std::runtime_error("glyph shard: world must be 1..254");
	e.world = world;
	e.n_files = font.files().size();
	e.shard.world = world;
	e.shard.cost.assign(0x10000, 0.0);
	const std::vector<GlyphBlock> &blocks = font.blocks();
	if (pool_ && pool_->size() > 1) {
		std::vector<OutlineBatch> recs(pool_->size());
		pool_->run(blocks.size(), [&](size_t i, unsigned wid) { block_costs(blocks[i], recs[wid], e.shard.cost); });
	} else {
		OutlineBatch rec;
		for (const GlyphBlock &b : blocks)
			block_costs(b, rec, e.shard.cost);
	}
	assign_lpt(e.shard);
	return e.shard;
}

void FontManager::set_glyph_shard(uint32_t rank, uint32_t world)
{
	if (world > 254 || (world > 1 && rank >= world))
		throw std::runtime_error("set_glyph_shard: need rank < world <= 254");
	shard_rank_ = rank;
	shard_world_ = world ? world : 1;
	shard_blocks_.clear();
}

const std::vector<GlyphBlock> &FontManager::task_blocks(const std::string &font_id, const FontWrapper &font) const
{
	if (shard_world_ <= 1)
		return font.blocks();
	auto it = shard_blocks_.find(font_id);
	if (it != shard_blocks_.end())
		return it->second; // (every add_* clears this table: invalidate_shards)
	const GlyphShard &sh = cached_shard(font_id, font, shard_world_);
	std::vector<GlyphBlock> blocks = font.blocks(); // copy, then drop what other ranks own
	for (GlyphBlock &b : blocks)
		for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
			if (b.glyphs[ci] && sh.owner[b.start_index + ci] != shard_rank_) {
				b.glyphs[ci] = nullptr;
				b.count--;
			}
	return shard_blocks_[font_id] = std::move(blocks);
}

std::vector<uint8_t> merge_pbf_partials(const std::vector<std::pair<const uint8_t *, size_t>> &parts)
{
	struct G {
		uint32_t id;
		const uint8_t *p; // the glyph message's payload
		size_t n;
	};
	auto varint = [](const uint8_t *&p, const uint8_t *end, uint64_t &v) {
		v = 0;
		for (int sh = 0; p < end && sh < 64; sh += 7) {
			const uint8_t b = *p++;
			v |= (uint64_t)(b & 0x7F) << sh;
			if (!(b & 0x80))
				return true;
		}
		return false;
	};
	std::string name, range;
	bool have = false;
	std::vector<G> glyphs;
	for (const auto &part : parts) {
		const uint8_t *p = part.first, *end = p + part.second;
		uint64_t len;
		if (p == end || *p++ != 0x0A || !varint(p, end, len) || len != (uint64_t)(end - p))
			throw std::runtime_error("merge_pbf_partials: not a glyphs PBF with one fontstack");
		std::string nm, rg;
		while (p < end) {
			const uint8_t tag = *p++;
			if (!varint(p, end, len) || len > (uint64_t)(end - p))
				throw std::runtime_error("merge_pbf_partials: truncated field");
			if (tag == 0x0A) {
				nm.assign((const char *)p, (size_t)len);
			} else if (tag == 0x12) {
				rg.assign((const char *)p, (size_t)len);
			} else if (tag == 0x1A) {
				const uint8_t *q = p, *qe = p + len;
				uint64_t id;
				if (q == qe || *q++ != 0x08 || !varint(q, qe, id))
					throw std::runtime_error("merge_pbf_partials: glyph without id");
				glyphs.push_back(G{(uint32_t)id, p, (size_t)len});
			} else {
				throw std::runtime_error("merge_pbf_partials: unexpected field");
			}
			p += len;
		}
		if (have && (nm != name || rg != range))
			throw std::runtime_error("merge_pbf_partials: parts of different blocks (" + name + "/" + range + " vs " + nm + "/" + rg + ")");
		name = nm, range = rg, have = true;
	}
	std::stable_sort(glyphs.begin(), glyphs.end(), [](const G &a, const G &b) { return a.id < b.id; });
	auto vsize = [](uint64_t v) {
		size_t n = 1;
		for (; v >= 0x80; v >>= 7)
			n++;
		return n;
	};
	auto put = [](std::vector<uint8_t> &o, uint64_t v) {
		for (; v >= 0x80; v >>= 7)
			o.push_back((uint8_t)(v | 0x80));
		o.push_back((uint8_t)v);
	};
	size_t stack = 1 + vsize(name.size()) + name.size() + 1 + vsize(range.size()) + range.size();
	for (const G &g : glyphs)
		stack += 1 + vsize(g.n) + g.n;
	std::vector<uint8_t> out;
	out.reserve(1 + vsize(stack) + stack);
	out.push_back(0x0A);
	put(out, stack);
	out.push_back(0x0A);
	put(out, name.size());
	out.insert(out.end(), name.begin(), name.end());
	out.push_back(0x12);
	put(out, range.size());
	out.insert(out.end(), range.begin(), range.end());
	for (const G &g : glyphs) {
		out.push_back(0x1A);
		put(out, g.n);
		out.insert(out.end(), g.p, g.p + g.n);
	}
	return out;
}

// The same for parts that hold CONSECUTIVE runs of a block's code points, in order (the split blocks of the hybrid lane plan):
// the entries of a part are already in ascending id and stay together, so the block's file is its header followed by the
// parts' entry regions as they are — three or four copies instead of a walk over every glyph message.  The first ids of the
// parts must ascend (checked); anything unexpected goes to merge_pbf_partials.
bool plan_pbf_concat(const std::vector<std::pair<const uint8_t *, size_t>> &parts, std::vector<uint8_t> &head,
                     std::vector<std::pair<const uint8_t *, size_t>> &pieces)
{
	auto varint = [](const uint8_t *&p, const uint8_t *end, uint64_t &v) {
		v = 0;
		for (int sh = 0; p < end && sh < 64; sh += 7) {
			const uint8_t b = *p++;
			v |= (uint64_t)(b & 0x7F) << sh;
			if (!(b & 0x80))
				return true;
		}
		return false;
	};
	head.clear();
	pieces.clear();
	const uint8_t *fields = nullptr; // name + range fields of the first part
	size_t fields_n = 0;
	uint64_t last_first_id = 0;
	bool any = false;
	for (const auto &part : parts) {
		const uint8_t *p = part.first, *end = p + part.second;
		uint64_t len;
		if (p == end || *p++ != 0x0A || !varint(p, end, len) || len != (uint64_t)(end - p))
			return false;
		const uint8_t *f0 = p;
		for (int k = 0; k < 2; k++) { // 0x0A name, 0x12 range (fontstack.rs:9-25: in tag order)
			if (p == end || *p++ != (k ? 0x12 : 0x0A) || !varint(p, end, len) || len > (uint64_t)(end - p))
				return false;
			p += len;
		}
		if (!fields) {
			fields = f0;
			fields_n = (size_t)(p - f0);
		} else if ((size_t)(p - f0) != fields_n || std::memcmp(f0, fields, fields_n) != 0) {
			return false; // (parts of different blocks: merge_pbf_partials says so)
		}
		if (p == end)
			continue; // a part without glyphs
		const uint8_t *q = p;
		uint64_t glen, id;
		if (*q++ != 0x1A || !varint(q, end, glen) || q == end || *q++ != 0x08 || !varint(q, end, id) || (any && id <= last_first_id))
			return false;
		last_first_id = id;
		any = true;
		pieces.emplace_back(p, (size_t)(end - p));
	}
	if (!fields)
		return false;
	size_t stack = fields_n;
	for (const auto &r : pieces)
		stack += r.second;
	head.push_back(0x0A);
	for (uint64_t v = stack;; v >>= 7) {
		if (v < 0x80) {
			head.push_back((uint8_t)v);
			break;
		}
		head.push_back((uint8_t)(v | 0x80));
	}
	head.insert(head.end(), fields, fields + fields_n);
	pieces.insert(pieces.begin(), std::make_pair((const uint8_t *)head.data(), head.size()));
	return true;
}

std::vector<uint8_t> concat_pbf_partials(const std::vector<std::pair<const uint8_t *, size_t>> &parts)
{
	std::vector<uint8_t> head, out;
	std::vector<std::pair<const uint8_t *, size_t>> pieces;
	if (!plan_pbf_concat(parts, head, pieces))
		return merge_pbf_partials(parts);
	size_t total = 0;
	for (const auto &pc : pieces)
		total += pc.second;
	out.reserve(total);
	for (const auto &pc : pieces)
		out.insert(out.end(), pc.first, pc.first + pc.second);
	return out;
}

void FontManager::render_glyphs(Writer &writer, const Renderer &renderer)
{
	if (renderer.n_devices() > 1 && renderer.mode() == Renderer::Mode::Hip && !parent_) {
		render_glyphs_multi(writer, renderer);
		return;
	}
	// manager.rs:86-97: one task per (font, block); all 256 blocks per font
	std::vector<Todo> tasks;
	for (const auto &[name, font] : fonts()) {
		writer.write_directory(name + "/");
		for (const GlyphBlock &b : task_blocks(name, font))
			tasks.push_back(Todo{&name, b});
	}
	std::memset(reduced_, 0, sizeof reduced_);
	run_tasks(tasks, writer, renderer);
}

namespace {
// what a lane of render_glyphs_multi writes into: its partial PBFs, in task order
struct CaptureWriter final : Writer {
	// (one store for all files of the lane — a vector per file was 5376 allocations per run over the 21 fixture fonts —, kept
	// by the lane between runs: a fresh store of a few megabytes is mapped and faulted in page by page every time)
	using File = FontManager::CaptureFile;
	std::vector<uint8_t> &store;
	std::vector<File> &files;
	CaptureWriter(std::vector<uint8_t> &s, std::vector<File> &f) : store(s), files(f)
	{
		store.clear();
		files.clear();
	}
	const uint8_t *data(size_t i) const { return store.data() + files[i].at; }
	size_t size(size_t i) const { return files[i].len; }
	void write_directory(const std::string &) override {}
	void write_file(const std::string &path, const std::vector<uint8_t> &d) override { write_bytes(path, d.data(), d.size()); }
	void write_gather(const std::string &, const Piece *pieces, size_t n) override
	{
		const size_t at = store.size();
		for (size_t i = 0; i < n; i++)
			store.insert(store.end(), pieces[i].first, pieces[i].first + pieces[i].second);
		files.push_back(File{at, store.size() - at});
	}
	void write_bytes(const std::string &, const uint8_t *d, size_t len) override
	{
		files.push_back(File{store.size(), len});
		store.insert(store.end(), d, d + len);
	}
};
} // namespace

// The lanes take WHOLE (font, block) tasks — the reference's own unit (manager.rs:86-97) — dealt out longest first; every
// file is rendered, assembled and captured by one lane and nothing is merged.  One font's 20-45 unequal non-empty blocks do
// not balance over 8 devices that way (estimated raster cost per lane up to 1.3 / 1.5 x the mean for Noto Sans Regular / all
// files): the HYBRID plan (form 2, the default) then splits the glyphs of the few heaviest blocks between lanes — those
// blocks' partial PBFs are merged afterwards, everything else stays whole (manager.rs:117-121 has rayon steal whole tasks;
// with 8 devices and one font there is nothing to steal).
void FontManager::build_lane_plan(uint32_t world, int form)
{
	LanePlan plan;
	plan.world = world;
	plan.form = form;
	for (const auto &[name, font] : fonts_) {
		plan.names.push_back(&name);
		for (const GlyphBlock &b : font.blocks())
			plan.all.push_back(Todo{&name, b});
	}
	const size_t n_tasks = plan.all.size();
	// Step 1, cheap weights: a block's glyphs' outline sizes (command slots of their glyf entries + a constant per glyph; 40
	// per glyph of a CFF font) — a stand-in for the raster's w*h*N that correlates 0.87 with it and costs one table walk
	// per glyph on the pool.  Many fonts (hundreds of non-empty tasks) balance on it and need nothing else.
	std::vector<double> weight(n_tasks, 0.0);
	pool().run(n_tasks, [&](size_t i, unsigned) {
		const GlyphBlock &blk = plan.all[i].block;
		if (blk.is_empty())
			return;
		std::vector<GlyfPart> parts;
		std::vector<uint8_t> bytes;
		uint64_t w = 0;
		for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
			if (const FontFileEntry *f = blk.glyphs[ci]) {
				uint32_t slots = 32;
				if (f->face().has_glyf_outlines()) {
					slots = 0;
					parts.clear();
					bytes.clear();
					if (const auto gid = f->face().glyph_index(blk.start_index + ci))
						(void)f->face().glyph_parts(*gid, parts, bytes, slots);
				}
				w += 8 + slots;
			}
		weight[i] = (double)w;
	});
	// an item = a whole task or one part of a split task
	struct Item {
		uint32_t task, part, n_parts;
		double w;
	};
	std::vector<Item> items;
	for (size_t i = 0; i < n_tasks; i++)
		if (!plan.all[i].block.is_empty())
			items.push_back(Item{(uint32_t)i, 0, 1, weight[i]});
	std::vector<uint32_t> item_lane;
	std::vector<double> load;
	auto lpt = [&]() { // longest processing time first (ties: task order, so the plan is deterministic); -> max / mean
		std::vector<uint32_t> order(items.size());
		for (size_t k = 0; k < items.size(); k++)
			order[k] = (uint32_t)k;
		std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return items[x].w > items[y].w; });
		load.assign(world, 0.0);
		item_lane.assign(items.size(), 0);
		for (uint32_t k : order) {
			const uint32_t r = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
			item_lane[k] = r;
			load[r] += items[k].w;
		}
		double sum = 0, mx = 0;
		for (double l : load)
			sum += l, mx = std::max(mx, l);
		return sum > 0 ? mx / (sum / world) : 1.0;
	};
	double ratio = lpt();
	// Step 2, hybrid: few tasks per lane (fewer than 32 non-empty blocks each: one font, or a handful, on several devices).
	// The cheap weights are too coarse for that — they balance Noto Sans' 45 blocks over 8 lanes to 1.02 in their own
	// measure and to 1.57 in true raster cost — and whole blocks too large.  Weights become the estimated raster cost w*h*N of every glyph (from
	// its recorded outline: correlation 0.9996 with the true cost; the shard tables, built on the pool once per font set),
	// and while the fullest lane is more than 4 % over the mean, the heaviest splittable item ON that lane is cut into twice
	// as many parts (contiguous code point ranges of equal estimated cost).  A dozen iterations for one font on 8 lanes.
	std::vector<std::vector<double>> glyph_cost; // per task, per code point of the block (hybrid only)
	std::vector<uint32_t> task_parts(n_tasks, 1);
	constexpr double kGood = 1.04;
	if (form == 2 && world > 1 && items.size() < 32u * world) {
		plan.accurate = true;
		glyph_cost.resize(n_tasks);
		size_t t0 = 0;
		for (const auto &[name, font] : fonts_) {
			const GlyphShard &sh = cached_shard(name, font, world);
			const size_t nb = font.blocks().size();
			for (size_t bi = 0; bi < nb; bi++) {
				const GlyphBlock &blk = plan.all[t0 + bi].block;
				if (blk.is_empty())
					continue;
				std::vector<double> &gc = glyph_cost[t0 + bi];
				gc.assign(GLYPH_BLOCK_SIZE, 0.0);
				double w = 0;
				for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
					if (blk.glyphs[ci]) {
						gc[ci] = std::max(1.0, sh.cost[blk.start_index + ci]);
						w += gc[ci];
					}
				weight[t0 + bi] = w;
			}
			t0 += nb;
		}
		for (Item &it : items)
			it.w = weight[it.task];
		// cost of part p of n of a task: the glyphs whose cumulative cost (up to and including their own) falls into
		// ((p / n) W, ((p + 1) / n) W]
		auto part_of = [&](uint32_t task, double cum, uint32_t n) {
			const double W = weight[task];
			uint32_t p = (uint32_t)std::ceil(cum / W * n) - 1;
			return std::min(p, n - 1);
		};
		auto rebuild_items = [&]() {
			items.clear();
			for (size_t i = 0; i < n_tasks; i++) {
				if (plan.all[i].block.is_empty())
					continue;
				const uint32_t n = task_parts[i];
				if (n == 1) {
					items.push_back(Item{(uint32_t)i, 0, 1, weight[i]});
					continue;
				}
				std::vector<double> pw(n, 0.0);
				double cum = 0;
				for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
					if (glyph_cost[i][ci] > 0) {
						cum += glyph_cost[i][ci];
						pw[part_of((uint32_t)i, cum, n)] += glyph_cost[i][ci];
					}
				for (uint32_t p = 0; p < n; p++)
					if (pw[p] > 0)
						items.push_back(Item{(uint32_t)i, p, n, pw[p]});
			}
		};
		for (int iter = 0; iter < 256; iter++) {
			ratio = lpt();
			if (ratio <= kGood)
				break;
			const uint32_t full = (uint32_t)(std::max_element(load.begin(), load.end()) - load.begin());
			// the heaviest item of the fullest lane whose task can still be cut finer
			int best = -1;
			for (size_t k = 0; k < items.size(); k++)
				if (item_lane[k] == full && 2 * task_parts[items[k].task] <= plan.all[items[k].task].block.len() &&
				    (best < 0 || items[k].w > items[(size_t)best].w))
					best = (int)k;
			if (best < 0)
				break;
			task_parts[items[(size_t)best].task] *= 2;
			rebuild_items();
		}
		ratio = lpt();
		// the parts' glyph subsets
		for (size_t i = 0; i < n_tasks; i++) {
			const uint32_t n = task_parts[i];
			if (n == 1)
				continue;
			const GlyphBlock &blk = plan.all[i].block;
			std::vector<GlyphBlock> sub(n);
			for (GlyphBlock &sb : sub)
				sb.start_index = blk.start_index;
			double cum = 0;
			for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
				if (blk.glyphs[ci]) {
					cum += glyph_cost[i][ci];
					sub[part_of((uint32_t)i, cum, n)].set_glyph_font((uint8_t)ci, blk.glyphs[ci]);
				}
			LanePlan::Split sp{(uint32_t)i, (uint32_t)plan.part_blocks.size(), 0};
			for (GlyphBlock &sb : sub)
				if (!sb.is_empty()) {
					plan.part_blocks.push_back(std::move(sb));
					sp.n_parts++;
				}
			plan.splits.push_back(sp);
		}
	}
	plan.est_max_over_mean = ratio;
	// lanes: whole tasks and parts in task order; empty blocks go round
	plan.owner.assign(n_tasks, 0);
	plan.slot.assign(n_tasks, 0);
	plan.part_owner.assign(plan.part_blocks.size(), 0);
	plan.part_slot.assign(plan.part_blocks.size(), 0);
	plan.lane_tasks.resize(world);
	plan.lane_blocks.assign(world, 0);
	std::vector<int> split_of(n_tasks, -1);
	for (size_t k = 0; k < plan.splits.size(); k++)
		split_of[plan.splits[k].task] = (int)k;
	// lane of every (task, live part): items list the parts with glyphs in ascending part order, as part_blocks does
	std::vector<std::vector<uint32_t>> lanes_of(n_tasks);
	for (size_t k = 0; k < items.size(); k++)
		lanes_of[items[k].task].push_back(item_lane[k]);
	uint32_t rr = 0;
	for (size_t i = 0; i < n_tasks; i++) {
		if (plan.all[i].block.is_empty()) {
			const uint32_t r = rr++ % world;
			plan.owner[i] = r;
			plan.slot[i] = (uint32_t)plan.lane_tasks[r].size();
			plan.lane_tasks[r].push_back(plan.all[i]);
			plan.lane_blocks[r]++;
		} else if (split_of[i] < 0) {
			const uint32_t r = lanes_of[i].at(0);
			plan.owner[i] = r;
			plan.slot[i] = (uint32_t)plan.lane_tasks[r].size();
			plan.lane_tasks[r].push_back(plan.all[i]);
			plan.lane_blocks[r]++;
		} else {
			const LanePlan::Split &sp = plan.splits[(size_t)split_of[i]];
			if (lanes_of[i].size() != sp.n_parts)
				throw std::logic_error("lane plan: a split block's parts and items disagree");
			plan.owner[i] = LanePlan::kSplit;
			plan.slot[i] = (uint32_t)split_of[i];
			for (uint32_t pi = 0; pi < sp.n_parts; pi++) {
				const uint32_t r = lanes_of[i][pi];
				plan.part_owner[sp.first_part + pi] = r;
				plan.part_slot[sp.first_part + pi] = (uint32_t)plan.lane_tasks[r].size();
				plan.lane_tasks[r].push_back(Todo{plan.all[i].name, plan.part_blocks[sp.first_part + pi]});
				if (pi == 0)
					plan.lane_blocks[r]++;
			}
		}
	}
	lane_plan_ = std::move(plan);
}

bool FontManager::plan_lanes(const std::string &font_id, uint32_t world, std::vector<uint8_t> &owner, uint32_t &n_split_blocks, double *est_max_over_mean,
                             std::string *err)
{
	if (fonts_.find(font_id) == fonts_.end() || world == 0 || world > 254) {
		if (err)
			*err = fonts_.find(font_id) == fonts_.end() ? "unknown font id " + font_id : "plan_lanes: world must be 1..254";
		return false;
	}
	const int form = lane_form_ == 1 ? 1 : 2;
	if (lane_plan_.world != world || lane_plan_.form != form)
		build_lane_plan(world, form);
	const LanePlan &plan = lane_plan_;
	owner.assign(0x10000, 0xFF);
	n_split_blocks = 0;
	for (size_t i = 0; i < plan.all.size(); i++) {
		if (*plan.all[i].name != font_id || plan.all[i].block.is_empty())
			continue;
		const GlyphBlock &blk = plan.all[i].block;
		if (plan.owner[i] != LanePlan::kSplit) {
			for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
				if (blk.glyphs[ci])
					owner[blk.start_index + ci] = (uint8_t)plan.owner[i];
			continue;
		}
		n_split_blocks++;
		const LanePlan::Split &sp = plan.splits[plan.slot[i]];
		for (uint32_t pi = 0; pi < sp.n_parts; pi++) {
			const GlyphBlock &pb = plan.part_blocks[sp.first_part + pi];
			for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
				if (pb.glyphs[ci])
					owner[pb.start_index + ci] = (uint8_t)plan.part_owner[sp.first_part + pi];
		}
	}
	if (est_max_over_mean)
		*est_max_over_mean = plan.est_max_over_mean;
	return true;
}

void FontManager::render_tasks_multi(Writer &writer, const Renderer &renderer, int form)
{
	const double t_start = now_s();
	const uint32_t world = (uint32_t)renderer.n_devices();
	if (lane_plan_.world != world || lane_plan_.form != form) // (a font set's plan is kept: building it costs as much as a small run)
		build_lane_plan(world, form);
	std::vector<Todo> &all = lane_plan_.all;
	const std::vector<const std::string *> &names = lane_plan_.names;
	const std::vector<uint32_t> &owner = lane_plan_.owner, &slot = lane_plan_.slot;
	std::vector<std::vector<Todo>> &lane_tasks = lane_plan_.lane_tasks;
	const double t_sharded = now_s();

	renderer.reset_counters();
	std::vector<CaptureWriter> parts;
	parts.reserve(world);
	for (uint32_t r = 0; r < world; r++)
		parts.emplace_back(children_[r]->capture_store_, children_[r]->capture_files_);
	// one item per lane on this manager's pool: its workers carry the lanes (each lane forks on its own, smaller pool);
	// creating a thread per lane and run cost as much as a lane's share of a small run.  First error aborts (manager.rs:117-121)
	pool().run(world, [&](size_t r, unsigned) { children_[r]->run_tasks(lane_tasks[r], parts[r], renderer.device_lane((int)r)); });
	const double t_rendered = now_s();
	for (uint32_t r = 0; r < world; r++)
		if (parts[r].files.size() != lane_tasks[r].size())
			throw std::runtime_error("render_glyphs: a device lane produced " + std::to_string(parts[r].files.size()) + " files instead of " +
			                         std::to_string(lane_tasks[r].size()));
	// the split blocks: their parts hold consecutive runs of one block's glyphs and are handed to the sink as they lie — the
	// block's header and the lanes' entry regions (Writer::write_gather) — instead of being joined first
	const std::vector<LanePlan::Split> &splits = lane_plan_.splits;
	const double t_merged = now_s();
	timings_ = RenderTimings{};
	for (const std::string *name : names)
		writer.write_directory(*name + "/");
	std::string path;
	std::vector<std::pair<const uint8_t *, size_t>> ps, pieces;
	std::vector<uint8_t> head;
	for (size_t i = 0; i < all.size(); i++) {
		all[i].block.path_into(*all[i].name, path);
		if (owner[i] == LanePlan::kSplit) {
			const LanePlan::Split &sp = splits[slot[i]];
			ps.clear();
			for (uint32_t pi = 0; pi < sp.n_parts; pi++) {
				const CaptureWriter &lane = parts[lane_plan_.part_owner[sp.first_part + pi]];
				const uint32_t at = lane_plan_.part_slot[sp.first_part + pi];
				ps.emplace_back(lane.data(at), lane.size(at));
			}
			if (plan_pbf_concat(ps, head, pieces)) {
				writer.write_gather(path, pieces.data(), pieces.size());
				for (const auto &pc : pieces)
					timings_.pbf_bytes += pc.second;
			} else {
				const std::vector<uint8_t> joined = merge_pbf_partials(ps);
				writer.write_bytes(path, joined.data(), joined.size());
				timings_.pbf_bytes += joined.size();
			}
			continue;
		}
		const CaptureWriter &lane = parts[owner[i]];
		writer.write_bytes(path, lane.data(slot[i]), lane.size(slot[i]));
		timings_.pbf_bytes += lane.size(slot[i]);
	}
	const double t_written = now_s();
	uint64_t want[3] = {all.size(), 0, 0};
	for (uint32_t r = 0; r < world; r++) {
		const RenderTimings &ct = children_[r]->timings_;
		renderer.device_lane(r).add_counters(lane_plan_.lane_blocks[r], ct.glyphs, ct.pixels);
		want[1] += ct.glyphs;
		want[2] += ct.pixels;
		timings_.tessellate_s = std::max(timings_.tessellate_s, ct.tessellate_s);
		timings_.pack_s = std::max(timings_.pack_s, ct.pack_s);
		timings_.device_s = std::max(timings_.device_s, ct.device_s);
		timings_.encode_s = std::max(timings_.encode_s, ct.encode_s);
		timings_.glyphs += ct.glyphs;
		timings_.rasters += ct.rasters;
		timings_.pixels += ct.pixels;
		timings_.segments += ct.segments;
		timings_.glyf_groups += ct.glyf_groups;
		timings_.glyf_fallbacks += ct.glyf_fallbacks;
	}
	renderer.reduce_counters(reduced_);
	if (std::memcmp(reduced_, want, sizeof want) != 0)
		throw std::runtime_error("render_glyphs: the reduced run counters differ from the lanes' own");
	timings_.blocks = all.size();
	timings_.pack_s += t_sharded - t_start;
	timings_.encode_s += t_merged - t_rendered; // merge of the split blocks' parts
	timings_.write_s = t_written - t_merged;
	timings_.total_s = now_s() - t_start;
}

void FontManager::render_glyphs_multi(Writer &writer, const Renderer &renderer)
{
	const double t_start = now_s();
	const uint32_t world = (uint32_t)renderer.n_devices();
	ThreadPool &tp = pool();
	// lanes: one per device entry, each with its share of the host threads; kept between runs
	if (children_.size() != world) {
		children_.clear();
		for (uint32_t r = 0; r < world; r++)
			children_.emplace_back(new FontManager(this, r, world));
	}
	const unsigned per_lane = std::max(1u, worker_count() / world);
	for (auto &c : children_) {
		c->device_front_end_ = device_front_end_;
		c->in_place_pbf_ = in_place_pbf_;
		c->glyf_on_device_ = glyf_on_device_;
		c->batch_blocks_ = batch_blocks_;
		c->batch_blocks_set_ = batch_blocks_set_;
		c->set_threads(per_lane);
	}
	// Lane form: 2 (default) = whole (font, block) tasks, the heaviest blocks split between lanes where whole tasks do not
	// balance (hybrid); 1 = whole tasks only; 0 = every font's glyphs sharded over the lanes and every block merged.
	// set_lane_form / VG_LANE_TASKS = 0 / 1 / 2 forces one.
	{
		static const char *force = std::getenv("VG_LANE_TASKS");
		const int form = lane_form_ >= 0 ? lane_form_ : (force && force[0] >= '0' && force[0] <= '2' ? force[0] - '0' : 2);
		if (form != 0) {
			render_tasks_multi(writer, renderer, form);
			return;
		}
	}
	// shard tables of every font, built on this manager's pool before the lanes start (they only read them)
	for (const auto &[name, font] : fonts_)
		(void)cached_shard(name, font, world);
	const double t_sharded = now_s();

	renderer.reset_counters();
	std::vector<CaptureWriter> parts;
	parts.reserve(world);
	for (uint32_t r = 0; r < world; r++)
		parts.emplace_back(children_[r]->capture_store_, children_[r]->capture_files_);
	// (the pool's workers carry the lanes: see render_tasks_multi; first error aborts, manager.rs:117-121)
	tp.run(world, [&](size_t r, unsigned) { children_[r]->render_glyphs(parts[r], renderer.device_lane((int)r)); });
	const double t_rendered = now_s();

	// merge: block b of every lane holds a disjoint subset of the block's glyphs
	std::vector<const std::string *> names;
	for (const auto &kv : fonts_)
		names.push_back(&kv.first);
	const size_t n_files = names.size() * (0x10000 / GLYPH_BLOCK_SIZE);
	for (const CaptureWriter &p : parts)
		if (p.files.size() != n_files)
			throw std::runtime_error("render_glyphs: a device lane produced " + std::to_string(p.files.size()) + " files instead of " + std::to_string(n_files));
	std::vector<std::vector<uint8_t>> merged(n_files);
	tp.run(n_files, [&](size_t i, unsigned) {
		std::vector<std::pair<const uint8_t *, size_t>> ps;
		for (const CaptureWriter &p : parts)
			ps.emplace_back(p.data(i), p.size(i));
		merged[i] = merge_pbf_partials(ps);
	});
	const double t_merged = now_s();
	timings_ = RenderTimings{};
	for (const std::string *name : names)
		writer.write_directory(*name + "/");
	for (size_t i = 0; i < n_files; i++) {
		const uint32_t start = (uint32_t)(i % (0x10000 / GLYPH_BLOCK_SIZE)) * GLYPH_BLOCK_SIZE;
		writer.write_file(*names[i / (0x10000 / GLYPH_BLOCK_SIZE)] + "/" + std::to_string(start) + "-" + std::to_string(start + GLYPH_BLOCK_SIZE - 1) + ".pbf",
		                  merged[i]);
		timings_.pbf_bytes += merged[i].size();
	}
	const double t_written = now_s();

	// run counters: lane r is credited with its own glyphs and pixels and with the blocks it owns (block index mod N),
	// the lanes' contexts sum them (RCCL when the devices are distinct) and the sum must be what this process knows
	uint64_t want[3] = {n_files, 0, 0};
	for (uint32_t r = 0; r < world; r++) {
		const RenderTimings &ct = children_[r]->timings_;
		renderer.device_lane(r).add_counters((n_files + world - 1 - r) / world, ct.glyphs, ct.pixels);
		want[1] += ct.glyphs;
		want[2] += ct.pixels;
		timings_.tessellate_s = std::max(timings_.tessellate_s, ct.tessellate_s);
		timings_.pack_s = std::max(timings_.pack_s, ct.pack_s);
		timings_.device_s = std::max(timings_.device_s, ct.device_s);
		timings_.encode_s = std::max(timings_.encode_s, ct.encode_s);
		timings_.glyphs += ct.glyphs;
		timings_.rasters += ct.rasters;
		timings_.pixels += ct.pixels;
		timings_.segments += ct.segments;
		timings_.glyf_groups += ct.glyf_groups;
		timings_.glyf_fallbacks += ct.glyf_fallbacks;
	}
	renderer.reduce_counters(reduced_);
	if (std::memcmp(reduced_, want, sizeof want) != 0)
		throw std::runtime_error("render_glyphs: the reduced run counters differ from the lanes' own");
	timings_.blocks = n_files;
	timings_.pack_s += t_sharded - t_start;              // (shard tables: first run of a font set only)
	timings_.encode_s += t_merged - t_rendered;          // merge of the partials
	timings_.write_s = t_written - t_merged;
	timings_.total_s = now_s() - t_start;
}

void FontManager::render_blocks(Writer &writer, const Renderer &renderer, const std::string &font_id,
                                const std::vector<uint32_t> &block_starts)
{
	auto it = fonts().find(font_id);
	if (it == fonts().end())
		throw std::runtime_error("unknown font id " + font_id);
	const std::vector<GlyphBlock> &blocks = task_blocks(it->first, it->second);
	std::vector<Todo> tasks;
	for (uint32_t start : block_starts) {
		if (start % GLYPH_BLOCK_SIZE || start / GLYPH_BLOCK_SIZE >= blocks.size())
			throw std::runtime_error("bad block start " + std::to_string(start));
		tasks.push_back(Todo{&it->first, blocks[start / GLYPH_BLOCK_SIZE]});
	}
	run_tasks(tasks, writer, renderer);
}

// The same dispatcher with the DEVICE front-end: host threads only look glyphs up and record
// their outline commands; flattening, ring rules, scale/shift, bbox and the raster run on the
// GPU (one prepare + one render submission per group of blocks).
// Host half of the device front-end for tasks [G.g0, G.g1): look the glyphs up, record their outline
// commands on the pool (64-code-point slices, worker-local buffers), merge in task order into the
// group's page-locked arrays.
// units of host work of a group: 64-code-point slices of its non-empty blocks, in task order
void FontManager::fe_make_slices(const std::vector<Todo> &tasks, FeGroup &G, uint32_t per_slice)
{
	G.slices.clear();
	G.slice_ci.clear();
	for (size_t t = G.g0; t < G.g1; t++) {
		if (tasks[t].block.is_empty())
			continue;
		for (uint32_t c = 0; c < GLYPH_BLOCK_SIZE; c += per_slice) {
			OSlice s;
			s.task = (uint32_t)t;
			G.slices.push_back(s);
			G.slice_ci.push_back(c);
		}
	}
}

// The group's glyphs for the device's glyf decoder (vgsdf_outlines_glyf): the workers look every glyph up and copy the
// arrays of its simple glyphs as they stand — no point is decoded on the host (0.56 us of CPU per glyph with the
// recorder below, ~0.1 here).  Same slices, same merge in task order as fe_record.
bool FontManager::fe_record_glyf(const std::vector<Todo> &tasks, FeGroup &G)
{
	constexpr uint32_t kSlice = 64;
	ThreadPool &tp = pool();
	const double t0 = now_s();
	std::vector<OSlice> &slices = G.slices;
	fe_make_slices(tasks, G, kSlice);
	for (Worker &w : workers_)
		w.plocal.clear();
	tp.run(slices.size(), [&](size_t i, unsigned wid) {
		OSlice &s = slices[i];
		Worker &w = workers_[wid];
		s.worker = wid;
		s.job0 = (uint32_t)w.plocal.jobs.size();
		const GlyphBlock &blk = tasks[s.task].block;
		for (uint32_t ci = G.slice_ci[i]; ci < G.slice_ci[i] + kSlice; ci++)
			if (const FontFileEntry *f = blk.glyphs[ci])
				Renderer::record_parts(f->face(), blk.start_index + ci, w.plocal);
		s.job1 = (uint32_t)w.plocal.jobs.size();
		const GlyfPartsBatch &l = w.plocal;
		auto end_of = [&](uint32_t part) { return part < l.parts.size() ? l.parts[part].byte_off : (uint32_t)l.bytes.size(); };
		s.n_cmd = l.slot_off[s.job1] - l.slot_off[s.job0];
		s.n_dat = l.part_off[s.job1] - l.part_off[s.job0];
		s.n_byte = end_of(l.part_off[s.job1]) - end_of(l.part_off[s.job0]);
	}, true);
	const double t1 = now_s();
	timings_.tessellate_s += t1 - t0;
	// a worker's batch stays below 2^26 bytes / slots (Face::glyph_parts); the merged batch must fit 32-bit offsets too
	{
		uint64_t bytes = 0, slots = 0, n_p = 0;
		bool overflow = false;
		for (const Worker &w : workers_) {
			overflow = overflow || w.plocal.overflow;
			bytes += w.plocal.bytes.size();
			slots += w.plocal.slots;
			n_p += w.plocal.parts.size();
		}
		if (overflow || bytes >= (1ull << 31) || slots >= (1ull << 31) || n_p >= (1ull << 31))
			return false;
	}

	static const bool trace_pack = std::getenv("VG_TRACE_PACK") != nullptr;
	// merge in task order: jobs, command slots, parts and bytes of a slice are contiguous in its worker's batch
	uint32_t n_jobs = 0, n_slots = 0, n_parts = 0, n_bytes = 0;
	G.slice_cmd.resize(slices.size());
	G.slice_part.resize(slices.size());
	G.slice_byte.resize(slices.size());
	auto byte_at = [](const GlyfPartsBatch &l, uint32_t part) { // first byte of `part` (the end of the store behind the last one)
		return part < l.parts.size() ? l.parts[part].byte_off : (uint32_t)l.bytes.size();
	};
	for (size_t i = 0; i < slices.size(); i++) {
		OSlice &s = slices[i];
		s.g_job = n_jobs;
		G.slice_cmd[i] = n_slots;
		G.slice_part[i] = n_parts;
		G.slice_byte[i] = n_bytes;
		n_jobs += s.job1 - s.job0;
		n_slots += s.n_cmd;
		n_parts += s.n_dat;
		n_bytes += s.n_byte;
	}
	G.n_jobs = n_jobs;
	MergedOutlines &m = G.m;
	m.jobs.resize(n_jobs);
	G.in_place = in_place_pbf_;
	const double tp0 = now_s();
	m.layout_glyf(n_jobs, n_parts, n_bytes, G.in_place);
	m.cmd_off[0] = 0;
	const double tp1 = now_s();
	tp.run(slices.size(), [&](size_t i, unsigned) {
		const OSlice &s = slices[i];
		const GlyfPartsBatch &l = workers_[s.worker].plocal;
		const uint32_t p0 = l.part_off[s.job0], p1 = l.part_off[s.job1], s0 = l.slot_off[s.job0];
		const uint32_t b0 = byte_at(l, p0), b1 = byte_at(l, p1);
		if (b1 > b0)
			std::memcpy(m.glyf_bytes + G.slice_byte[i], l.bytes.data() + b0, b1 - b0);
		for (uint32_t k = p0; k < p1; k++) {
			vgsdf_glyf_part q;
			static_assert(sizeof q == sizeof l.parts[k], "same record");
			std::memcpy(&q, &l.parts[k], sizeof q);
			q.byte_off = G.slice_byte[i] + (l.parts[k].byte_off - b0);
			q.cmd_at = G.slice_cmd[i] + (l.parts[k].cmd_at - s0);
			m.parts[G.slice_part[i] + (k - p0)] = q;
		}
		for (uint32_t j = s.job0; j < s.job1; j++) {
			const uint32_t g = s.g_job + (j - s.job0);
			m.jobs[g] = l.jobs[j];
			m.scale[g] = l.scale[j];
			m.shift_x[g] = l.shift_x[j];
			if (m.pbf_fix) {
				m.pbf_pre[g] = 0;
				m.pbf_fix[g] = pbf_fix_of(l.jobs[j].id, l.jobs[j].advance);
			}
			m.cmd_off[g + 1] = G.slice_cmd[i] + (l.slot_off[j + 1] - s0);
		}
	}, true);
	const double tp2 = now_s();
	fe_layout_common(tasks, G);
	if (trace_pack)
		std::fprintf(stderr, "[pack] slices %zu jobs %u parts %u bytes %u: sums %.1f us, layout %.1f, copy fork %.1f, common %.1f\n", slices.size(), n_jobs,
		             n_parts, n_bytes, (tp0 - t1) * 1e6, (tp1 - tp0) * 1e6, (tp2 - tp1) * 1e6, (now_s() - tp2) * 1e6);
	timings_.pack_s += now_s() - t1;
	return true;
}

// jobs of a task are contiguous in the merged batch: [task_g0[t], task_g0[t + 1]); the first glyph of a block leaves room
// for the block's file + fontstack header in front of its entry
void FontManager::fe_layout_common(const std::vector<Todo> &tasks, FeGroup &G)
{
	MergedOutlines &m = G.m;
	const std::vector<OSlice> &slices = G.slices;
	G.task_g0.assign(G.g1 - G.g0 + 1, G.n_jobs);
	size_t t_next = 0;
	for (size_t i = 0; i < slices.size(); i++)
		for (; t_next <= slices[i].task - G.g0; t_next++)
			G.task_g0[t_next] = slices[i].g_job;
	if (m.pbf_pre)
		for (size_t t = G.g0; t < G.g1; t++) {
			const uint32_t a = G.task_g0[t - G.g0], b = G.task_g0[t - G.g0 + 1];
			if (a < b)
				m.pbf_pre[a] = kPbfHeadRoom + pbf_block_fields(tasks[t].name->size(), tasks[t].block.range().size());
		}
}

void FontManager::fe_record(const std::vector<Todo> &tasks, FeGroup &G, bool allow_glyf)
{
	if (allow_glyf && glyf_on_device_) {
		bool all_glyf = true;
		for (size_t t = G.g0; t < G.g1 && all_glyf; t++)
			all_glyf = tasks[t].block.all_glyf && !glyf_refused_.count(tasks[t].name);
		if (all_glyf) {
			if (fe_record_glyf(tasks, G))
				return;
			for (size_t t = G.g0; t < G.g1; t++) // composite fan-out past the batch bounds: the host's reader from now on
				glyf_refused_.insert(tasks[t].name);
			timings_.glyf_fallbacks++;
		}
	}
	constexpr uint32_t kSlice = 64;
	ThreadPool &tp = pool();
	const double t0 = now_s();
	std::vector<OSlice> &slices = G.slices;
	fe_make_slices(tasks, G, kSlice);
	for (Worker &w : workers_)
		w.olocal.clear();
	tp.run(slices.size(), [&](size_t i, unsigned wid) {
		OSlice &s = slices[i];
		Worker &w = workers_[wid];
		s.worker = wid;
		s.job0 = (uint32_t)w.olocal.jobs.size();
		const GlyphBlock &blk = tasks[s.task].block;
		for (uint32_t ci = G.slice_ci[i]; ci < G.slice_ci[i] + kSlice; ci++)
			if (const FontFileEntry *f = blk.glyphs[ci])
				Renderer::record(f->face(), blk.start_index + ci, w.olocal);
		s.job1 = (uint32_t)w.olocal.jobs.size();
		s.n_cmd = w.olocal.cmd_off[s.job1] - w.olocal.cmd_off[s.job0];
		s.n_dat = w.olocal.dat_off[s.job1] - w.olocal.dat_off[s.job0];
	});
	const double t1 = now_s();
	timings_.tessellate_s += t1 - t0;

	// merge in task order, into the compact upload form (a kind byte per command + the coordinates its kind carries)
	uint32_t n_jobs = 0, n_cmds = 0, n_floats = 0;
	G.slice_cmd.resize(slices.size());
	G.slice_dat.resize(slices.size());
	for (size_t i = 0; i < slices.size(); i++) {
		OSlice &s = slices[i];
		s.g_job = n_jobs;
		G.slice_cmd[i] = n_cmds;
		G.slice_dat[i] = n_floats;
		n_jobs += s.job1 - s.job0;
		n_cmds += s.n_cmd;
		n_floats += s.n_dat;
	}
	G.n_jobs = n_jobs;
	MergedOutlines &m = G.m;
	m.jobs.resize(n_jobs);
	G.in_place = in_place_pbf_;
	m.layout(n_jobs, n_cmds, n_floats, G.in_place);
	m.cmd_off[0] = 0;
	m.dat_off[0] = 0;
	tp.run(slices.size(), [&](size_t i, unsigned) {
		const OSlice &s = slices[i];
		const PackedOutlineBatch &l = workers_[s.worker].olocal;
		const uint32_t lc0 = l.cmd_off[s.job0], lc1 = l.cmd_off[s.job1], ld0 = l.dat_off[s.job0], ld1 = l.dat_off[s.job1];
		// (the workers record in the upload form itself: merging is a copy)
		if (lc1 > lc0)
			std::memcpy(m.kinds + G.slice_cmd[i], l.kinds.data() + lc0, lc1 - lc0);
		if (ld1 > ld0)
			std::memcpy(m.coords + G.slice_dat[i], l.coords.data() + ld0, sizeof(float) * (ld1 - ld0));
		for (uint32_t j = s.job0; j < s.job1; j++) {
			const uint32_t g = s.g_job + (j - s.job0);
			m.jobs[g] = l.jobs[j];
			m.scale[g] = l.scale[j];
			m.shift_x[g] = l.shift_x[j];
			if (m.pbf_fix) {
				m.pbf_pre[g] = 0;
				m.pbf_fix[g] = pbf_fix_of(l.jobs[j].id, l.jobs[j].advance);
			}
			m.cmd_off[g + 1] = G.slice_cmd[i] + (l.cmd_off[j + 1] - lc0);
			m.dat_off[g + 1] = G.slice_dat[i] + (l.dat_off[j + 1] - ld0);
		}
	});
	fe_layout_common(tasks, G);
	timings_.pack_s += now_s() - t1;
}

// In-place assembly: the raster has stored every bitmap of the group where its block's finished PBF has it (the arena
// G.out, laid out by outline_plan from pbf_pre / pbf_fix); what is left is the ~20 bytes around each bitmap and the
// block headers, written here on the pool.  A block without a glyph of this group is encoded on its own (32 bytes).
// Every position the device reports is checked against this side's own arithmetic.
// What of the assembly needs no result of the device: the files of the blocks without a glyph of this group (211 of a
// font's 256, typically: name + range only, ~35 bytes each, written into one store — a vector per file cost more than all
// the header bytes of the font together) and the list of the others.  Runs on the calling thread between the submission and
// the wait for the front-end's results, when it has nothing else to do.
void FontManager::fe_prepare_pieces(const std::vector<Todo> &tasks, FeGroup &G)
{
	const double t3 = now_s();
	const size_t nb = G.g1 - G.g0;
	using Piece = FeGroup::Piece;
	std::vector<Piece> &piece = G.piece;
	piece.assign(nb, Piece{});
	size_t small_stride = 0;
	for (size_t i = 0; i < nb; i++)
		small_stride = std::max(small_stride, tasks[G.g0 + i].name->size() + 48);
	std::vector<uint8_t> &small = G.small;
	small.resize(nb * small_stride);
	G.busy.clear();
	auto empty_file = [&](size_t i) { // the file of a block without a glyph of this group: name + range
		const Todo &td = tasks[G.g0 + i];
		const std::string &range = td.block.range();
		uint8_t *entries = small.data() + i * small_stride + kPbfHeadRoom + pbf_block_fields(td.name->size(), range.size());
		uint8_t *file = write_pbf_block_header(entries, *td.name, range, 0);
		piece[i] = Piece{file, (size_t)(entries - file)};
	};
	if (nb >= 1024) {
		// many fonts in one group (21 fixture fonts: 2688 tasks per group, 2500 of them such files): 50 us on the calling
		// thread, which a run over many fonts has no device latency to hide behind — runs of 128 tasks on the pool
		constexpr size_t kRun = 128;
		pool().run((nb + kRun - 1) / kRun, [&](size_t c, unsigned) {
			for (size_t i = c * kRun; i < std::min(nb, (c + 1) * kRun); i++)
				if (G.task_g0[i] == G.task_g0[i + 1])
					empty_file(i);
		}, true);
		for (size_t i = 0; i < nb; i++)
			if (G.task_g0[i] != G.task_g0[i + 1])
				G.busy.push_back((uint32_t)i);
	} else {
		for (size_t i = 0; i < nb; i++) {
			if (G.task_g0[i] != G.task_g0[i + 1])
				G.busy.push_back((uint32_t)i);
			else
				empty_file(i);
		}
	}
	timings_.encode_s += now_s() - t3;
}

void FontManager::fe_assemble(const std::vector<Todo> &tasks, FeGroup &G)
{
	ThreadPool &tp = pool();
	const double t3 = now_s();
	MergedOutlines &m = G.m;
	using Piece = FeGroup::Piece;
	std::vector<Piece> &piece = G.piece;
	std::atomic<uint64_t> n_raster{0}, n_pixels{0};
	std::atomic<bool> mismatch{false};
	uint8_t *arena = G.out.data();
	tp.run(G.busy.size(), [&](size_t bi, unsigned) {
		const size_t i = G.busy[bi];
		const Todo &td = tasks[G.g0 + i];
		const uint32_t a = G.task_g0[i], b = G.task_g0[i + 1];
		uint64_t rasters = 0, pixels = 0;
		uint8_t *first = nullptr, *end = nullptr;
		for (uint32_t g = a; g < b; g++) {
			const vgsdf_rect &r = G.rects[g];
			const GlyphJob &job = m.jobs[g];
			const bool has = r.has_raster != 0;
			// start of the entry from the bitmap's position: 0x1A varint(msg) 0x08 varint(id) [0x12 varint(w h)] come before it
			const uint64_t px = has ? (uint64_t)r.w * r.h : 0;
			const PbfEntrySize es = pbf_entry_size(job.id, job.advance, has, r.w, r.h, r.x0, r.y0);
			const uint64_t before = es.bitmap_at;
			if (G.pbf_at[g] < before || G.pbf_at[g] - before + es.total > G.out_bytes) {
				mismatch = true; // the entry does not lie inside the arena
				return;
			}
			uint8_t *entry = arena + (G.pbf_at[g] - before);
			if (g == a) {
				first = entry;
			} else if (entry != end) {
				mismatch = true; // the entries of a block follow each other without a gap
				return;
			}
			end = entry + write_pbf_entry_headers(entry, job.id, job.advance, has, r.w, r.h, r.x0, r.y0);
			rasters += has;
			pixels += px;
		}
		if ((size_t)(first - arena) < m.pbf_pre[a]) {
			mismatch = true;
			return;
		}
		uint8_t *file = write_pbf_block_header(first, *td.name, td.block.range(), (size_t)(end - first));
		piece[i] = Piece{file, (size_t)(end - file)};
		n_raster += rasters;
		n_pixels += pixels;
	}, true);
	if (mismatch)
		throw std::runtime_error("in-place PBF assembly: the device's layout of the arena differs from the host's");
	G.n_raster = n_raster;
	G.n_pixels = n_pixels;
	timings_.encode_s += now_s() - t3;
}

void FontManager::fe_write_pieces(const std::vector<Todo> &tasks, FeGroup &G, Writer &writer)
{
	const double t4 = now_s();
	const size_t nb = G.g1 - G.g0;
	std::string path;
	for (size_t i = 0; i < nb; i++) {
		tasks[G.g0 + i].block.path_into(*tasks[G.g0 + i].name, path);
		writer.write_bytes(path, G.piece[i].p, G.piece[i].n);
		timings_.pbf_bytes += G.piece[i].n;
	}
	timings_.write_s += now_s() - t4;
	timings_.blocks += nb;
	timings_.glyphs += G.n_jobs;
	timings_.rasters += G.n_raster;
	timings_.pixels += G.n_pixels;
	timings_.segments += G.n_segs;
}

// Rects + bitmaps of a rendered group -> PbfGlyphs per block (pool), written in task order.
void FontManager::fe_encode_write(const std::vector<Todo> &tasks, FeGroup &G, Writer &writer)
{
	ThreadPool &tp = pool();
	const double t3 = now_s();
	const size_t nb = G.g1 - G.g0;
	const uint32_t n_jobs = G.n_jobs;
	MergedOutlines &m = G.m;
	// bitmap offsets: rasterised glyphs are packed in job order
	std::vector<uint64_t> boff((size_t)n_jobs + 1, 0);
	uint64_t n_raster = 0;
	for (uint32_t g = 0; g < n_jobs; g++) {
		const vgsdf_rect &r = G.rects[g];
		GlyphJob &job = m.jobs[g];
		job.has_raster = r.has_raster != 0;
		job.x0 = r.x0;
		job.y0 = r.y0;
		job.width = r.w;
		job.height = r.h;
		job.x1 = r.x0 + (int32_t)r.w;
		job.y1 = r.y0 + (int32_t)r.h;
		job.n_segments = r.n_segments;
		boff[g + 1] = boff[g] + (job.has_raster ? (uint64_t)r.w * r.h : 0);
		n_raster += job.has_raster;
	}
	std::vector<std::pair<size_t, size_t>> span(nb, {0, 0});
	for (size_t i = 0; i < G.slices.size(); i++) {
		auto &sp = span[G.slices[i].task - G.g0];
		if (sp.second == 0)
			sp.first = i;
		sp.second = i + 1;
	}
	std::vector<std::vector<uint8_t>> encoded(nb);
	tp.run(nb, [&](size_t i, unsigned) {
		std::vector<PbfGlyphRef> refs;
		for (size_t k = span[i].first; k < span[i].second; k++) {
			const OSlice &s = G.slices[k];
			for (uint32_t j = 0; j < s.job1 - s.job0; j++) {
				const uint32_t g = s.g_job + j;
				const GlyphJob &job = m.jobs[g];
				refs.push_back(job.to_pbf(job.has_raster ? G.out.data() + boff[g] : nullptr));
			}
		}
		encoded[i] = PbfGlyphs::encode(*tasks[G.g0 + i].name, tasks[G.g0 + i].block.range(), std::move(refs));
	});
	const double t4 = now_s();
	timings_.encode_s += t4 - t3;
	for (size_t i = 0; i < nb; i++) {
		writer.write_file(*tasks[G.g0 + i].name + "/" + tasks[G.g0 + i].block.filename(), encoded[i]);
		timings_.pbf_bytes += encoded[i].size();
	}
	timings_.write_s += now_s() - t4;
	timings_.blocks += nb;
	timings_.glyphs += n_jobs;
	timings_.rasters += n_raster;
	timings_.pixels += G.out_bytes;
	timings_.segments += G.n_segs;
}

// Device front-end dispatcher: groups of tasks (part of a large font, or several small ones) go through
// record (host pool) -> device (flatten, raster; one submission, one synchronisation) -> encode + write (host
// pool).  Two groups are in flight: while the GPU works on group k the host records group k + 1 and then encodes
// group k - 1 (the submissions alternate between the renderer's two lanes = device contexts; the calling thread
// only enqueues and waits, there is no second host thread).  Files are written in task order; the first error
// aborts (manager.rs:117-121).
void FontManager::run_tasks_device_front_end(std::vector<Todo> &tasks, Writer &writer, const Renderer &renderer)
{
	timings_ = RenderTimings{};
	const double t_start = now_s();
	(void)pool();
	// Group size: every group costs ~0.1 ms of device latency and three fork/joins of the host pool, so small fonts
	// are grouped (21 fixture fonts: 12.9 ms one font per group, 2.1 ms in one group) and a run is cut into several
	// groups — to overlap host and device — only when each keeps >= 5000 glyphs (measured in round 3, 32 threads on a
	// 16-CPU quota: the 14 180 glyphs of the 21 fixture fonts 2.5 / 2.1 / 1.8 / 2.0 / 2.1 ms with groups of at least
	// 2000 / 3500 / 5000 / 8000 / 20 000 glyphs; Noto Sans' 6445 glyphs 1.14 / 1.09 / 1.07 ms at 2000 / 5000 / 20 000).
	// An explicit set_batch_blocks() bounds the group in blocks instead.  (Round 4: a half-size FIRST group, to start the device
	// earlier — 275 of the 21 fonts' 1110 us pass before the first submission — made three groups of two and the run slower,
	// 1106 -> 1271 us: with the host phases of a group at 120-220 us whatever its size the run is bound by this thread.)
	size_t total_glyphs = 0;
	for (const Todo &t : tasks)
		total_glyphs += t.block.len();
	constexpr size_t kFeGlyphBudget = 32768;
	static const char *mg = std::getenv("VG_FE_MIN_GROUP"); // (measurement switch)
	// (round 4, after the device stage and the host phases of a group got shorter: one or two fonts — up to 512 tasks — do best
	// in groups of >= 3000 glyphs: Noto Sans' 20 files, 6480 glyphs, 0.69 ms as one group, 0.60 ms as two, 0.79 ms as three;
	// a group's host cost grows with its TASKS, most of them empty blocks, so the 21 fixture fonts keep >= 5000: 1.05 ms in two
	// groups, 1.50 ms in four)
	const size_t kFeMinGroup = mg ? (size_t)std::max(1, std::atoi(mg)) : (tasks.size() <= 512 ? 3000 : 5000);
	const size_t n_groups = std::max<size_t>(1, total_glyphs / kFeMinGroup);
	const size_t budget = std::min(kFeGlyphBudget, (total_glyphs + n_groups - 1) / n_groups);
	std::vector<std::pair<size_t, size_t>> groups;
	for (size_t g0 = 0; g0 < tasks.size();) {
		size_t g1 = g0, glyphs = 0;
		while (g1 < tasks.size() && (batch_blocks_set_ ? g1 - g0 < (size_t)batch_blocks_ : (g1 == g0 || glyphs < budget))) {
			glyphs += tasks[g1].block.len();
			g1++;
		}
		groups.emplace_back(g0, g1);
		g0 = g1;
	}
	bool in_flight[2] = {false, false};
	// VG_TRACE_PHASES=1 (measurement switch): the calling thread's time line of the run, one line per group on stderr
	static const bool trace_ph = std::getenv("VG_TRACE_PHASES") != nullptr;
	struct Mark {
		const char *what;
		size_t k;
		double t;
	};
	std::vector<Mark> marks;
	auto mark = [&](const char *what, size_t k) {
		if (trace_ph)
			marks.push_back(Mark{what, k, now_s()});
	};
	auto submit = [&](size_t k) {
		FeGroup &G = fe_group_[k & 1];
		G.g0 = groups[k].first;
		G.g1 = groups[k].second;
		mark("record+pack >", k);
		fe_record(tasks, G);
		mark("submit >", k);
		const double t = now_s();
		if (G.n_jobs) {
			if (G.m.glyf) {
				renderer.submit_outlines((int)(k & 1), G.m.view_glyf(), G.out);
				timings_.glyf_groups++;
			} else
				renderer.submit_outlines((int)(k & 1), G.m.view(), G.out);
			in_flight[k & 1] = true;
		}
		timings_.device_s += now_s() - t;
		mark("submitted", k);
	};
	auto collect = [&](size_t k) {
		FeGroup &G = fe_group_[k & 1];
		double t = now_s();
		mark("pieces >", k);
		G.rects.clear();
		G.out_bytes = G.n_segs = 0;
		// in-place assembly: the rects come back right behind the plan kernel, a good 100 us before the bitmaps — the
		// headers are written while the raster is still storing the bitmaps between them
		bool early = false;
		if (G.in_place && G.n_jobs)
			fe_prepare_pieces(tasks, G);
		t = now_s();
		mark("peek >", k);
		if (in_flight[k & 1] && G.in_place && G.n_jobs) {
			early = renderer.peek_outlines((int)(k & 1), G.rects, G.out_bytes, G.n_jobs, &G.pbf_at);
			timings_.device_s += now_s() - t;
			mark("assemble >", k);
			if (early)
				fe_assemble(tasks, G);
			t = now_s();
		}
		mark("wait >", k);
		if (in_flight[k & 1]) {
			in_flight[k & 1] = false;
			try {
				renderer.wait_outlines((int)(k & 1), G.rects, G.out, G.out_bytes, G.n_segs, G.n_jobs, G.in_place ? &G.pbf_at : nullptr);
			} catch (const GlyfEntryError &) {
				// a malformed `glyf` entry somewhere in the group: ttf-parser's rules for such glyphs (None for the glyph, the
				// rest of a composite skipped) are the host reader's — the group is recorded there and rendered again, now
				early = false;
				timings_.glyf_fallbacks++;
				for (size_t tk = G.g0; tk < G.g1; tk++) // (later groups and runs of these fonts skip the glyf form)
					glyf_refused_.insert(tasks[tk].name);
				const std::vector<uint32_t> g0_before = G.task_g0;
				fe_record(tasks, G, false);
				// the pieces prepared above (files of the empty blocks, list of the others) depend on which jobs a task has: both
				// recorders must look up the same glyphs
				if (G.task_g0 != g0_before)
					throw std::runtime_error("render_glyphs: the host's reader and the glyf parts disagree on a group's glyphs");
				t = now_s();
				renderer.submit_outlines((int)(k & 1), G.m.view(), G.out);
				renderer.wait_outlines((int)(k & 1), G.rects, G.out, G.out_bytes, G.n_segs, G.n_jobs, G.in_place ? &G.pbf_at : nullptr);
			}
		}
		timings_.device_s += now_s() - t;
		mark("write >", k);
		if (G.in_place && G.n_jobs) {
			if (!early)
				fe_assemble(tasks, G);
			fe_write_pieces(tasks, G, writer);
		} else {
			fe_encode_write(tasks, G, writer);
		}
		mark("done", k);
	};
	try {
		for (size_t k = 0; k < groups.size(); k++) {
			submit(k);
			if (k > 0)
				collect(k - 1);
		}
		if (!groups.empty())
			collect(groups.size() - 1);
	} catch (...) {
		// leave no submission behind (its lane stays held until it is waited for)
		for (int lane = 0; lane < 2; lane++)
			if (in_flight[lane]) {
				in_flight[lane] = false;
				try {
					FeGroup &G = fe_group_[lane];
					renderer.wait_outlines(lane, G.rects, G.out, G.out_bytes, G.n_segs, G.n_jobs);
				} catch (...) {
				}
			}
		throw;
	}
	timings_.total_s = now_s() - t_start;
	if (trace_ph) {
		std::string line = "[phases] " + std::to_string(groups.size()) + " groups, " + std::to_string(total_glyphs) + " glyphs:";
		char buf[96];
		for (const Mark &mk : marks) {
			std::snprintf(buf, sizeof buf, " | %s g%zu @%.0f", mk.what, mk.k, (mk.t - t_start) * 1e6);
			line += buf;
		}
		std::snprintf(buf, sizeof buf, " | end @%.0f us\n", timings_.total_s * 1e6);
		line += buf;
		std::fputs(line.c_str(), stderr);
	}
}

void FontManager::run_tasks(std::vector<Todo> &tasks, Writer &writer, const Renderer &renderer)
{
	if (device_front_end_ && renderer.mode() == Renderer::Mode::Hip) {
		run_tasks_device_front_end(tasks, writer, renderer);
		return;
	}
	timings_ = RenderTimings{};
	const double t_start = now_s();
	ThreadPool &tp = pool();
	std::vector<Slice> slices;

	// GPU batch dispatcher (replaces manager.rs:104-121): groups of `batch_blocks_` tasks are
	// tessellated on the pool, packed into page-locked SoA arrays, rendered with ONE device
	// submission, then PBF-encoded on the pool and written in task order.
	for (size_t g0 = 0; g0 < tasks.size(); g0 += batch_blocks_) {
		const size_t g1 = std::min(tasks.size(), g0 + (size_t)batch_blocks_);
		const size_t nb = g1 - g0;

		tessellate_and_pack(tasks, g0, g1, slices, packed_);
		// slices of a task are contiguous and in code point order
		std::vector<std::pair<size_t, size_t>> span(nb, {0, 0});
		for (size_t i = 0; i < slices.size(); i++) {
			auto &sp = span[slices[i].task - g0];
			if (sp.second == 0)
				sp.first = i;
			sp.second = i + 1;
		}
		// In-place assembly (as the device front-end's, fe_assemble_write): the bitmaps are rendered where the blocks'
		// finished PBFs have them.  The host knows every rect here, so it lays the arena out itself: out_off[g] = position
		// of bitmap g, out_off[n] = size of the arena (vgsdf.h allows gaps between the bitmaps).
		std::vector<uint64_t> entry_at; // per slice: first entry; entries of a slice follow each other
		if (in_place_pbf_) {
			const double tl = now_s();
			entry_at.assign(slices.size() + 1, 0);
			uint64_t pos = 0;
			for (size_t i = 0; i < nb; i++)
				for (size_t k = span[i].first; k < span[i].second; k++) {
					const Slice &s = slices[k];
					const GlyphBatch &l = workers_[s.worker].local;
					if (k == span[i].first)
						pos += kPbfHeadRoom + pbf_block_fields(tasks[g0 + i].name->size(), tasks[g0 + i].block.range().size());
					entry_at[k] = pos;
					uint32_t r = s.g_raster;
					for (uint32_t j = s.job0; j < s.job1; j++) {
						const GlyphJob &job = l.jobs[j];
						const PbfEntrySize es = pbf_entry_size(job.id, job.advance, job.has_raster, job.width, job.height, job.x0, job.y0);
						if (job.has_raster)
							packed_.out_off[r++] = pos + es.bitmap_at;
						pos += es.total;
					}
				}
			entry_at[slices.size()] = pos;
			packed_.out_off[packed_.n_raster] = pos;
			packed_.out_bytes = pos;
			packed_.out.ensure((size_t)pos + 1);
			timings_.pack_s += now_s() - tl;
		}
		const double t1 = now_s();

		renderer.render_packed(packed_);
		const double t2 = now_s();
		timings_.device_s += t2 - t1;

		std::vector<std::vector<uint8_t>> encoded(nb);
		struct Piece {
			const uint8_t *p = nullptr;
			size_t n = 0;
		};
		std::vector<Piece> piece(nb);
		uint64_t n_pixels = 0;
		for (const Slice &s : slices) {
			const GlyphBatch &l = workers_[s.worker].local;
			n_pixels += l.out_off[s.raster1] - l.out_off[s.raster0];
		}
		tp.run(nb, [&](size_t i, unsigned) {
			if (in_place_pbf_ && span[i].second > span[i].first) {
				uint8_t *arena = packed_.out.data();
				uint8_t *first = arena + entry_at[span[i].first], *p = first;
				for (size_t k = span[i].first; k < span[i].second; k++) {
					const Slice &s = slices[k];
					const GlyphBatch &l = workers_[s.worker].local;
					for (uint32_t j = s.job0; j < s.job1; j++) {
						const GlyphJob &job = l.jobs[j];
						p += write_pbf_entry_headers(p, job.id, job.advance, job.has_raster, job.width, job.height, job.x0, job.y0);
					}
				}
				uint8_t *file = write_pbf_block_header(first, *tasks[g0 + i].name, tasks[g0 + i].block.range(), (size_t)(p - first));
				piece[i] = Piece{file, (size_t)(p - file)};
				return;
			}
			std::vector<PbfGlyphRef> refs;
			for (size_t k = span[i].first; k < span[i].second; k++) {
				const Slice &s = slices[k];
				const GlyphBatch &l = workers_[s.worker].local;
				uint32_t r = s.raster0;
				for (uint32_t j = s.job0; j < s.job1; j++) {
					const GlyphJob &job = l.jobs[j];
					const uint8_t *bm = nullptr;
					if (job.has_raster) {
						bm = packed_.out.data() + packed_.out_off[s.g_raster + (r - s.raster0)];
						r++;
					}
					refs.push_back(job.to_pbf(bm));
				}
			}
			encoded[i] = PbfGlyphs::encode(*tasks[g0 + i].name, tasks[g0 + i].block.range(), std::move(refs));
			piece[i] = Piece{encoded[i].data(), encoded[i].size()};
		});
		const double t3 = now_s();
		timings_.encode_s += t3 - t2;

		std::string path;
		for (size_t i = 0; i < nb; i++) {
			tasks[g0 + i].block.path_into(*tasks[g0 + i].name, path);
			writer.write_bytes(path, piece[i].p, piece[i].n);
			timings_.pbf_bytes += piece[i].n;
		}
		timings_.write_s += now_s() - t3;

		timings_.blocks += nb;
		for (const Slice &s : slices)
			timings_.glyphs += s.job1 - s.job0;
		timings_.rasters += packed_.n_raster;
		timings_.pixels += n_pixels;
		timings_.segments += packed_.n_seg;
	}
	timings_.total_s = now_s() - t_start;
}

} // namespace vg
