#include "font_manager.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <thread>

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

// runs fn(i) for i in [0,n) on `workers` threads (dynamic scheduling); rethrows the first error
template <class F> void parallel_for(size_t n, unsigned workers, F &&fn)
{
	if (workers <= 1 || n <= 1) {
		for (size_t i = 0; i < n; i++)
			fn(i, 0u);
		return;
	}
	std::atomic<size_t> next{0};
	std::atomic<bool> failed{false};
	std::string first_error;
	std::mutex err_mu;
	std::vector<std::thread> pool;
	const unsigned nt = (unsigned)std::min<size_t>(workers, n);
	for (unsigned t = 0; t < nt; t++)
		pool.emplace_back([&, t] {
			for (;;) {
				const size_t i = next.fetch_add(1);
				if (i >= n || failed.load())
					return;
				try {
					fn(i, t);
				} catch (const std::exception &e) {
					std::lock_guard<std::mutex> l(err_mu);
					if (!failed.exchange(true))
						first_error = e.what();
				}
			}
		});
	for (auto &th : pool)
		th.join();
	if (failed.load())
		throw std::runtime_error(first_error);
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
	e->codepoints_ = e->face_.unicode_codepoints();
	return e;
}

std::string GlyphBlock::range() const
{
	return std::to_string(start_index) + "-" + std::to_string(start_index + GLYPH_BLOCK_SIZE - 1);
}

void GlyphBlock::prepare(TessScratch &scratch, GlyphBatch &batch) const
{
	for (uint32_t ci = 0; ci < GLYPH_BLOCK_SIZE; ci++)
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

bool FontManager::add_font_with_name(const std::string &name, const std::vector<std::string> &sources, std::string *err)
{
	return fonts_[name_to_id(name)].add_paths(sources, err);
}

bool FontManager::add_font_data(const std::string &name, std::vector<uint8_t> data, std::string *err)
{
	auto e = FontFileEntry::create(std::move(data), err);
	if (!e)
		return false;
	fonts_[name_to_id(name)].add_file(std::move(e));
	return true;
}

bool FontManager::add_path(const std::string &, std::string *err)
{
	if (err)
		*err = "add_path needs font family-name parsing (parse_font_name.rs), which is outside the "
		       "accelerated path; use add_font_with_name(name, sources)";
	return false;
}

unsigned FontManager::worker_count() const
{
	if (!parallel_)
		return 1;
	if (threads_)
		return threads_;
	// default: the machine's threads, capped at one GPU's CPU share of an 8-GPU node (16);
	// VG_THREADS or set_threads() override
	if (const char *e = std::getenv("VG_THREADS"))
		if (int v = std::atoi(e); v > 0)
			return (unsigned)v;
	const unsigned hc = std::thread::hardware_concurrency();
	return hc ? std::min(hc, 16u) : 1;
}

bool FontManager::build_batch(const std::string &font_id, GlyphBatch &out, std::string *err) const
{
	auto it = fonts_.find(font_id);
	if (it == fonts_.end()) {
		if (err)
			*err = "unknown font id " + font_id;
		return false;
	}
	const std::vector<GlyphBlock> blocks = it->second.get_blocks();
	std::vector<GlyphBatch> parts(blocks.size());
	const unsigned workers = worker_count();
	std::vector<TessScratch> scratch(workers);
	parallel_for(blocks.size(), workers, [&](size_t i, unsigned t) { blocks[i].prepare(scratch[t], parts[i]); });
	out.clear();
	for (const GlyphBatch &p : parts)
		out.append(p);
	return true;
}

void FontManager::render_glyphs(Writer &writer, const Renderer &renderer)
{
	// manager.rs:86-97: one task per (font, block); all 256 blocks per font
	std::vector<Todo> tasks;
	for (const auto &[name, font] : fonts_) {
		writer.write_directory(name + "/");
		for (GlyphBlock &b : font.get_blocks())
			tasks.push_back(Todo{&name, std::move(b)});
	}
	run_tasks(tasks, writer, renderer);
}

void FontManager::render_blocks(Writer &writer, const Renderer &renderer, const std::string &font_id,
                                const std::vector<uint32_t> &block_starts)
{
	auto it = fonts_.find(font_id);
	if (it == fonts_.end())
		throw std::runtime_error("unknown font id " + font_id);
	std::vector<GlyphBlock> blocks = it->second.get_blocks();
	std::vector<Todo> tasks;
	for (uint32_t start : block_starts) {
		if (start % GLYPH_BLOCK_SIZE || start / GLYPH_BLOCK_SIZE >= blocks.size())
			throw std::runtime_error("bad block start " + std::to_string(start));
		tasks.push_back(Todo{&it->first, blocks[start / GLYPH_BLOCK_SIZE]});
	}
	run_tasks(tasks, writer, renderer);
}

void FontManager::run_tasks(std::vector<Todo> &tasks, Writer &writer, const Renderer &renderer)
{
	timings_ = RenderTimings{};
	const double t_start = now_s();

	const unsigned workers = worker_count();
	std::vector<TessScratch> scratch(workers);
	std::vector<uint8_t> pixels;

	// GPU batch dispatcher (replaces manager.rs:104-121): groups of `batch_blocks_` tasks are
	// tessellated on host threads, rendered with one device submission, then encoded + written.
	for (size_t g0 = 0; g0 < tasks.size(); g0 += batch_blocks_) {
		const size_t g1 = std::min(tasks.size(), g0 + (size_t)batch_blocks_);
		const size_t nb = g1 - g0;

		double t0 = now_s();
		std::vector<GlyphBatch> parts(nb);
		parallel_for(nb, workers, [&](size_t i, unsigned t) { tasks[g0 + i].block.prepare(scratch[t], parts[i]); });
		GlyphBatch batch;
		std::vector<size_t> first_raster(nb + 1, 0);
		for (size_t i = 0; i < nb; i++) {
			first_raster[i] = batch.n_raster();
			batch.append(parts[i]);
		}
		first_raster[nb] = batch.n_raster();
		double t1 = now_s();
		timings_.tessellate_s += t1 - t0;

		pixels.resize((size_t)batch.out_bytes());
		renderer.render_batch(batch, pixels.data());
		double t2 = now_s();
		timings_.device_s += t2 - t1;

		std::vector<std::vector<uint8_t>> encoded(nb);
		parallel_for(nb, workers, [&](size_t i, unsigned) {
			const GlyphBatch &p = parts[i];
			std::vector<PbfGlyphRef> refs;
			refs.reserve(p.jobs.size());
			size_t r = first_raster[i];
			for (const GlyphJob &j : p.jobs) {
				const uint8_t *bm = j.has_raster ? pixels.data() + batch.out_off[r++] : nullptr;
				refs.push_back(j.to_pbf(bm));
			}
			encoded[i] = PbfGlyphs::encode(*tasks[g0 + i].name, tasks[g0 + i].block.range(), std::move(refs));
		});
		double t3 = now_s();
		timings_.encode_s += t3 - t2;

		for (size_t i = 0; i < nb; i++) {
			writer.write_file(*tasks[g0 + i].name + "/" + tasks[g0 + i].block.filename(), encoded[i]);
			timings_.pbf_bytes += encoded[i].size();
		}
		timings_.write_s += now_s() - t3;

		timings_.blocks += nb;
		timings_.glyphs += batch.jobs.size();
		timings_.rasters += batch.n_raster();
		timings_.pixels += batch.out_bytes();
		timings_.segments += batch.seg_off.back();
	}
	timings_.total_s = now_s() - t_start;
}

} // namespace vg
