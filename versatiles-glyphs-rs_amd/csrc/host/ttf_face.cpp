// ttf_face.cpp — see ttf_face.hpp.  Behavioural restatement of ttf-parser 0.25.1 for
// static `glyf` fonts: table lookup, cmap formats 0/4/6/10/12/13, hmtx advances, and the
// glyf outline emitter (implied on-curve midpoints in f32, composite transforms in f32).
// All f32 arithmetic is done operation by operation (library built with
// -ffp-contract=off), as Rust does.
#include "ttf_face.hpp"

#include "cff.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace vg {

namespace {

// start of the table directory of face 0 (ttf-parser RawFace::parse with index 0): the file itself, or — a font
// collection, magic 'ttcf' — the first face the collection lists; npos for any other magic than 0x00010000 / 'true' / 'OTTO'
size_t face_directory(Bytes file)
{
	constexpr size_t npos = (size_t)-1;
	auto is_face = [](uint32_t m) { return m == 0x00010000u || m == 0x74727565u || m == 0x4F54544Fu; };
	if (!file.has(0, 4))
		return npos;
	const uint32_t magic = file.u32(0);
	if (is_face(magic))
		return 0;
	if (magic != 0x74746366u || !file.has(4, 8)) // 'ttcf', version, numFonts
		return npos;
	// RawFace::parse reads the WHOLE offsets[numFonts] array (out of bounds: no face) and takes the face's offset relative
	// to the end of that array (`checked_sub`): a face inside the header or the array is refused
	const uint64_t n_fonts = file.u32(8);
	if (n_fonts == 0 || 4 * n_fonts > file.size() || !file.has(12, (size_t)(4 * n_fonts)))
		return npos;
	const size_t at = file.u32(12);
	if (at < 12 + 4 * n_fonts || !file.has(at, 4) || !is_face(file.u32(at))) // (a face is not a collection)
		return npos;
	return at;
}

Bytes find_table(Bytes file, const char tag[4])
{
	const size_t dir = face_directory(file);
	if (dir == (size_t)-1 || !file.has(dir, 12))
		return {};
	const uint16_t n = file.u16(dir + 4);
	for (uint16_t i = 0; i < n; i++) {
		const size_t rec = dir + 12 + (size_t)i * 16;
		if (!file.has(rec, 16))
			return {};
		if (std::memcmp(file.data() + rec, tag, 4) == 0)
			return file.sub(file.u32(rec + 8), file.u32(rec + 12));
	}
	return {};
}

} // namespace

std::optional<Face> Face::parse(const uint8_t *data, size_t len)
{
	Bytes file(data, len);
	const Bytes head = find_table(file, "head"), maxp = find_table(file, "maxp"),
	            hhea = find_table(file, "hhea");
	if (!head.has(0, 54) || !maxp.has(0, 6) || !hhea.has(0, 36))
		return std::nullopt;
	Face f;
	f.units_per_em_ = head.u16(18);
	if (f.units_per_em_ < 16 || f.units_per_em_ > 16384)
		return std::nullopt;
	f.loca_long_ = head.i16(50) != 0;
	f.num_glyphs_ = maxp.u16(4);
	f.num_hmetrics_ = hhea.u16(34);
	f.hmtx_ = find_table(file, "hmtx");
	f.loca_ = find_table(file, "loca");
	f.glyf_ = find_table(file, "glyf");
	if (!f.loca_.empty()) {
		size_t want = (size_t)f.num_glyphs_ + 1;
		if (f.num_glyphs_ == 0xFFFF)
			want = 0xFFFF;
		f.loca_entries_ = std::min(want, f.loca_.size() / (f.loca_long_ ? 4u : 2u));
	}
	f.name_ = find_table(file, "name");
	if (const Bytes cff = find_table(file, "CFF "); !cff.empty()) {
		if (auto t = CffTable::parse(cff))
			f.cff_ = std::make_shared<const CffTable>(std::move(*t));
		else
			f.cff_unreadable_ = true;
	}
	if (const Bytes cff2 = find_table(file, "CFF2"); !f.cff_ && !cff2.empty()) {
		// ttf-parser: glyf, then cff1, then cff2 — drawn at the face's variation coordinates, which the reference
		// leaves at their defaults: one (zero) coordinate per `fvar` axis, at most 64, none without a readable `fvar`
		// (version 1.0, at least one axis, the axis records of 20 bytes each inside the table)
		uint32_t n_coords = 0;
		if (const Bytes fvar = find_table(file, "fvar"); fvar.has(0, 10) && fvar.u32(0) == 0x00010000u) {
			const size_t axes_at = fvar.u16(4);
			const uint32_t n_axes = fvar.u16(8);
			if (n_axes != 0 && fvar.has(axes_at, (size_t)n_axes * 20))
				n_coords = std::min<uint32_t>(n_axes, 64);
		}
		if (auto t = CffTable::parse2(cff2, n_coords))
			f.cff_ = std::make_shared<const CffTable>(std::move(*t));
		else
			f.cff_unreadable_ = true;
	}
	const Bytes cmap = find_table(file, "cmap");
	f.has_cmap_ = cmap.has(0, 4);
	if (cmap.has(0, 4)) {
		const uint16_t n = cmap.u16(2);
		for (uint16_t i = 0; i < n; i++) {
			const size_t rec = 4 + (size_t)i * 8;
			if (!cmap.has(rec, 8))
				break;
			CmapSubtable st;
			st.platform = cmap.u16(rec);
			st.encoding = cmap.u16(rec + 2);
			st.data = cmap.from(cmap.u32(rec + 4));
			if (st.data.has(0, 2))
				st.format = st.data.u16(0);
			f.cmap_.push_back(st);
		}
	}
	return f;
}

// ---- name ------------------------------------------------------------------------------
std::vector<std::pair<uint16_t, std::string>> Face::names() const
{
	std::vector<std::pair<uint16_t, std::string>> out;
	if (!name_.has(0, 6))
		return out;
	const uint16_t version = name_.u16(0), count = name_.u16(2);
	const size_t storage_off = name_.u16(4);
	if (version > 1 || !name_.has(6, (size_t)count * 12))
		return out;
	const Bytes storage = name_.from(std::max(storage_off, (size_t)6 + (size_t)count * 12));
	for (uint16_t i = 0; i < count; i++) {
		const size_t rec = 6 + (size_t)i * 12;
		const uint16_t platform = name_.u16(rec), encoding = name_.u16(rec + 2), name_id = name_.u16(rec + 6);
		const size_t len = name_.u16(rec + 8), off = name_.u16(rec + 10);
		if (!storage.has(off, len))
			break; // the crate's iterator ends at the first record whose bytes are out of range
		std::string text;
		const bool unicode = platform == 0 || (platform == 3 && (encoding == 0 || encoding == 1));
		if (unicode) { // String::from_utf16 over the big-endian units; an unpaired surrogate -> None -> ""
			bool ok = true;
			for (size_t k = 0; k + 1 < len && ok; k += 2) {
				uint32_t c = storage.u16(off + k);
				if (c >= 0xD800 && c <= 0xDBFF) {
					const uint32_t lo = (k + 3 < len) ? storage.u16(off + k + 2) : 0;
					if (lo >= 0xDC00 && lo <= 0xDFFF) {
						c = 0x10000 + ((c - 0xD800) << 10) + (lo - 0xDC00);
						k += 2;
					} else {
						ok = false;
					}
				} else if (c >= 0xDC00 && c <= 0xDFFF) {
					ok = false;
				}
				if (!ok)
					break;
				if (c < 0x80) {
					text.push_back((char)c);
				} else if (c < 0x800) {
					text.push_back((char)(0xC0 | (c >> 6)));
					text.push_back((char)(0x80 | (c & 63)));
				} else if (c < 0x10000) {
					text.push_back((char)(0xE0 | (c >> 12)));
					text.push_back((char)(0x80 | ((c >> 6) & 63)));
					text.push_back((char)(0x80 | (c & 63)));
				} else {
					text.push_back((char)(0xF0 | (c >> 18)));
					text.push_back((char)(0x80 | ((c >> 12) & 63)));
					text.push_back((char)(0x80 | ((c >> 6) & 63)));
					text.push_back((char)(0x80 | (c & 63)));
				}
			}
			if (!ok)
				text.clear();
		}
		out.emplace_back(name_id, std::move(text));
	}
	return out;
}

// ---- hmtx ------------------------------------------------------------------------------
std::optional<uint16_t> Face::glyph_hor_advance(uint16_t gid) const
{
	if (hmtx_.empty() || num_hmetrics_ == 0 || num_glyphs_ == 0 || gid >= num_glyphs_)
		return std::nullopt;
	if (!hmtx_.has(0, (size_t)num_hmetrics_ * 4))
		return std::nullopt;
	// fewer long metrics than glyphs: the last advance applies to the rest
	const size_t i = std::min<size_t>(gid, (size_t)num_hmetrics_ - 1);
	return hmtx_.u16(i * 4);
}

// ---- cmap ------------------------------------------------------------------------------
bool Face::CmapSubtable::is_unicode() const
{
	switch (platform) {
	case 0:
		return true;
	case 3:
		if (encoding == 1)
			return true;
		return encoding == 10 && (format == 12 || format == 13);
	default:
		return false;
	}
}

namespace {

struct Format4 {
	Bytes d;
	size_t segs = 0, ends = 0, starts = 0, deltas = 0, offsets = 0;
	bool ok = false;
	explicit Format4(Bytes data) : d(data)
	{
		if (!d.has(0, 14))
			return;
		const uint16_t x2 = d.u16(6);
		if (x2 < 2)
			return;
		segs = x2 / 2;
		ends = 14;
		starts = ends + segs * 2 + 2;
		deltas = starts + segs * 2;
		offsets = deltas + segs * 2;
		ok = d.has(offsets, segs * 2);
	}
};

} // namespace

std::optional<uint16_t> Face::CmapSubtable::glyph_index(uint32_t cp) const
{
	switch (format) {
	case 0: {
		if (cp >= 256 || !data.has(6, 256))
			return std::nullopt;
		const uint8_t g = data.u8(6 + cp);
		if (g == 0)
			return std::nullopt;
		return g;
	}
	case 4: {
		const Format4 t(data);
		if (!t.ok || cp > 0xFFFF)
			return std::nullopt;
		const uint16_t c = (uint16_t)cp;
		size_t lo = 0, hi = t.segs;
		while (lo < hi) { // the crate's own bisection over endCode[]
			const size_t mid = (lo + hi) / 2;
			if (data.u16(t.ends + mid * 2) < c) {
				lo = mid + 1;
				continue;
			}
			const uint16_t first = data.u16(t.starts + mid * 2);
			if (first > c) {
				hi = mid;
				continue;
			}
			const uint16_t range_off = data.u16(t.offsets + mid * 2);
			const uint16_t delta = data.u16(t.deltas + mid * 2);
			if (range_off == 0)
				return (uint16_t)(c + delta);
			if (range_off == 0xFFFF)
				return std::nullopt;
			const uint32_t twice = ((uint32_t)c - first) * 2;
			if (twice > 0xFFFF)
				return std::nullopt;
			// u16 wrapping position arithmetic, relative to the subtable start
			const uint16_t pos = (uint16_t)((uint16_t)(t.offsets + mid * 2) + (uint16_t)twice + range_off);
			if (!data.has(pos, 2))
				return std::nullopt;
			const uint16_t raw = data.u16(pos);
			if (raw == 0)
				return std::nullopt;
			const int16_t id = (int16_t)(uint16_t)(raw + delta);
			if (id < 0)
				return std::nullopt;
			return (uint16_t)id;
		}
		return std::nullopt;
	}
	case 6: {
		if (cp > 0xFFFF || !data.has(0, 10))
			return std::nullopt;
		const uint32_t first = data.u16(6), count = data.u16(8);
		if (cp < first || cp - first >= count || !data.has(10 + (size_t)(cp - first) * 2, 2))
			return std::nullopt;
		return data.u16(10 + (size_t)(cp - first) * 2);
	}
	case 10: {
		if (!data.has(0, 20))
			return std::nullopt;
		const uint32_t first = data.u32(12), count = data.u32(16);
		if (cp < first || cp - first >= count || !data.has(20 + (size_t)(cp - first) * 2, 2))
			return std::nullopt;
		return data.u16(20 + (size_t)(cp - first) * 2);
	}
	case 12:
	case 13: {
		if (!data.has(0, 16))
			return std::nullopt;
		const uint32_t n = data.u32(12);
		if (!data.has(16, (size_t)n * 12))
			return std::nullopt;
		size_t lo = 0, hi = n;
		while (lo < hi) {
			const size_t mid = lo + (hi - lo) / 2, g = 16 + mid * 12;
			if (data.u32(g) > cp)
				hi = mid;
			else if (data.u32(g + 4) < cp)
				lo = mid + 1;
			else {
				uint64_t id = data.u32(g + 8);
				if (format == 12) {
					id += cp;
					if (id > 0xFFFFFFFFull)
						return std::nullopt;
					id -= data.u32(g);
				}
				if (id > 0xFFFF)
					return std::nullopt;
				return (uint16_t)id;
			}
		}
		return std::nullopt;
	}
	default: // 2, 8, 14: not produced by any fixture; treated as "no mapping"
		return std::nullopt;
	}
}

template <class F> void Face::CmapSubtable::for_each_codepoint(F &&f) const
{
	switch (format) {
	case 0:
		if (data.has(6, 256))
			for (uint32_t c = 0; c < 256; c++)
				if (data.u8(6 + c) != 0)
					f(c);
		break;
	case 4: {
		const Format4 t(data);
		if (!t.ok)
			break;
		for (size_t s = 0; s < t.segs; s++) {
			const uint32_t a = data.u16(t.starts + s * 2), b = data.u16(t.ends + s * 2);
			if (a == 0xFFFF && b == 0xFFFF)
				break;
			for (uint32_t c = a; c <= b; c++)
				f(c);
		}
		break;
	}
	case 6:
		if (data.has(0, 10)) {
			const uint32_t first = data.u16(6), count = data.u16(8);
			for (uint32_t k = 0; k < count && first + k <= 0xFFFF; k++)
				f(first + k);
		}
		break;
	case 10:
		if (data.has(0, 20)) {
			const uint32_t first = data.u32(12), count = data.u32(16);
			for (uint32_t k = 0; k < count && first + k >= first; k++)
				f(first + k);
		}
		break;
	case 12:
	case 13:
		if (data.has(0, 16)) {
			const uint32_t n = data.u32(12);
			if (!data.has(16, (size_t)n * 12))
				break;
			for (uint32_t k = 0; k < n; k++) {
				const size_t g = 16 + (size_t)k * 12;
				for (uint64_t c = data.u32(g), e = data.u32(g + 4); c <= e; c++)
					f((uint32_t)c);
			}
		}
		break;
	default:
		break;
	}
}

std::optional<uint16_t> Face::glyph_index(uint32_t cp) const
{
	for (const CmapSubtable &st : cmap_) {
		if (st.format == 0xFFFF || !st.is_unicode())
			continue;
		if (auto g = st.glyph_index(cp))
			return g;
	}
	return std::nullopt;
}

std::vector<uint32_t> Face::unicode_codepoints() const
{
	std::vector<uint32_t> cps;
	for (const CmapSubtable &st : cmap_) {
		if (st.format == 0xFFFF || !st.is_unicode())
			continue;
		st.for_each_codepoint([&](uint32_t c) {
			if (st.glyph_index(c)) // metadata.rs:111-113
				cps.push_back(c);
		});
	}
	std::sort(cps.begin(), cps.end());
	cps.erase(std::unique(cps.begin(), cps.end()), cps.end());
	return cps;
}

// ---- glyf ------------------------------------------------------------------------------
std::optional<Bytes> Face::glyph_data(uint16_t gid) const
{
	if (loca_.empty() || glyf_.empty() || gid == 0xFFFF || (size_t)gid + 1 >= loca_entries_)
		return std::nullopt;
	size_t a, b;
	if (loca_long_) {
		a = loca_.u32((size_t)gid * 4);
		b = loca_.u32((size_t)gid * 4 + 4);
	} else {
		a = (size_t)loca_.u16((size_t)gid * 2) * 2;
		b = (size_t)loca_.u16((size_t)gid * 2 + 2) * 2;
	}
	if (a >= b || b > glyf_.size())
		return std::nullopt;
	return glyf_.sub(a, b - a);
}

namespace {

// 2x3 affine of a composite component; f32 like ttf-parser's Transform
struct Affine {
	float a = 1.f, b = 0.f, c = 0.f, d = 1.f, e = 0.f, f = 0.f;
	bool identity() const { return a == 1.f && b == 0.f && c == 0.f && d == 1.f && e == 0.f && f == 0.f; }
	// parent.then(child): Transform::combine(parent, child)
	Affine then(const Affine &k) const
	{
		Affine r;
		r.a = a * k.a + c * k.b;
		r.b = b * k.a + d * k.b;
		r.c = a * k.c + c * k.d;
		r.d = b * k.c + d * k.d;
		r.e = a * k.e + c * k.f + e;
		r.f = b * k.e + d * k.f + f;
		return r;
	}
	void map(float &x, float &y) const
	{
		const float tx = x, ty = y;
		x = a * tx + c * ty + e;
		y = b * tx + d * ty + f;
	}
};

// Turns TrueType contour points into move/line/quad callbacks (ttf-parser glyf.rs Builder).  B: the sink — the
// OutlineBuilder interface (virtual calls), or a concrete recorder whose calls inline (the recording path of the device
// front-end spends as long in the sink as in the walk: 0.8 of 1.75 us per glyph).
template <class B> class ContourEmitter {
public:
	ContourEmitter(B &out, const Affine &t) : out_(out), t_(t), plain_(t.identity()) {}
	// a simple glyph of n_points in n_contours is about to be walked / has been walked: at most one command per point
	// plus, per contour, its close and up to two closing curves (a quad carries 4 floats)
	// (a sink with kRawCursor hands out two store cursors for that many commands; they live HERE, in an object that
	// never leaves walk()'s frame: byte stores through a cursor kept in the sink — which the recursive walk holds by
	// reference — would force the compiler to reload both cursors after every store)
	void begin(uint32_t n_points, uint32_t n_contours)
	{
		if constexpr (B::kRawCursor)
			out_.open((size_t)n_points + 3u * n_contours, 4u * ((size_t)n_points + 3u * n_contours), kp_, cp_);
	}
	void end()
	{
		if constexpr (B::kRawCursor)
			out_.shut(kp_, cp_);
	}

	void point(float x, float y, bool on_curve, bool last_of_contour)
	{
		if (!start_) {
			if (on_curve) {
				start_ = P{x, y};
				move(x, y);
			} else if (lead_off_) {
				const P m = mid(*lead_off_, P{x, y});
				start_ = m;
				pending_off_ = P{x, y};
				move(m.x, m.y);
			} else {
				lead_off_ = P{x, y};
			}
		} else if (pending_off_) {
			const P c = *pending_off_;
			if (on_curve) {
				pending_off_.reset();
				quad(c, P{x, y});
			} else {
				pending_off_ = P{x, y};
				quad(c, mid(c, P{x, y}));
			}
		} else if (on_curve) {
			line(x, y);
		} else {
			pending_off_ = P{x, y};
		}
		if (last_of_contour)
			finish();
	}

private:
	struct P {
		float x, y;
	};
	static P mid(P a, P b) { return P{a.x + 0.5f * (b.x - a.x), a.y + 0.5f * (b.y - a.y)}; } // lerp(.., 0.5)

	void finish()
	{
		if (lead_off_ && pending_off_) {
			const P c = *pending_off_;
			pending_off_.reset();
			quad(c, mid(c, *lead_off_));
		}
		if (start_ && lead_off_)
			quad(*lead_off_, *start_);
		else if (start_ && pending_off_)
			quad(*pending_off_, *start_);
		else if (start_)
			line(start_->x, start_->y);
		start_.reset();
		lead_off_.reset();
		pending_off_.reset();
		if constexpr (B::kRawCursor)
			*kp_++ = 4;
		else
			out_.close();
	}
	void move(float x, float y)
	{
		if (!plain_)
			t_.map(x, y);
		if constexpr (B::kRawCursor) {
			*kp_++ = 0;
			cp_[0] = x, cp_[1] = y;
			cp_ += 2;
		} else {
			out_.move_to(x, y);
		}
	}
	void line(float x, float y)
	{
		if (!plain_)
			t_.map(x, y);
		if constexpr (B::kRawCursor) {
			*kp_++ = 1;
			cp_[0] = x, cp_[1] = y;
			cp_ += 2;
		} else {
			out_.line_to(x, y);
		}
	}
	void quad(P c, P p)
	{
		if (!plain_) {
			t_.map(c.x, c.y);
			t_.map(p.x, p.y);
		}
		if constexpr (B::kRawCursor) {
			*kp_++ = 2;
			cp_[0] = c.x, cp_[1] = c.y, cp_[2] = p.x, cp_[3] = p.y;
			cp_ += 4;
		} else {
			out_.quad_to(c.x, c.y, p.x, p.y);
		}
	}

	B &out_;
	Affine t_;
	bool plain_;
	std::optional<P> start_, lead_off_, pending_off_;
	uint8_t *kp_ = nullptr; // kRawCursor: where the next kind byte / the next coordinates go
	float *cp_ = nullptr;
};

constexpr int kMaxComponentDepth = 32;

enum : uint8_t { ON_CURVE = 0x01, X_SHORT = 0x02, Y_SHORT = 0x04, REPEAT = 0x08, X_SAME_POS = 0x10, Y_SAME_POS = 0x20 };

// simple glyph body (after numberOfContours + bbox); false = malformed (ttf-parser -> None)
template <class B> bool walk_simple(Bytes body, uint16_t n_contours, ContourEmitter<B> &em)
{
	if (!body.has(0, (size_t)n_contours * 2))
		return false;
	const uint16_t last_end = body.u16((size_t)(n_contours - 1) * 2);
	if (last_end == 0xFFFF)
		return false;
	const uint32_t n_points = (uint32_t)last_end + 1;
	if (n_points == 1)
		return true; // a lone point yields nothing
	size_t cur = (size_t)n_contours * 2;
	if (!body.has(cur, 2))
		return false;
	cur += 2 + body.u16(cur); // instructions
	if (cur > body.size())
		return false;
	// pass 1: size of the flag / x / y arrays
	const size_t flags_at = cur;
	size_t xs = 0, ys = 0;
	for (uint32_t left = n_points; left;) {
		if (!body.has(cur, 1))
			return false;
		const uint8_t fl = body.u8(cur++);
		uint32_t run = 1;
		if (fl & REPEAT) {
			if (!body.has(cur, 1))
				return false;
			run += body.u8(cur++);
		}
		if (run > left)
			return false;
		xs += (fl & X_SHORT) ? run : ((fl & X_SAME_POS) ? 0 : 2 * run);
		ys += (fl & Y_SHORT) ? run : ((fl & Y_SAME_POS) ? 0 : 2 * run);
		left -= run;
	}
	const size_t x_at = cur, y_at = x_at + xs, y_end = y_at + ys;
	if (y_end > body.size())
		return false;
	// pass 2: decode
	size_t fp = flags_at, xp = x_at, yp = y_at;
	uint8_t fl = 0;
	uint32_t run_left = 0;
	int16_t x = 0, y = 0;
	uint32_t contour = 1, in_contour_left = body.u16(0); // EndpointsIter
	em.begin(n_points, n_contours);
	for (uint32_t i = 0; i < n_points; i++) {
		bool last;
		if (in_contour_left == 0) {
			if (contour < n_contours) {
				const uint16_t end = body.u16((size_t)contour * 2), prev = body.u16((size_t)(contour - 1) * 2);
				const uint16_t span = end > prev ? (uint16_t)(end - prev) : 0;
				in_contour_left = span ? span - 1u : 0u;
			}
			contour++;
			last = true;
		} else {
			in_contour_left--;
			last = false;
		}
		if (run_left == 0) {
			fl = fp < x_at ? body.u8(fp++) : 0;
			if (fl & REPEAT)
				run_left = fp < x_at ? body.u8(fp++) : 0;
		} else {
			run_left--;
		}
		int16_t dx = 0, dy = 0;
		if (fl & X_SHORT) {
			const int v = xp < y_at ? body.u8(xp++) : 0;
			dx = (int16_t)((fl & X_SAME_POS) ? v : -v);
		} else if (!(fl & X_SAME_POS) && xp + 2 <= y_at) {
			dx = body.i16(xp);
			xp += 2;
		}
		if (fl & Y_SHORT) {
			const int v = yp < y_end ? body.u8(yp++) : 0;
			dy = (int16_t)((fl & Y_SAME_POS) ? v : -v);
		} else if (!(fl & Y_SAME_POS) && yp + 2 <= y_end) {
			dy = body.i16(yp);
			yp += 2;
		}
		x = (int16_t)(uint16_t)((uint16_t)x + (uint16_t)dx); // wrapping
		y = (int16_t)(uint16_t)((uint16_t)y + (uint16_t)dy);
		em.point((float)x, (float)y, (fl & ON_CURVE) != 0, last);
	}
	em.end();
	return true;
}

} // namespace

// PARTS: the sink does not take callbacks but every simple glyph as it stands (B::part): the device decodes it
template <class B, bool PARTS> struct GlyfWalker {
	const Face &face;
	B &out;

	bool walk(Bytes glyph, int depth, const Affine &t)
	{
		if (depth >= kMaxComponentDepth || !glyph.has(0, 2))
			return false;
		const int16_t n_contours = glyph.i16(0);
		const Bytes body = glyph.from(10);
		if (n_contours > 0) {
			if (glyph.size() < 10)
				return false;
			if constexpr (PARTS) {
				return out.part(body, (uint16_t)n_contours, t.a, t.b, t.c, t.d, t.e, t.f, t.identity());
			} else {
				ContourEmitter<B> em(out, t);
				return walk_simple(body, (uint16_t)n_contours, em);
			}
		}
		if (n_contours == 0 || glyph.size() < 10)
			return n_contours == 0;
		// composite
		size_t p = 0;
		for (;;) {
			if (!body.has(p, 4))
				break;
			const uint16_t flags = body.u16(p), child = body.u16(p + 2);
			p += 4;
			Affine k;
			if (flags & 0x0002) { // ARGS_ARE_XY_VALUES
				if (flags & 0x0001) {
					if (!body.has(p, 4))
						break;
					k.e = (float)body.i16(p);
					k.f = (float)body.i16(p + 2);
					p += 4;
				} else {
					if (!body.has(p, 2))
						break;
					k.e = (float)(int8_t)body.u8(p);
					k.f = (float)(int8_t)body.u8(p + 1);
					p += 2;
				}
			} // anchor-point arguments are not consumed by ttf-parser 0.25 (parity unpinned)
			auto f2dot14 = [&](size_t at) { return (float)body.i16(at) / 16384.0f; };
			if (flags & 0x0080) {
				if (!body.has(p, 8))
					break;
				k.a = f2dot14(p);
				k.b = f2dot14(p + 2);
				k.c = f2dot14(p + 4);
				k.d = f2dot14(p + 6);
				p += 8;
			} else if (flags & 0x0040) {
				if (!body.has(p, 4))
					break;
				k.a = f2dot14(p);
				k.d = f2dot14(p + 2);
				p += 4;
			} else if (flags & 0x0008) {
				if (!body.has(p, 2))
					break;
				k.a = k.d = f2dot14(p);
				p += 2;
			}
			if (auto cd = face.glyph_data(child))
				if (!walk(*cd, depth + 1, t.then(k)))
					return false;
			if (!(flags & 0x0020)) // MORE_COMPONENTS
				break;
		}
		return true;
	}
};

bool Face::outline_glyph(uint16_t gid, OutlineBuilder &builder) const
{
	if (!has_glyf_outlines() && cff_) // ttf-parser: glyf first, then cff
		return cff_->outline(gid, builder);
	const auto g = glyph_data(gid);
	if (!g)
		return false;
	GlyfWalker<OutlineBuilder, false> w{*this, builder};
	return w.walk(*g, 0, Affine{});
}

namespace {
// the callbacks in the compact upload form of vgsdf_outlines_packed: a kind byte per command + the coordinates it carries
struct PackedSink {
	std::vector<uint8_t> &kinds;
	std::vector<float> &coords;
	void move_to(float x, float y)
	{
		kinds.push_back(0);
		coords.push_back(x);
		coords.push_back(y);
	}
	void line_to(float x, float y)
	{
		kinds.push_back(1);
		coords.push_back(x);
		coords.push_back(y);
	}
	void quad_to(float x1, float y1, float x, float y)
	{
		kinds.push_back(2);
		coords.push_back(x1);
		coords.push_back(y1);
		coords.push_back(x);
		coords.push_back(y);
	}
	void close() { kinds.push_back(4); }
};
// The glyf walk's recorder: it announces every simple glyph's size first, room for its commands is made once (open),
// the emitter stores kind bytes and coordinates through two cursors of its own instead of three to five push_backs per
// command, shut() trims to what was written (2973 glyphs of Noto Sans: 30 -> 16 ns per command).
struct PackedCursorSink {
	static constexpr bool kRawCursor = true;
	std::vector<uint8_t> &kinds;
	std::vector<float> &coords;
	void open(size_t cmds, size_t floats, uint8_t *&kp, float *&cp)
	{
		const size_t k0 = kinds.size(), c0 = coords.size();
		// (batches are addressed with 32-bit offsets: a composite that fans one glyph out to gigabytes of commands is an error,
		// not a wrapped offset)
		if (k0 + cmds > (1ull << 30) || c0 + floats > (1ull << 30))
			throw std::length_error("glyph outlines: more than 2^30 commands in one batch (composite fan-out)");
		kinds.resize(k0 + cmds);
		coords.resize(c0 + floats);
		kp = kinds.data() + k0;
		cp = coords.data() + c0;
	}
	void shut(const uint8_t *kp, const float *cp)
	{
		kinds.resize((size_t)(kp - kinds.data()));
		coords.resize((size_t)(cp - coords.data()));
	}
};
// (CFF charstrings go through the OutlineBuilder interface)
struct PackedSinkVirtual final : OutlineBuilder {
	PackedSink s;
	explicit PackedSinkVirtual(PackedSink p) : s(p) {}
	void move_to(float x, float y) override { s.move_to(x, y); }
	void line_to(float x, float y) override { s.line_to(x, y); }
	void quad_to(float x1, float y1, float x, float y) override { s.quad_to(x1, y1, x, y); }
	void curve_to(float x1, float y1, float x2, float y2, float x, float y) override
	{
		s.kinds.push_back(3);
		for (float v : {x1, y1, x2, y2, x, y})
			s.coords.push_back(v);
	}
	void close() override { s.close(); }
};
} // namespace

namespace {
// Sink of the parts walk: what walk_simple checks BEFORE it touches the flag / coordinate arrays is checked here, with
// the same outcome (false: ttf-parser returns None, the walk of a composite stops); the arrays themselves are copied
// as they stand and checked where they are decoded (a mismatch there fails the batch: vgsdf.h, VGSDF_E_GLYF).
struct PartsSink {
	std::vector<GlyfPart> &parts;
	std::vector<uint8_t> &bytes;
	uint32_t &slots;
	bool *overflow;
	// A batch of parts is addressed with 32-bit offsets (vgsdf_glyf_part); a composite copies its simple glyphs once per
	// LEAF of its tree, so a small font can ask for gigabytes here (one 32 KB glyph named by 140 000 components).  Past
	// these bounds nothing more is recorded and the walk stops: the caller drops the glyf form for the batch.
	static constexpr uint64_t kMaxBatchSlots = 1ull << 26, kMaxBatchParts = 1ull << 22;
	static uint64_t max_batch_bytes()
	{
		static const uint64_t v = [] {
			const char *e = std::getenv("VG_GLYF_PARTS_LIMIT"); // (test switch: a small bound sends ordinary fonts down the fallback)
			return e ? std::max<uint64_t>(1024, std::strtoull(e, nullptr, 10)) : (1ull << 26);
		}();
		return v;
	}
	bool part(Bytes body, uint16_t n_contours, float a, float b, float c, float d, float e, float f, bool plain)
	{
		if (!body.has(0, (size_t)n_contours * 2))
			return false;
		const uint16_t last_end = body.u16((size_t)(n_contours - 1) * 2);
		if (last_end == 0xFFFF)
			return false;
		const uint32_t n_points = (uint32_t)last_end + 1;
		if (n_points == 1)
			return true; // a lone point yields nothing
		size_t cur = (size_t)n_contours * 2;
		if (!body.has(cur, 2))
			return false;
		cur += 2 + body.u16(cur); // instructions: not needed on the device
		if (cur > body.size())
			return false;
		// (an entry longer than the device's decoder takes is not copied — one damaged `loca` entry could otherwise make every
		// glyph that names it carry megabytes: without its arrays the part fails there and the batch goes to the host's reader)
		constexpr size_t kMaxEntry = 32 * 1024;
		const bool fits = (size_t)n_contours * 2 + (body.size() - cur) <= kMaxEntry;
		const size_t ends = fits ? (size_t)n_contours * 2 : 0, arrays = fits ? body.size() - cur : 0;
		if ((uint64_t)bytes.size() + ends + arrays + 4 > max_batch_bytes() || (uint64_t)slots + n_points + 2ull * n_contours > kMaxBatchSlots ||
		    parts.size() >= kMaxBatchParts) {
			if (overflow)
				*overflow = true;
			return false;
		}
		GlyfPart p;
		p.byte_off = (uint32_t)bytes.size(); // (a multiple of 4: padded below)
		p.byte_len = (uint32_t)(ends + arrays);
		p.cmd_at = slots;
		// callbacks of a contour of L points: at most one per point, at most two from Builder::finish (the closing curves of a
		// contour that begins AND ends off the curve — whose first point then emitted nothing) and close(): L + 2 in every case
		p.cmd_cap = n_points + 2u * n_contours;
		p.n_contours = n_contours;
		p.plain = plain ? 1u : 0u;
		p.a = a, p.b = b, p.c = c, p.d = d, p.e = e, p.f = f;
		const size_t at = bytes.size(), padded = ((size_t)p.byte_len + 3) & ~(size_t)3;
		bytes.resize(at + padded); // (zero padding)
		if (fits) {
			std::memcpy(bytes.data() + at, body.data(), ends);
			std::memcpy(bytes.data() + at + ends, body.data() + cur, arrays);
		}
		slots += p.cmd_cap;
		parts.push_back(p);
		return true;
	}
};
} // namespace

bool Face::glyph_parts(uint16_t gid, std::vector<GlyfPart> &parts, std::vector<uint8_t> &bytes, uint32_t &slots, bool *overflow) const
{
	const auto g = glyph_data(gid);
	if (!g)
		return false;
	PartsSink sink{parts, bytes, slots, overflow};
	GlyfWalker<PartsSink, true> w{*this, sink};
	return w.walk(*g, 0, Affine{});
}

bool Face::outline_glyph_packed(uint16_t gid, std::vector<uint8_t> &kinds, std::vector<float> &coords) const
{
	PackedSink sink{kinds, coords};
	if (!has_glyf_outlines() && cff_) {
		PackedSinkVirtual v(sink);
		return cff_->outline(gid, v);
	}
	const auto g = glyph_data(gid);
	if (!g)
		return false;
	PackedCursorSink cursor{kinds, coords};
	GlyfWalker<PackedCursorSink, false> w{*this, cursor};
	return w.walk(*g, 0, Affine{});
}

} // namespace vg
