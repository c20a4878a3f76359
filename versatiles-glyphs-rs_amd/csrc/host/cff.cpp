// cff.cpp — see cff.hpp.
#include "cff.hpp"

#include <cmath>
#include <limits>

namespace vg {

namespace {

constexpr int kMaxOperands = 48;   // ttf-parser: MAX_ARGUMENTS_STACK_LEN of cff1
constexpr int kMaxOperands2 = 513; // ... of cff2
constexpr size_t kMaxRegions = 64; // ttf-parser: scalars of one ItemVariationData (cff2)
constexpr int kMaxDepth = 10;    // ttf-parser: STACK_LIMIT (nested subroutine calls)

// DICT data (Technical Note #5176, section 4): operands followed by an operator
struct DictReader {
	explicit DictReader(Bytes dict, bool cff2_dict = false) : d(dict), cff2(cff2_dict) {}
	Bytes d;
	bool cff2; // CFF2 DICTs: 22 (vsindex), 23 (blend), 24 (vstore) are operators; so is every other non-number byte
	size_t pos = 0;
	std::vector<double> operands;
	// next operator (two-byte operators as 1200 + second byte), -1 at the end, -2 on malformed data
	int next()
	{
		operands.clear();
		while (pos < d.size()) {
			const uint8_t b = d.u8(pos);
			if (b <= 21 || (cff2 && (b <= 27 || b == 31 || b == 255))) {
				pos++;
				if (b != 12)
					return b;
				if (pos >= d.size())
					return -2;
				return 1200 + d.u8(pos++);
			}
			if (b == 28) {
				if (!d.has(pos, 3))
					return -2;
				operands.push_back((double)d.i16(pos + 1));
				pos += 3;
			} else if (b == 29) {
				if (!d.has(pos, 5))
					return -2;
				operands.push_back((double)(int32_t)d.u32(pos + 1));
				pos += 5;
			} else if (b == 30) { // real number: nibbles up to 0xf; only skipped (no operator of interest takes one)
				pos++;
				bool done = false;
				while (!done) {
					if (pos >= d.size())
						return -2;
					const uint8_t v = d.u8(pos++);
					done = (v >> 4) == 0xF || (v & 0xF) == 0xF;
				}
				operands.push_back(0.0);
			} else if (b >= 32 && b <= 246) {
				operands.push_back((double)((int)b - 139));
				pos++;
			} else if (b >= 247 && b <= 250) {
				if (!d.has(pos, 2))
					return -2;
				operands.push_back((double)(((int)b - 247) * 256 + d.u8(pos + 1) + 108));
				pos += 2;
			} else if (b >= 251 && b <= 254) {
				if (!d.has(pos, 2))
					return -2;
				operands.push_back((double)(-((int)b - 251) * 256 - d.u8(pos + 1) - 108));
				pos += 2;
			} else {
				return -2; // 22-27, 31, 255: reserved
			}
			if (operands.size() > 513)
				return -2;
		}
		return -1;
	}
};

bool to_offset(double v, size_t &out)
{
	if (!(v >= 0.0 && v <= 4294967295.0))
		return false;
	out = (size_t)v;
	return true;
}

} // namespace

bool CffTable::parse_index(Bytes table, size_t at, Index &out, size_t &end, bool count32)
{
	out = Index{};
	const size_t cw = count32 ? 4 : 2; // CFF2: card32 counts
	if (!table.has(at, cw))
		return false;
	const uint32_t count = count32 ? table.u32(at) : table.u16(at);
	if (count == 0 || count == 0xFFFFFFFFu) {
		end = at + cw;
		return true;
	}
	if (!table.has(at, cw + 1))
		return false;
	const uint8_t off_size = table.u8(at + cw);
	if (off_size < 1 || off_size > 4)
		return false;
	const size_t offsets_at = at + cw + 1, offsets_len = ((size_t)count + 1) * off_size;
	if (!table.has(offsets_at, offsets_len))
		return false;
	auto off = [&](uint32_t i) {
		uint32_t v = 0;
		for (uint8_t k = 0; k < off_size; k++)
			v = (v << 8) | table.u8(offsets_at + (size_t)i * off_size + k);
		return v;
	};
	const uint32_t last = off(count);
	if (last < 1)
		return false;
	const size_t data_at = offsets_at + offsets_len; // offsets are relative to the byte before the data
	if (!table.has(data_at, (size_t)last - 1))
		return false;
	out.data = table;
	out.count = count;
	out.off_size = off_size;
	out.offsets_at = offsets_at;
	out.data_at = data_at;
	end = data_at + last - 1;
	return true;
}

std::optional<Bytes> CffTable::Index::get(uint32_t i) const
{
	if (i >= count)
		return std::nullopt;
	auto off = [&](uint32_t k) {
		uint32_t v = 0;
		for (uint8_t b = 0; b < off_size; b++)
			v = (v << 8) | data.u8(offsets_at + (size_t)k * off_size + b);
		return v;
	};
	const uint32_t a = off(i), b = off(i + 1);
	if (a < 1 || b < a)
		return std::nullopt;
	if (!data.has(data_at + a - 1, b - a))
		return std::nullopt;
	return data.sub(data_at + a - 1, b - a);
}

bool CffTable::parse_private(Bytes table, size_t offset, size_t size, PrivateDict &out)
{
	out = PrivateDict{};
	if (!table.has(offset, size))
		return false;
	DictReader r(table.sub(offset, size));
	for (int op = r.next(); op != -1; op = r.next()) {
		if (op == -2)
			return false;
		if (op == 19 && r.operands.size() == 1) { // Subrs: offset relative to the Private DICT
			size_t rel;
			if (!to_offset(r.operands[0], rel))
				return false;
			size_t end;
			if (!parse_index(table, offset + rel, out.local_subrs, end))
				return false;
		}
	}
	return true;
}

std::optional<CffTable> CffTable::parse(Bytes table)
{
	if (!table.has(0, 4) || table.u8(0) != 1) // major version 1 (CFF2 is another table)
		return std::nullopt;
	CffTable t;
	t.table_ = table;
	size_t pos = table.u8(2); // hdrSize
	Index names, top_dicts, strings;
	if (!parse_index(table, pos, names, pos) || !parse_index(table, pos, top_dicts, pos) || !parse_index(table, pos, strings, pos) ||
	    !parse_index(table, pos, t.global_subrs_, pos))
		return std::nullopt;
	const auto top = top_dicts.get(0);
	if (!top)
		return std::nullopt;
	size_t charstrings_at = 0, private_size = 0, private_at = 0, fd_array_at = 0, fd_select_at = 0;
	bool have_private = false;
	DictReader r(*top);
	for (int op = r.next(); op != -1; op = r.next()) {
		if (op == -2)
			return std::nullopt;
		const auto &v = r.operands;
		if (op == 15 && v.size() == 1) {
			if (!to_offset(v[0], t.charset_at_))
				return std::nullopt;
		} else if (op == 17 && v.size() == 1) {
			if (!to_offset(v[0], charstrings_at))
				return std::nullopt;
		} else if (op == 18 && v.size() == 2) {
			if (!to_offset(v[0], private_size) || !to_offset(v[1], private_at))
				return std::nullopt;
			have_private = true;
		} else if (op == 1206 && v.size() == 1) { // CharstringType
			if (v[0] != 2.0)
				return std::nullopt;
		} else if (op == 1230) { // ROS: CID-keyed
			t.cid_ = true;
		} else if (op == 1236 && v.size() == 1) {
			if (!to_offset(v[0], fd_array_at))
				return std::nullopt;
		} else if (op == 1237 && v.size() == 1) {
			if (!to_offset(v[0], fd_select_at))
				return std::nullopt;
		}
	}
	size_t end;
	if (charstrings_at == 0 || !parse_index(table, charstrings_at, t.charstrings_, end) || t.charstrings_.count == 0)
		return std::nullopt;
	if (t.cid_) {
		Index fd_array;
		if (fd_array_at == 0 || fd_select_at == 0 || !parse_index(table, fd_array_at, fd_array, end) || !table.has(fd_select_at, 1))
			return std::nullopt;
		t.fd_select_ = table.from(fd_select_at);
		for (uint32_t i = 0; i < fd_array.count; i++) {
			PrivateDict pd;
			if (const auto fd = fd_array.get(i)) {
				DictReader fr(*fd);
				for (int op = fr.next(); op != -1 && op != -2; op = fr.next())
					if (op == 18 && fr.operands.size() == 2) {
						size_t sz, at;
						if (to_offset(fr.operands[0], sz) && to_offset(fr.operands[1], at))
							(void)parse_private(table, at, sz, pd);
					}
			}
			t.fd_priv_.push_back(pd);
		}
	} else if (have_private) {
		(void)parse_private(table, private_at, private_size, t.private_); // (a broken Private DICT only costs the local subroutines)
	}
	return t;
}

namespace {

// factor of one axis of a variation region at normalised coordinate 0 (the crate's evaluate_axis with coord = 0; all
// values F2Dot14 as integers).  Only 0 and 1 can come out: the interpolating branches need start < 0 < end with the
// peak off 0, which the second rule has already answered with 1.
float axis_factor_at_default(int start, int peak, int end)
{
	if (start > peak || peak > end)
		return 1.0f;
	if (start < 0 && end > 0 && peak != 0)
		return 1.0f;
	if (peak == 0)
		return 1.0f;
	return 0.0f; // 0 <= start or end <= 0: outside the region (or on its border)
}

} // namespace

// OpenType `CFF2` table, as the crate's cff2::Table::parse reads it: header, Top DICT (CharStrings 17, FDArray 12 36,
// vstore 24), global subroutines behind the Top DICT, ItemVariationStore, the first local subroutines found.
std::optional<CffTable> CffTable::parse2(Bytes table, uint32_t n_coords)
{
	if (!table.has(0, 5) || table.u8(0) != 2)
		return std::nullopt;
	CffTable t;
	t.table_ = table;
	t.cff2_ = true;
	const size_t header_size = table.u8(2), top_len = table.u16(3);
	const size_t top_at = header_size > 5 ? header_size : 5;
	if (!table.has(top_at, top_len))
		return std::nullopt;
	size_t charstrings_at = 0, fd_array_at = 0, vstore_at = 0;
	bool have_fd_array = false, have_vstore = false;
	{
		DictReader r(table.sub(top_at, top_len), true);
		for (int op = r.next(); op != -1; op = r.next()) {
			if (op == -2)
				break; // (the crate stops at a number it cannot read and keeps what it has)
			const auto &v = r.operands;
			if (op == 17) {
				if (v.size() != 1 || !to_offset(v[0], charstrings_at))
					return std::nullopt;
			} else if (op == 1236) {
				have_fd_array = v.size() == 1 && to_offset(v[0], fd_array_at);
			} else if (op == 24) {
				have_vstore = v.size() == 1 && to_offset(v[0], vstore_at);
			}
		}
	}
	if (charstrings_at == 0)
		return std::nullopt;
	size_t end;
	if (!parse_index(table, top_at + top_len, t.global_subrs_, end, true) || !parse_index(table, charstrings_at, t.charstrings_, end, true))
		return std::nullopt;
	if (have_vstore) {
		// u16 length, then the ItemVariationStore: format (1), offset of the region list, offsets of the
		// ItemVariationData subtables — all relative to the store
		if (!table.has(vstore_at, 2))
			return std::nullopt;
		const Bytes st = table.from(vstore_at + 2);
		if (!st.has(0, 8) || st.u16(0) != 1)
			return std::nullopt;
		const size_t regions_at = st.u32(2);
		const uint32_t n_data = st.u16(6);
		if (!st.has(8, (size_t)n_data * 4) || !st.has(regions_at, 4))
			return std::nullopt;
		const uint32_t axis_count = st.u16(regions_at), region_count = st.u16(regions_at + 2);
		const uint32_t n_records = axis_count * region_count;
		if (n_records > 0xFFFFu || !st.has(regions_at + 4, (size_t)n_records * 6))
			return std::nullopt;
		auto region_factor = [&](uint32_t region) {
			float v = 1.0f;
			for (uint32_t i = 0; i < n_coords; i++) {
				// (flat record index in 16 bits, as the crate computes it: a face with more coordinates than the
				// store has axes reads on into the next region's records)
				const uint32_t rec = region * axis_count + i;
				if (region * axis_count > 0xFFFFu || rec > 0xFFFFu || rec >= n_records)
					return 0.0f;
				const size_t at = regions_at + 4 + (size_t)rec * 6;
				const float f = axis_factor_at_default(st.i16(at), st.i16(at + 2), st.i16(at + 4));
				if (f == 0.0f)
					return 0.0f;
				v *= f;
			}
			return v;
		};
		t.blend_sets_.resize(n_data);
		for (uint32_t d = 0; d < n_data; d++) {
			const size_t at = st.u32(8 + (size_t)d * 4);
			if (at > st.size() || !st.has(at + 4, 2))
				continue; // not ok: a charstring that selects it ends there
			const uint32_t n = st.u16(at + 4);
			if (!st.has(at + 6, (size_t)n * 2) || n > kMaxRegions)
				continue;
			BlendSet &set = t.blend_sets_[d];
			set.ok = true;
			for (uint32_t k = 0; k < n; k++)
				set.scalars.push_back(region_factor(st.u16(at + 6 + (size_t)k * 2)));
		}
	}
	if (have_fd_array) {
		Index fd_array;
		if (!parse_index(table, fd_array_at, fd_array, end, true))
			return std::nullopt;
		for (uint32_t i = 0; i < fd_array.count; i++) {
			const auto fd = fd_array.get(i);
			if (!fd)
				continue; // (the crate's iterator would stop here; an INDEX that parsed has no such entry)
			size_t priv_size = 0, priv_at = 0;
			bool have_priv = false;
			DictReader fr(*fd, true);
			for (int op = fr.next(); op != -1 && op != -2; op = fr.next())
				if (op == 18) {
					have_priv = fr.operands.size() == 2 && to_offset(fr.operands[0], priv_size) && to_offset(fr.operands[1], priv_at);
					break;
				}
			if (!have_priv)
				continue;
			if (!table.has(priv_at, priv_size))
				return std::nullopt;
			size_t subrs_rel = 0;
			bool have_subrs = false;
			DictReader pr(table.sub(priv_at, priv_size), true);
			for (int op = pr.next(); op != -1 && op != -2; op = pr.next())
				if (op == 19) {
					have_subrs = pr.operands.size() == 1 && to_offset(pr.operands[0], subrs_rel);
					break;
				}
			if (!have_subrs)
				continue;
			if (!parse_index(table, priv_at + subrs_rel, t.private_.local_subrs, end, true))
				return std::nullopt;
			break; // the first Font DICT with local subroutines serves every glyph
		}
	}
	return t;
}

const CffTable::Index *CffTable::local_subrs_for(uint16_t gid) const
{
	if (!cid_)
		return &private_.local_subrs;
	// FDSelect (Technical Note #5176, section 19): format 0 = one byte per glyph, format 3 = ranges
	uint32_t fd = 0xFFFFFFFFu;
	const Bytes &s = fd_select_;
	if (s.has(0, 1) && s.u8(0) == 0) {
		if (s.has(1 + (size_t)gid, 1))
			fd = s.u8(1 + (size_t)gid);
	} else if (s.has(0, 3) && s.u8(0) == 3) {
		const uint32_t n = s.u16(1);
		for (uint32_t i = 0; i < n; i++) {
			const size_t rec = 3 + (size_t)i * 3;
			if (!s.has(rec, 5))
				break;
			const uint32_t first = s.u16(rec), next = s.u16(rec + 3);
			if (gid >= first && gid < next) {
				fd = s.u8(rec + 2);
				break;
			}
		}
	}
	return fd < fd_priv_.size() ? &fd_priv_[fd].local_subrs : nullptr;
}

// StandardEncoding code -> string id (Technical Note #5176, appendices A and B), then string id -> glyph through
// the charset (section 13: format 0 = one SID per glyph, formats 1 / 2 = ranges with an 8- / 16-bit count).
std::optional<uint16_t> CffTable::standard_code_to_glyph(uint32_t code) const
{
	uint32_t sid = 0;
	if (code >= 32 && code <= 126)
		sid = code - 31;
	else if (code >= 161 && code <= 175)
		sid = code - 65;
	else if (code >= 177 && code <= 180)
		sid = code - 66;
	else if (code >= 182 && code <= 189)
		sid = code - 67;
	else if (code == 191)
		sid = 123;
	else if (code >= 193 && code <= 200)
		sid = code - 69;
	else if (code >= 202 && code <= 203)
		sid = code - 70;
	else if (code >= 205 && code <= 208)
		sid = code - 71;
	else {
		static const uint16_t rest[][2] = {{225, 138}, {227, 139}, {232, 140}, {233, 141}, {234, 142}, {235, 143},
		                                   {241, 144}, {245, 145}, {248, 146}, {249, 147}, {250, 148}, {251, 149}};
		for (const auto &r : rest)
			if (r[0] == code)
				sid = r[1];
	}
	if (sid == 0 || cid_)
		return std::nullopt;
	const uint32_t n = charstrings_.count;
	if (charset_at_ == 0) // ISOAdobe: glyph id = string id
		return sid <= 228 && sid < n ? std::optional<uint16_t>((uint16_t)sid) : std::nullopt;
	if (charset_at_ <= 2) // the Expert charsets hold none of these glyphs under their standard names (as ttf-parser: none)
		return std::nullopt;
	const Bytes c = table_.from(charset_at_);
	if (!c.has(0, 1))
		return std::nullopt;
	const uint8_t format = c.u8(0);
	if (format == 0) {
		for (uint32_t g = 1; g < n; g++) {
			if (!c.has(1 + 2 * (size_t)(g - 1), 2))
				break;
			if (c.u16(1 + 2 * (size_t)(g - 1)) == sid)
				return (uint16_t)g;
		}
		return std::nullopt;
	}
	if (format != 1 && format != 2)
		return std::nullopt;
	const size_t rec = format == 1 ? 3 : 4;
	uint32_t g = 1;
	for (size_t at = 1; g < n; at += rec) {
		if (!c.has(at, rec))
			break;
		const uint32_t first = c.u16(at), left = format == 1 ? c.u8(at + 2) : c.u16(at + 2);
		if (sid >= first && sid <= first + left) {
			const uint32_t hit = g + (sid - first);
			return hit < n ? std::optional<uint16_t>((uint16_t)hit) : std::nullopt;
		}
		g += left + 1;
	}
	return std::nullopt;
}

// bbox of everything handed to the builder (control points included), as ttf-parser's Builder keeps it
struct CffBounds {
	float x0 = std::numeric_limits<float>::max(), y0 = std::numeric_limits<float>::max();
	float x1 = std::numeric_limits<float>::lowest(), y1 = std::numeric_limits<float>::lowest();
};

// One glyph's charstring program (Technical Note #5177).
struct CharStringRun {
	CharStringRun(const CffTable &table, OutlineBuilder &builder, CffBounds &bounds)
	    : t(table), out(builder), bb(bounds), cff2(table.cff2_), cap(table.cff2_ ? kMaxOperands2 : kMaxOperands), have_width(table.cff2_)
	{
	}
	const CffTable &t;
	OutlineBuilder &out;
	CffBounds &bb;
	const bool cff2;
	const int cap; // operand stack limit of the table's version
	const CffTable::Index *local = nullptr;
	float stack[kMaxOperands2];
	int sp = 0;
	float x = 0.0f, y = 0.0f;
	// (CFF2 charstrings carry no width: "already seen" makes every operand count exact)
	bool has_move_to = false, first_move_to = true, have_width, has_endchar = false;
	uint32_t stems = 0;
	// CFF2: factors of the regions of the selected ItemVariationData, `vsindex` / `blend` bookkeeping
	const std::vector<float> *scalars = nullptr;
	bool had_vsindex = false, had_blend = false;
	bool select_blend_set(uint32_t index)
	{
		if (index >= t.blend_sets_.size() || !t.blend_sets_[index].ok)
			return false;
		scalars = &t.blend_sets_[index].scalars;
		return true;
	}
	void extend(float px, float py)
	{
		bb.x0 = px < bb.x0 ? px : bb.x0;
		bb.y0 = py < bb.y0 ? py : bb.y0;
		bb.x1 = px > bb.x1 ? px : bb.x1;
		bb.y1 = py > bb.y1 ? py : bb.y1;
	}
	void move_to(float px, float py)
	{
		extend(px, py);
		out.move_to(px, py);
	}
	void line_to(float px, float py)
	{
		extend(px, py);
		out.line_to(px, py);
	}
	void curve_to(float x1, float y1, float x2, float y2, float px, float py)
	{
		extend(x1, y1);
		extend(x2, y2);
		extend(px, py);
		out.curve_to(x1, y1, x2, y2, px, py);
	}
	bool push(float v)
	{
		if (sp >= cap)
			return false;
		stack[sp++] = v;
		return true;
	}
	static uint32_t bias(uint32_t n) { return n < 1240 ? 107 : (n < 33900 ? 1131 : 32768); }

	bool do_move(int skip, bool hx, bool hy)
	{
		const int want = (hx ? 1 : 0) + (hy ? 1 : 0);
		if (sp != skip + want)
			return false;
		if (first_move_to)
			first_move_to = false;
		else
			out.close();
		has_move_to = true;
		int i = skip;
		if (hx)
			x += stack[i++];
		if (hy)
			y += stack[i++];
		move_to(x, y);
		sp = 0;
		return true;
	}
	bool do_alternating_lines(bool horizontal)
	{
		if (!has_move_to || sp == 0)
			return false;
		for (int i = 0; i < sp; i++) {
			if (horizontal)
				x += stack[i];
			else
				y += stack[i];
			horizontal = !horizontal;
			line_to(x, y);
		}
		sp = 0;
		return true;
	}
	void curve_rel(int i)
	{
		const float x1 = x + stack[i], y1 = y + stack[i + 1];
		const float x2 = x1 + stack[i + 2], y2 = y1 + stack[i + 3];
		x = x2 + stack[i + 4];
		y = y2 + stack[i + 5];
		curve_to(x1, y1, x2, y2, x, y);
	}
	// hvcurveto / vhcurveto: curves that start horizontal and vertical in turn; the last may carry a fifth operand
	bool do_hv_curves(bool horizontal)
	{
		if (!has_move_to || sp < 4)
			return false;
		int i = 0;
		while (i < sp) {
			const int left = sp - i;
			if (left < 4)
				return false;
			const float last = left == 5 ? stack[i + 4] : 0.0f;
			if (horizontal) {
				const float x1 = x + stack[i], y1 = y;
				const float x2 = x1 + stack[i + 1], y2 = y1 + stack[i + 2];
				y = y2 + stack[i + 3];
				x = x2 + last;
				curve_to(x1, y1, x2, y2, x, y);
			} else {
				const float x1 = x, y1 = y + stack[i];
				const float x2 = x1 + stack[i + 1], y2 = y1 + stack[i + 2];
				x = x2 + stack[i + 3];
				y = y2 + last;
				curve_to(x1, y1, x2, y2, x, y);
			}
			i += left == 5 ? 5 : 4;
			horizontal = !horizontal;
		}
		sp = 0;
		return true;
	}

	// false: the glyph has no outline (ttf-parser: Err -> None)
	bool run(Bytes cs, int depth)
	{
		size_t pos = 0;
		while (pos < cs.size()) {
			const uint8_t op = cs.u8(pos++);
			if (op >= 32 || op == 28) { // operands
				float v;
				if (op == 28) {
					if (!cs.has(pos, 2))
						return false;
					v = (float)cs.i16(pos);
					pos += 2;
				} else if (op <= 246) {
					v = (float)((int)op - 139);
				} else if (op <= 250) {
					if (!cs.has(pos, 1))
						return false;
					v = (float)(((int)op - 247) * 256 + cs.u8(pos) + 108);
					pos += 1;
				} else if (op <= 254) {
					if (!cs.has(pos, 1))
						return false;
					v = (float)(-((int)op - 251) * 256 - cs.u8(pos) - 108);
					pos += 1;
				} else { // 255: 16.16 fixed
					if (!cs.has(pos, 4))
						return false;
					v = (float)(int32_t)cs.u32(pos) / 65536.0f;
					pos += 4;
				}
				if (!push(v))
					return false;
				continue;
			}
			switch (op) {
			case 1:  // hstem
			case 3:  // vstem
			case 18: // hstemhm
			case 23: // vstemhm
			{
				int len = sp;
				if ((len & 1) && !have_width) { // an odd count: the first operand is the width
					have_width = true;
					len--;
				}
				stems += (uint32_t)len >> 1;
				sp = 0;
				break;
			}
			case 19: // hintmask
			case 20: // cntrmask
			{
				int len = sp;
				sp = 0;
				if (len & 1) {
					len--;
					have_width = true;
				}
				stems += (uint32_t)len >> 1; // an implied vstem
				pos += (stems + 7) >> 3;
				if (pos > cs.size()) {
					if (!cff2)
						return false;
					pos = cs.size(); // (cff2: the stream simply ends; there is no endchar to miss)
				}
				break;
			}
			case 21: // rmoveto
			{
				int skip = 0;
				if (sp == 3 && !have_width) { // (a second width operand is an error: ttf-parser checks the stack length)
					skip = 1;
					have_width = true;
				}
				if (!do_move(skip, true, true))
					return false;
				break;
			}
			case 22: // hmoveto
			{
				int skip = 0;
				if (sp == 2 && !have_width) {
					skip = 1;
					have_width = true;
				}
				if (!do_move(skip, true, false))
					return false;
				break;
			}
			case 4: // vmoveto
			{
				int skip = 0;
				if (sp == 2 && !have_width) {
					skip = 1;
					have_width = true;
				}
				if (!do_move(skip, false, true))
					return false;
				break;
			}
			case 5: // rlineto
				if (!has_move_to || (sp & 1))
					return false;
				for (int i = 0; i < sp; i += 2) {
					x += stack[i];
					y += stack[i + 1];
					line_to(x, y);
				}
				sp = 0;
				break;
			case 6: // hlineto
				if (!do_alternating_lines(true))
					return false;
				break;
			case 7: // vlineto
				if (!do_alternating_lines(false))
					return false;
				break;
			case 8: // rrcurveto
				if (!has_move_to || sp % 6 != 0)
					return false;
				for (int i = 0; i < sp; i += 6)
					curve_rel(i);
				sp = 0;
				break;
			case 24: // rcurveline
			{
				if (!has_move_to || sp < 8 || (sp - 2) % 6 != 0)
					return false;
				int i = 0;
				for (; i + 6 <= sp - 2; i += 6)
					curve_rel(i);
				x += stack[i];
				y += stack[i + 1];
				line_to(x, y);
				sp = 0;
				break;
			}
			case 25: // rlinecurve
			{
				if (!has_move_to || sp < 8 || ((sp - 6) & 1))
					return false;
				int i = 0;
				for (; i + 2 <= sp - 6; i += 2) {
					x += stack[i];
					y += stack[i + 1];
					line_to(x, y);
				}
				curve_rel(i);
				sp = 0;
				break;
			}
			case 26: // vvcurveto
			{
				if (!has_move_to)
					return false;
				int i = 0;
				if (sp & 1) {
					x += stack[0];
					i = 1;
				}
				if ((sp - i) % 4 != 0)
					return false;
				for (; i < sp; i += 4) {
					const float x1 = x, y1 = y + stack[i];
					const float x2 = x1 + stack[i + 1], y2 = y1 + stack[i + 2];
					x = x2;
					y = y2 + stack[i + 3];
					curve_to(x1, y1, x2, y2, x, y);
				}
				sp = 0;
				break;
			}
			case 27: // hhcurveto
			{
				if (!has_move_to)
					return false;
				int i = 0;
				if (sp & 1) {
					y += stack[0];
					i = 1;
				}
				if ((sp - i) % 4 != 0)
					return false;
				for (; i < sp; i += 4) {
					const float x1 = x + stack[i], y1 = y;
					const float x2 = x1 + stack[i + 1], y2 = y1 + stack[i + 2];
					x = x2 + stack[i + 3];
					y = y2;
					curve_to(x1, y1, x2, y2, x, y);
				}
				sp = 0;
				break;
			}
			case 30: // vhcurveto
				if (!do_hv_curves(false))
					return false;
				break;
			case 31: // hvcurveto
				if (!do_hv_curves(true))
					return false;
				break;
			case 10: // callsubr
			case 29: // callgsubr
			{
				if (sp == 0 || depth == kMaxDepth)
					return false;
				const CffTable::Index *subrs = op == 29 ? &t.global_subrs_ : local;
				if (!subrs)
					return false;
				const float fidx = stack[--sp];
				const long idx = (long)fidx + (long)bias(subrs->count);
				if ((float)(long)fidx != fidx || idx < 0 || idx >= (long)subrs->count)
					return false;
				const auto sub = subrs->get((uint32_t)idx);
				if (!sub || !run(*sub, depth + 1))
					return false;
				if (has_endchar) {
					if (pos != cs.size())
						return false; // data after endchar
					return true;
				}
				break;
			}
			case 15: // vsindex (CFF2): |- ivs vsindex |- , once and before the first blend
			{
				if (!cff2 || had_blend || had_vsindex || sp != 1)
					return false;
				const float v = stack[0];
				if (!(v >= 0.0f && v <= 65535.0f) || !select_blend_set((uint32_t)v))
					return false;
				had_vsindex = true;
				sp = 0;
				break;
			}
			case 16: // blend (CFF2): n values, then their k deltas each, then n; the values stay, moved to the default position's sum
			{
				if (!cff2 || sp == 0)
					return false;
				had_blend = true;
				const float fn = stack[--sp];
				if (!(fn >= 0.0f && fn <= 65535.0f))
					return false;
				const size_t n = (size_t)fn, k = scalars->size();
				const size_t len = n * (k + 1);
				if ((size_t)sp < len)
					return false;
				const size_t start = (size_t)sp - len;
				// popped from the top: value n - 1 first, each with its last region's delta first (f32, one product and one sum each)
				for (size_t i = n; i-- > 0;)
					for (size_t j = 0; j < k; j++) {
						const float delta = stack[--sp];
						stack[start + i] += delta * (*scalars)[k - j - 1];
					}
				break;
			}
			case 11: // return
				if (cff2)
					return false; // (not an operator of CFF2: a subroutine ends with its data)
				return true;
			case 14: // endchar
				if (cff2)
					return false;
				if (sp == 4 || (!have_width && sp == 5)) {
					// seac form: adx ady bchar achar — the base glyph, then the accent moved by (adx, ady), each a
					// charstring of its own (path, hints and width operand start afresh)
					const int a = sp - 4;
					const float adx = stack[a], ady = stack[a + 1], bchar = stack[a + 2], achar = stack[a + 3];
					have_width = true;
					sp = 0;
					if (depth == kMaxDepth)
						return false;
					if (!first_move_to) { // (a path of the accented glyph itself ends here)
						first_move_to = true;
						out.close();
					}
					if (!(bchar >= 0.0f && bchar <= 255.0f && achar >= 0.0f && achar <= 255.0f))
						return false;
					const auto bg = t.standard_code_to_glyph((uint32_t)bchar), ag = t.standard_code_to_glyph((uint32_t)achar);
					if (!bg || !ag)
						return false;
					const float origin[2][2] = {{0.0f, 0.0f}, {adx, ady}};
					const uint16_t part[2] = {*bg, *ag};
					for (int k = 0; k < 2; k++) {
						const auto pcs = t.charstrings_.get(part[k]);
						if (!pcs)
							return false;
						CharStringRun sub(t, out, bb);
						sub.local = local;
						sub.x = origin[k][0];
						sub.y = origin[k][1];
						if (!sub.run(*pcs, depth + 1) || !sub.has_endchar)
							return false;
					}
					if (pos != cs.size())
						return false; // data after endchar
					has_endchar = true;
					return true;
				}
				if (sp == 1 && !have_width)
					have_width = true;
				sp = 0;
				if (!first_move_to) {
					first_move_to = true;
					out.close();
				}
				if (pos != cs.size())
					return false; // data after endchar
				has_endchar = true;
				return true;
			case 12: {
				if (pos >= cs.size())
					return false;
				const uint8_t op2 = cs.u8(pos++);
				if (!has_move_to)
					return false;
				if (op2 == 35) { // flex
					if (sp != 13)
						return false;
					curve_rel(0);
					curve_rel(6);
				} else if (op2 == 34) { // hflex
					if (sp != 7)
						return false;
					const float y0 = y;
					float x1 = x + stack[0], y1 = y;
					float x2 = x1 + stack[1], y2 = y1 + stack[2];
					x = x2 + stack[3];
					y = y2;
					curve_to(x1, y1, x2, y2, x, y);
					x1 = x + stack[4], y1 = y;
					x2 = x1 + stack[5], y2 = y0;
					x = x2 + stack[6];
					y = y0;
					curve_to(x1, y1, x2, y2, x, y);
				} else if (op2 == 36) { // hflex1
					if (sp != 9)
						return false;
					const float y0 = y;
					float x1 = x + stack[0], y1 = y + stack[1];
					float x2 = x1 + stack[2], y2 = y1 + stack[3];
					x = x2 + stack[4];
					y = y2;
					curve_to(x1, y1, x2, y2, x, y);
					x1 = x + stack[5], y1 = y;
					x2 = x1 + stack[6], y2 = y1 + stack[7];
					x = x2 + stack[8];
					y = y0;
					curve_to(x1, y1, x2, y2, x, y);
				} else if (op2 == 37) { // flex1
					if (sp != 11)
						return false;
					const float x0 = x, y0 = y;
					curve_rel(0);
					const float x1 = x + stack[6], y1 = y + stack[7];
					const float x2 = x1 + stack[8], y2 = y1 + stack[9];
					if (std::fabs(x2 - x0) > std::fabs(y2 - y0)) {
						x = x2 + stack[10];
						y = y0;
					} else {
						x = x0;
						y = y2 + stack[10];
					}
					curve_to(x1, y1, x2, y2, x, y);
				} else {
					return false; // arithmetic, storage and conditional operators: unsupported (as in ttf-parser)
				}
				sp = 0;
				break;
			}
			default:
				return false; // 0, 2, 9, 13, 17 (and 15, 16 outside CFF2): reserved
			}
		}
		return true;
	}
};

bool CffTable::outline(uint16_t gid, OutlineBuilder &builder) const
{
	const auto cs = charstrings_.get(gid);
	if (!cs)
		return false;
	CffBounds bb;
	CharStringRun r(*this, builder, bb);
	r.local = local_subrs_for(gid);
	if (cff2_) {
		// the scalars of ItemVariationData 0 are loaded before the first operator: no store, no outline
		if (!r.select_blend_set(0) || !r.run(*cs, 0))
			return false;
		// (no endchar in CFF2 and no close() for the last contour: RingBuilder::into_rings saves the open ring,
		// /root/reference/src/render/ring_builder.rs:26-29)
	} else if (!r.run(*cs, 0) || !r.has_endchar) {
		return false;
	}
	// ttf-parser: a glyph that produced no point has no outline (ZeroBBox); neither has one whose bbox leaves i16
	if (bb.x0 == std::numeric_limits<float>::max())
		return false;
	auto fits = [](float v) { return v >= -32768.0f && v <= 32767.0f; };
	return fits(bb.x0) && fits(bb.y0) && fits(bb.x1) && fits(bb.y1);
}

} // namespace vg
