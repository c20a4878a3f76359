#include "pbf.hpp"

#include <algorithm>
#include <cstring>

namespace vg {

namespace {

inline size_t varint_size(uint64_t v)
{
	size_t n = 1;
	for (; v >= 0x80; v >>= 7)
		n++;
	return n;
}
inline uint8_t *write_varint(uint8_t *p, uint64_t v)
{
	for (; v >= 0x80; v >>= 7)
		*p++ = (uint8_t)(v | 0x80);
	*p++ = (uint8_t)v;
	return p;
}
inline uint32_t zigzag(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }

size_t glyph_payload_size(const PbfGlyphRef &g)
{
	size_t n = 1 + varint_size(g.id);
	if (g.bitmap)
		n += 1 + varint_size(g.bitmap_len) + g.bitmap_len;
	n += 1 + varint_size(g.width) + 1 + varint_size(g.height);
	n += 1 + varint_size(zigzag(g.left)) + 1 + varint_size(zigzag(g.top));
	n += 1 + varint_size(g.advance);
	return n;
}

} // namespace

std::vector<uint8_t> PbfGlyphs::encode(const std::string &name, const std::string &range,
                                       std::vector<PbfGlyphRef> glyphs)
{
	std::stable_sort(glyphs.begin(), glyphs.end(),
	                 [](const PbfGlyphRef &a, const PbfGlyphRef &b) { return a.id < b.id; });
	size_t stack = 1 + varint_size(name.size()) + name.size() + 1 + varint_size(range.size()) + range.size();
	std::vector<size_t> sizes(glyphs.size());
	for (size_t i = 0; i < glyphs.size(); i++) {
		sizes[i] = glyph_payload_size(glyphs[i]);
		stack += 1 + varint_size(sizes[i]) + sizes[i];
	}
	std::vector<uint8_t> out(1 + varint_size(stack) + stack);
	uint8_t *p = out.data();
	*p++ = 0x0A; // glyphs.stacks (tag 1, LEN)
	p = write_varint(p, stack);
	*p++ = 0x0A; // fontstack.name
	p = write_varint(p, name.size());
	std::memcpy(p, name.data(), name.size());
	p += name.size();
	*p++ = 0x12; // fontstack.range
	p = write_varint(p, range.size());
	std::memcpy(p, range.data(), range.size());
	p += range.size();
	for (size_t i = 0; i < glyphs.size(); i++) {
		const PbfGlyphRef &g = glyphs[i];
		*p++ = 0x1A; // fontstack.glyphs (tag 3, LEN)
		p = write_varint(p, sizes[i]);
		*p++ = 0x08;
		p = write_varint(p, g.id);
		if (g.bitmap) {
			*p++ = 0x12;
			p = write_varint(p, g.bitmap_len);
			std::memcpy(p, g.bitmap, g.bitmap_len);
			p += g.bitmap_len;
		}
		*p++ = 0x18;
		p = write_varint(p, g.width);
		*p++ = 0x20;
		p = write_varint(p, g.height);
		*p++ = 0x28;
		p = write_varint(p, zigzag(g.left));
		*p++ = 0x30;
		p = write_varint(p, zigzag(g.top));
		*p++ = 0x38;
		p = write_varint(p, g.advance);
	}
	return out;
}

PbfEntrySize pbf_entry_size(uint32_t id, uint32_t advance, bool has_raster, uint32_t w, uint32_t h, int32_t x0, int32_t y0)
{
	const uint32_t width = has_raster ? w - 6u : 0u, height = has_raster ? h - 6u : 0u;
	const int32_t left = has_raster ? (int32_t)((uint32_t)x0 + 3u) : 0, top = has_raster ? (int32_t)((uint32_t)y0 + h - 27u) : 0;
	const uint64_t px = has_raster ? (uint64_t)w * h : 0;
	uint64_t msg = 1 + varint_size(id) + 1 + varint_size(advance) + 1 + varint_size(width) + 1 + varint_size(height) + 1 +
	               varint_size(zigzag(left)) + 1 + varint_size(zigzag(top));
	if (has_raster)
		msg += 1 + varint_size(px) + px;
	PbfEntrySize e;
	e.bitmap_at = 1 + varint_size(msg) + 1 + varint_size(id) + (has_raster ? 1 + varint_size(px) : 0);
	e.total = 1 + varint_size(msg) + msg;
	return e;
}

size_t write_pbf_entry_headers(uint8_t *at, uint32_t id, uint32_t advance, bool has_raster, uint32_t w, uint32_t h, int32_t x0,
                               int32_t y0)
{
	// result.rs:66-76 after renderer.rs:146 (i32 arithmetic, two's complement); PbfGlyph::empty otherwise (glyph.rs:60-70)
	const uint32_t width = has_raster ? w - 6u : 0u, height = has_raster ? h - 6u : 0u;
	const int32_t left = has_raster ? (int32_t)((uint32_t)x0 + 3u) : 0, top = has_raster ? (int32_t)((uint32_t)y0 + h - 27u) : 0;
	const uint64_t px = has_raster ? (uint64_t)w * h : 0;
	uint64_t msg = 1 + varint_size(id) + 1 + varint_size(advance) + 1 + varint_size(width) + 1 + varint_size(height) + 1 +
	               varint_size(zigzag(left)) + 1 + varint_size(zigzag(top));
	if (has_raster)
		msg += 1 + varint_size(px) + px;
	uint8_t *p = at;
	*p++ = 0x1A;
	p = write_varint(p, msg);
	*p++ = 0x08;
	p = write_varint(p, id);
	if (has_raster) {
		*p++ = 0x12;
		p = write_varint(p, px);
		p += px; // the bitmap: stored by the raster
	}
	*p++ = 0x18;
	p = write_varint(p, width);
	*p++ = 0x20;
	p = write_varint(p, height);
	*p++ = 0x28;
	p = write_varint(p, zigzag(left));
	*p++ = 0x30;
	p = write_varint(p, zigzag(top));
	*p++ = 0x38;
	p = write_varint(p, advance);
	return (size_t)(p - at);
}

uint8_t *write_pbf_block_header(uint8_t *entries, const std::string &name, const std::string &range, size_t entries_bytes)
{
	const size_t fields = pbf_block_fields(name.size(), range.size());
	const size_t stack = fields + entries_bytes;
	uint8_t *p = entries - fields;
	uint8_t *file = p - 1 - varint_size(stack); // right-aligned in the kPbfHeadRoom bytes in front of the fields
	uint8_t *q = file;
	*q++ = 0x0A; // glyphs.stacks
	q = write_varint(q, stack);
	*q++ = 0x0A; // fontstack.name
	q = write_varint(q, name.size());
	std::memcpy(q, name.data(), name.size());
	q += name.size();
	*q++ = 0x12; // fontstack.range
	q = write_varint(q, range.size());
	std::memcpy(q, range.data(), range.size());
	return file;
}

std::vector<uint8_t> PbfGlyphs::into_vec() const
{
	std::vector<PbfGlyphRef> refs;
	refs.reserve(owned_.size());
	for (const PbfGlyph &g : owned_) {
		PbfGlyphRef r;
		r.id = g.id;
		if (g.bitmap) {
			static const uint8_t kEmpty = 0;
			r.bitmap = g.bitmap->empty() ? &kEmpty : g.bitmap->data();
			r.bitmap_len = g.bitmap->size();
		}
		r.width = g.width;
		r.height = g.height;
		r.left = g.left;
		r.top = g.top;
		r.advance = g.advance;
		refs.push_back(r);
	}
	return encode(name_, range_, std::move(refs));
}

} // namespace vg
