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
