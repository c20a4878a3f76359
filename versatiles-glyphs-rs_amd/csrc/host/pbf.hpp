// pbf.hpp — glyphs-PBF wire format (proto2), hand-encoded.  Output contract of the path:
//   PbfGlyph   /root/reference/src/protobuf/glyph.rs:10-41   (tags 1..7)
//   Fontstack  src/protobuf/fontstack.rs:9-25               (name, range, glyphs)
//   PbfGlyphs  src/protobuf/glyphs.rs:11-16,66-70           (stacks, into_vec)
// prost writes fields in tag order, always writes proto2 `required` scalars (even 0),
// writes `optional bytes` only when Some, zig-zags sint32, length-prefixes messages.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

namespace vg {

struct PbfGlyph {
	uint32_t id = 0;
	std::optional<std::vector<uint8_t>> bitmap;
	uint32_t width = 0, height = 0;
	int32_t left = 0, top = 0;
	uint32_t advance = 0;

	// glyph.rs:60-70
	static PbfGlyph empty(uint32_t id, uint32_t advance)
	{
		PbfGlyph g;
		g.id = id;
		g.advance = advance;
		return g;
	}
};

// A glyph whose bitmap lives in someone else's buffer (the batch download buffer): lets
// the block encoder copy pixels once, straight into the PBF.
struct PbfGlyphRef {
	uint32_t id = 0;
	const uint8_t *bitmap = nullptr; // nullptr = no bitmap field
	size_t bitmap_len = 0;
	uint32_t width = 0, height = 0;
	int32_t left = 0, top = 0;
	uint32_t advance = 0;
};

// ---- in-place assembly (include/vgsdf.h, vgsdf_outlines_packed::pbf_pre / pbf_fix) ------------------------------
// The raster stores every bitmap where the finished block file has it, inside an arena that holds the blocks of a
// submission one after the other; the ~20 bytes around each bitmap and the block headers are written here once the
// glyphs' rects are known.  The arithmetic below is the device's (csrc/outline_kernels.hip, pbf_place), which has
// already placed the bitmaps.
inline uint32_t pbf_varint_len(uint64_t v)
{
	uint32_t n = 1;
	for (; v >= 0x80; v >>= 7)
		n++;
	return n;
}
// bytes reserved in front of a block's first entry: the file header 0x0A varint(stack) right-aligned in
// kPbfHeadRoom bytes (so that nothing behind it depends on the length of that varint), then the name and range fields
constexpr uint32_t kPbfHeadRoom = 6;
inline uint32_t pbf_block_fields(size_t name_len, size_t range_len)
{
	return (uint32_t)(1 + pbf_varint_len(name_len) + name_len + 1 + pbf_varint_len(range_len) + range_len);
}
inline uint8_t pbf_fix_of(uint32_t id, uint32_t advance)
{
	return (uint8_t)((1 + pbf_varint_len(id)) | ((1 + pbf_varint_len(advance)) << 4));
}
// size of a glyph's entry and the offset of its bitmap in it (for a glyph without a raster: of the `width` tag)
struct PbfEntrySize {
	uint64_t total = 0, bitmap_at = 0;
};
PbfEntrySize pbf_entry_size(uint32_t id, uint32_t advance, bool has_raster, uint32_t w, uint32_t h, int32_t x0, int32_t y0);
// One glyph of the arena: writes the entry's header bytes around the bitmap at `at` (= start of the entry) and returns
// the entry's size.  has_raster, w, h, x0, y0: the glyph's rect (RenderResult incl. the 3 px buffer).
size_t write_pbf_entry_headers(uint8_t *at, uint32_t id, uint32_t advance, bool has_raster, uint32_t w, uint32_t h, int32_t x0,
                               int32_t y0);
// The block's file header + name + range in front of its first entry (`entries` = start of the first entry, `stack` =
// bytes of all its entries): returns the first byte of the file.
uint8_t *write_pbf_block_header(uint8_t *entries, const std::string &name, const std::string &range, size_t entries_bytes);

class PbfGlyphs {
public:
	// glyphs.rs:28-32
	PbfGlyphs(std::string name, std::string range) : name_(std::move(name)), range_(std::move(range)) {}
	// glyphs.rs:44-46
	void push(PbfGlyph g) { owned_.push_back(std::move(g)); }
	// glyphs.rs:66-70.  Glyphs are written in ascending id (the reference iterates a HashMap,
	// so its order is arbitrary; ascending id is its own canonical form, commands/debug.rs:79).
	std::vector<uint8_t> into_vec() const;

	static std::vector<uint8_t> encode(const std::string &name, const std::string &range,
	                                   std::vector<PbfGlyphRef> glyphs);

private:
	std::string name_, range_;
	std::vector<PbfGlyph> owned_;
};

} // namespace vg
