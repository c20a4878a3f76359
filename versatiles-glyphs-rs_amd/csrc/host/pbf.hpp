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
