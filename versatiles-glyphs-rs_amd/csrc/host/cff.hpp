// cff.hpp — `CFF ` (version 1) outlines: the slice of ttf-parser's cff1 table (crate ttf-parser 0.25.1, a
// third-party dependency of the reference, not vendored under /root/reference) behind
//   Face::outline_glyph      /root/reference/src/render/renderer.rs:110
// for OpenType fonts whose outlines are Type 2 charstrings.  The charstring interpreter follows Adobe
// Technical Note #5177 ("The Type 2 Charstring Format") and #5176 (CFF); the crate's observable rules are
// kept: callbacks in f32 with one addition per coordinate in operand order, `close()` before every further
// `move_to` and at `endchar`, the optional width operand taken once, at most 48 operands and 10 nested
// subroutine calls, a glyph without any point yields no outline.
// No fixture of the reference holds CFF outlines: parity with the crate is UNPINNED here; the reader is
// checked against fontTools' charstring interpreter instead (tests/test_cff_outlines.py).
// `endchar` in its seac form (an accented character assembled from two glyphs of the standard encoding) draws the
// base glyph and then the accent at (adx, ady), each as a charstring of its own, through the font's charset.
// Not handled (-> no outline for that glyph): the arithmetic / storage operators of ESC (ttf-parser rejects them
// too), seac in CID-keyed fonts or with the Expert charsets, CFF2.
#pragma once
#include <cstdint>
#include <optional>
#include <vector>

#include "ttf_face.hpp"

namespace vg {

class CffTable {
public:
	static std::optional<CffTable> parse(Bytes table);
	uint32_t number_of_glyphs() const { return charstrings_.count; }
	// false = ttf-parser returns None (callbacks already delivered stay delivered)
	bool outline(uint16_t glyph_id, OutlineBuilder &builder) const;

private:
	struct Index {
		Bytes data;        // the whole INDEX
		uint32_t count = 0;
		uint8_t off_size = 1;
		size_t offsets_at = 0, data_at = 0; // byte positions inside `data`
		std::optional<Bytes> get(uint32_t i) const;
	};
	static bool parse_index(Bytes table, size_t at, Index &out, size_t &end);
	struct PrivateDict {
		Index local_subrs;
	};
	static bool parse_private(Bytes table, size_t offset, size_t size, PrivateDict &out);
	const Index *local_subrs_for(uint16_t glyph_id) const;
	// glyph of a code of Adobe's StandardEncoding (seac operands), through the charset; nullopt: none
	std::optional<uint16_t> standard_code_to_glyph(uint32_t code) const;

	Bytes table_;
	Index global_subrs_, charstrings_;
	bool cid_ = false;
	PrivateDict private_;              // name-keyed fonts
	std::vector<PrivateDict> fd_priv_; // CID-keyed fonts: one per font dict
	Bytes fd_select_;                  // CID-keyed fonts: FDSelect, from its format byte
	size_t charset_at_ = 0;            // Top DICT `charset`: 0 / 1 / 2 = ISOAdobe / Expert / ExpertSubset, else its offset

	friend struct CharStringRun;
};

} // namespace vg
