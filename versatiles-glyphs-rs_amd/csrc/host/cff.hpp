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
// too), seac in CID-keyed fonts or with the Expert charsets.
//
// `CFF2` (variable fonts; the crate's cff2 table, used when the face has neither `glyf` nor a readable `CFF `): the
// same interpreter without width operand, `endchar` and `return` (operators 0, 2, 9, 11, 13, 14, 17 end the glyph), with
// up to 513 operands, 32-bit INDEX counts, one set of local subroutines (the first Font DICT whose Private DICT has
// one; FDSelect is not consulted) and the `vsindex` / `blend` operators.  The reference never sets variation
// coordinates (renderer.rs:103-110 uses the face as parsed), so every glyph is drawn at the default position of the
// design space: a region's factor is 1 where all its axes peak at 0 (or are malformed in one of the ways the crate
// maps to 1) and 0 otherwise, and `blend` adds delta * factor to its operands in f32, last region first.  Two
// rules of the crate that a reader written from the specification alone would not have: a face without `fvar` has
// NO coordinates, which makes every factor 1 (all deltas are added); and the scalars of ItemVariationData 0 are
// loaded before the first operator, so a table without a VariationStore yields no outline at all.
// PARITY UNPINNED like the rest of CFF, and more so: these rules are restated from the crate's documented behaviour
// without a fixture; the default-position outlines are checked against fontTools (tests/test_cff2_outlines.py).
#pragma once
#include <cstdint>
#include <optional>
#include <vector>

#include "ttf_face.hpp"

namespace vg {

class CffTable {
public:
	static std::optional<CffTable> parse(Bytes table);
	// `CFF2`; n_coords = the face's variation coordinates (fvar axes, at most 64; all at 0), 0 without `fvar`
	static std::optional<CffTable> parse2(Bytes table, uint32_t n_coords);
	bool is_cff2() const { return cff2_; }
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
	static bool parse_index(Bytes table, size_t at, Index &out, size_t &end, bool count32 = false);
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
	// CFF2: per ItemVariationData subtable the factor of each of its regions at the default position
	struct BlendSet {
		bool ok = false; // false: no such subtable, or more than 64 regions (the crate's limit)
		std::vector<float> scalars;
	};
	bool cff2_ = false;
	std::vector<BlendSet> blend_sets_;

	friend struct CharStringRun;
};

} // namespace vg
