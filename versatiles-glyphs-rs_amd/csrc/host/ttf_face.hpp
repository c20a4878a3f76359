// ttf_face.hpp — the slice of ttf_parser::Face (crate ttf-parser 0.25.1, a third-party
// dependency of the reference, not vendored under /root/reference) that the render path
// calls:
//   Face::glyph_index        /root/reference/src/render/renderer.rs:106
//   Face::units_per_em       renderer.rs:107
//   Face::outline_glyph      renderer.rs:110   (glyf outlines -> OutlineBuilder callbacks)
//   Face::glyph_hor_advance  renderer.rs:115
//   cmap subtable walk       src/font/metadata.rs:105-117 (code point coverage)
// Static fonts: TrueType (`glyf`) outlines and `CFF ` (version 1) charstrings (cff.hpp); CFF2 / variable
// fonts are out of scope (none in the reference's testdata) and are refused at load time.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace vg {

class CffTable;

// ttf_parser::OutlineBuilder — f32 font units
struct OutlineBuilder {
	virtual ~OutlineBuilder() = default;
	virtual void move_to(float x, float y) = 0;
	virtual void line_to(float x, float y) = 0;
	virtual void quad_to(float x1, float y1, float x, float y) = 0;
	virtual void curve_to(float x1, float y1, float x2, float y2, float x, float y) = 0;
	virtual void close() = 0;
	static constexpr bool kRawCursor = false; // (the glyf walk's packed recorder in ttf_face.cpp is the sink that has one)
};

// One simple glyph of a (possibly composite) glyph in the form the device's glyf decoder takes (vgsdf_glyf_part of
// include/vgsdf.h, field for field): the entry's end points + flag / coordinate arrays copied into a byte store, the
// component transform ttf-parser has accumulated, and the command slots the entry may fill.
struct GlyfPart {
	uint32_t byte_off, byte_len;
	uint32_t cmd_at, cmd_cap;
	uint32_t n_contours, plain;
	float a, b, c, d, e, f;
};

// Non-owning big-endian byte view with checked reads.
class Bytes {
public:
	Bytes() = default;
	Bytes(const uint8_t *p, size_t n) : p_(p), n_(n) {}
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	const uint8_t *data() const { return p_; }
	bool has(size_t off, size_t len) const { return off <= n_ && len <= n_ - off; }
	uint8_t u8(size_t off) const { return p_[off]; }
	uint16_t u16(size_t off) const { return (uint16_t)((p_[off] << 8) | p_[off + 1]); }
	int16_t i16(size_t off) const { return (int16_t)u16(off); }
	uint32_t u32(size_t off) const
	{
		return ((uint32_t)p_[off] << 24) | ((uint32_t)p_[off + 1] << 16) | ((uint32_t)p_[off + 2] << 8) |
		       (uint32_t)p_[off + 3];
	}
	Bytes sub(size_t off, size_t len) const { return has(off, len) ? Bytes(p_ + off, len) : Bytes(); }
	Bytes from(size_t off) const { return off <= n_ ? Bytes(p_ + off, n_ - off) : Bytes(); }

private:
	const uint8_t *p_ = nullptr;
	size_t n_ = 0;
};

class Face {
public:
	// Face::parse(data, 0).  The bytes must outlive the Face.
	static std::optional<Face> parse(const uint8_t *data, size_t len);

	uint16_t units_per_em() const { return units_per_em_; }
	uint16_t number_of_glyphs() const { return num_glyphs_; }
	std::optional<uint16_t> glyph_index(uint32_t code_point) const;
	std::optional<uint16_t> glyph_hor_advance(uint16_t glyph_id) const;
	// Emits the glyph's outline; returns false when ttf-parser would return None
	// (callbacks already delivered stay delivered, as in the crate).
	bool outline_glyph(uint16_t glyph_id, OutlineBuilder &builder) const;
	// The same callbacks appended to `kinds` / `coords` in the compact upload form of vgsdf_outlines_packed (kind byte
	// 0..4 = move / line / quad / curve / close + the coordinates the kind carries); glyf outlines are walked with the sink
	// inlined (no virtual call per point).
	bool outline_glyph_packed(uint16_t glyph_id, std::vector<uint8_t> &kinds, std::vector<float> &coords) const;
	// glyf fonts only (has_glyf_outlines()): the same walk up to the simple glyphs, which are not decoded but appended as
	// parts — their bytes to `bytes` (4-aligned), their command slots counted from `slots` on.  false: ttf-parser returns
	// None at this point (parts appended so far stay, as the callbacks delivered so far would).  *overflow is set (and the
	// walk stops) when the batch would pass what its 32-bit offsets address (2^26 bytes / slots, 2^22 parts per recorder): a composite
	// copies its simple glyphs once per leaf, so a small font can fan out to gigabytes — the caller then drops the glyf form.
	bool glyph_parts(uint16_t glyph_id, std::vector<GlyfPart> &parts, std::vector<uint8_t> &bytes, uint32_t &slots, bool *overflow = nullptr) const;
	// ttf-parser's `tables().cmap.is_some()`; the reference refuses fonts without one (metadata.rs:104-107)
	bool has_cmap() const { return has_cmap_; }
	// glyph outlines this reader can emit: `glyf` + `loca`, or `CFF ` charstrings (ttf-parser's order: glyf first).
	// A font whose outlines live in a table this reader cannot walk (`CFF2`, or a `CFF ` table it fails to parse)
	// is refused at load time instead of yielding empty glyphs.
	bool has_glyf_outlines() const { return !glyf_.empty() && !loca_.empty(); }
	bool has_cff_outlines() const { return cff_ != nullptr; }
	bool has_unsupported_outlines() const { return !has_glyf_outlines() && !cff_ && cff_unreadable_; }
	// Face::names() (src/font/metadata.rs:92-97): every record of the `name` table in table order as
	// (name_id, Name::to_string().unwrap_or_default()): UTF-16BE records of the Unicode platform and of
	// the Windows platform (encodings 0 and 1) decoded to UTF-8, every other record an empty string.
	std::vector<std::pair<uint16_t, std::string>> names() const;
	// Sorted unique code points that a unicode cmap subtable maps to a glyph.
	std::vector<uint32_t> unicode_codepoints() const;

private:
	struct CmapSubtable {
		uint16_t platform = 0, encoding = 0, format = 0xFFFF;
		Bytes data; // from the subtable start to the end of the cmap table
		bool is_unicode() const;
		std::optional<uint16_t> glyph_index(uint32_t cp) const;
		template <class F> void for_each_codepoint(F &&f) const;
	};

	std::optional<Bytes> glyph_data(uint16_t glyph_id) const;

	Bytes hmtx_, loca_, glyf_, name_;
	std::vector<CmapSubtable> cmap_;
	uint16_t units_per_em_ = 0, num_glyphs_ = 0, num_hmetrics_ = 0;
	bool loca_long_ = false, has_cmap_ = false, cff_unreadable_ = false;
	std::shared_ptr<const CffTable> cff_;
	size_t loca_entries_ = 0;

	template <class B, bool PARTS> friend struct GlyfWalker;
};

} // namespace vg
