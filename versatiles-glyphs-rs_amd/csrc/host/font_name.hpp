// font_name.hpp — family / style / weight / width of a font file, for the ingestion step that
// decides which files merge into one logical font:
//   parse_font_name           /root/reference/src/font/parse_font_name.rs:214-291
//   find_weight               parse_font_name.rs:295-322
//   FontMetadata              src/font/metadata.rs:16-36, 89-128 (name table -> metadata)
//   FontMetadata::generate_name   metadata.rs:43-68
// The behaviour is pinned by the reference's own test expectations, transcribed as a fixture in
// tests/golden/font_names.csv; the set of words stripped from a family is the reference's (SCRIPT_TOKENS,
// parse_font_name.rs:21-186; tests/golden/script_tokens.txt), since it decides font ids and directory names.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vg {

struct ParsedFontName {
	std::string family;
	std::string style = "normal"; // "normal" | "italic"
	uint16_t weight = 400;
	std::string width = "normal"; // "normal" | "condensed" | "semi-condensed" | "extra-condensed"
};

// parse_font_name(family, ps_name): see the header comment
ParsedFontName parse_font_name(const std::string &family, const std::string &ps_name);

// metadata.rs:16-36 (the code points live in FontFileEntry)
struct FontMetadata {
	std::string name; // raw name-table family (name id 1)
	std::string family, style = "normal", width = "normal";
	uint16_t weight = 400;
	std::string generate_name() const; // metadata.rs:43-68
};

} // namespace vg
