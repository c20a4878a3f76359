// index_files.hpp — the sidecar JSON files of an output tree and the
// fonts.json reader of the recursive scan (rank 3):
//   encode_codeblocks          /root/reference/src/font/index_files.rs:65-103
//   build_index_json           index_files.rs:113-117   (serde_json::to_vec_pretty of the sorted ids)
//   build_font_families_json   index_files.rs:129-143
//   FontConfig / fonts.json    src/commands/recurse.rs:57-63, 113-126
// The JSON text follows serde_json's pretty printer: two-space indent, `"key": value`, one array
// element per line, `[]` for an empty array.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vg {

class FontManager;

// 16-code-point blocks (cp >> 4) merged into ranges, upper-case hex, comma separated
std::string encode_codeblocks(const std::vector<uint32_t> &codepoints);
std::string json_quote(const std::string &utf8); // serde_json string escaping

std::vector<uint8_t> build_index_json(const FontManager &m);
// throws std::runtime_error("FontWrapper has no files") like wrapper.rs:84-90
std::vector<uint8_t> build_font_families_json(const FontManager &m);

// fonts.json: [{ "name": string, "sources": [string] }]; unknown keys are ignored, missing keys and
// wrong types are errors (serde's derive(Deserialize) behaviour).
struct FontConfig {
	std::string name;
	std::vector<std::string> sources;
};
bool parse_fonts_json(const std::string &text, std::vector<FontConfig> &out, std::string *err);

} // namespace vg
