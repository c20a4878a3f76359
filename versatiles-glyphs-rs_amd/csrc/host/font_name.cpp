// font_name.cpp — see font_name.hpp.
#include "font_name.hpp"

#include <algorithm>
#include <cstring>

namespace vg {

namespace {

// ---------------------------------------------------------------------------------------
// Words stripped from a family name.  Font projects that ship one file per script (Noto: "Noto Sans
// Tamil", "Noto Sans Old Italic", "Noto Sans JP", ...) put the script's name behind the family; the
// reference strips those words so that all subsets share one family (parse_font_name.rs:266-271, word
// by word against SCRIPT_TOKENS, :21-186).  Which words go decides the font id, hence the output
// directory and font_families.json, so the SET is part of the output contract: it is exactly the
// reference's 159 words (tests/golden/script_tokens.txt pins it; rounds 1-2 derived a superset from
// UAX #24 and turned "Source Han Sans" into "Source Sans").  One string, split at first use.
// ---------------------------------------------------------------------------------------
const char kStrippedWords[] =
	"aboriginal adlam albanian anatolian arabic aramaic armenian avestan balinese bamum bassa batak bengali "
	"bhaiksuki brahmi buginese buhid canadian carian caucasian chakma cham cherokee chiki cin coptic cuneiform "
	"cypriot deseret devanagari duployan egyptian elbasan elymaic ethiopic georgian glagolitic gondi gothic "
	"grantha gujarati gunjala gurmukhi hanifi hanunoo hatran hau hebrew hieroglyphs hmong hungarian imperial "
	"indic inscriptional italic javanese jp kaithi kannada kayah kharoshthi khmer khojki khudawadi kikakui "
	"kr lao le lepcha li limbu linear lisu lue lycian lydian mahajani malayalam mandaic manichaean marchen "
	"masaram mayan mayek medefaidrin meetei mende meroitic miao modi mongolian mro multani myanmar nabataean "
	"new newa nko north numbers nushu ogham ol old oriya osage osmanya pa pahawh pahlavi palmyrene parthian "
	"pau permic persian phags phoenician psalter rejang rohingya runic samaritan saurashtra sc sharada shavian "
	"siddham sinhala sogdian sompeng sora south soyombo square sundanese syloti symbols syriac tagalog tagbanwa "
	"tai takri tamil tangut tc telugu thaana thai tibetan tifinagh tirhuta turkic ugaritic vah vai wancho warang "
	"yi zanabazar";

// Rust's str::to_lowercase on the characters that can matter here: every comparison below is against ASCII
// words, and the only non-ASCII character whose lower case is an ASCII letter is U+212A KELVIN SIGN -> 'k'.
std::string ascii_lower(const std::string &s)
{
	std::string o;
	o.reserve(s.size());
	for (size_t i = 0; i < s.size(); i++) {
		const unsigned char c = (unsigned char)s[i];
		if (c == 0xE2 && i + 2 < s.size() && (unsigned char)s[i + 1] == 0x84 && (unsigned char)s[i + 2] == 0xAA) {
			o.push_back('k');
			i += 2;
		} else {
			o.push_back((c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c);
		}
	}
	return o;
}

const std::vector<std::string> &script_words()
{
	static const std::vector<std::string> words = [] {
		std::vector<std::string> w;
		std::string cur;
		for (const char *p = kStrippedWords;; p++) {
			if (*p == ' ' || *p == 0) {
				if (!cur.empty())
					w.push_back(cur);
				cur.clear();
				if (*p == 0)
					break;
			} else {
				cur.push_back(*p);
			}
		}
		std::sort(w.begin(), w.end());
		return w;
	}();
	return words;
}

bool is_script_word(const std::string &lower)
{
	const auto &w = script_words();
	return std::binary_search(w.begin(), w.end(), lower);
}

bool contains(const std::string &s, const char *needle) { return s.find(needle) != std::string::npos; }

// parse_font_name.rs:295-322: keyword search, most specific first; 400 = "no keyword"
uint16_t find_weight(const std::string &s)
{
	if (contains(s, "hairline") || contains(s, "thin"))
		return 100;
	if (contains(s, "extralight") || contains(s, "ultralight"))
		return 200;
	if (contains(s, "light"))
		return 300;
	if (contains(s, "regular") || contains(s, "normal") || contains(s, "book"))
		return 400;
	if (contains(s, "medium"))
		return 500;
	if (contains(s, "demibold") || contains(s, "semibold"))
		return 600;
	if (contains(s, "bold"))
		return (contains(s, "extra") || contains(s, "ultra")) ? 800 : 700;
	if (contains(s, "black") || contains(s, "heavy"))
		return 900;
	return 400;
}

// Rust's str::split_whitespace: Unicode White_Space.  Returns the byte length of the white-space
// character at s[i] (0 if none).
size_t whitespace_at(const std::string &s, size_t i)
{
	const unsigned char c = (unsigned char)s[i];
	if (c == ' ' || (c >= 0x09 && c <= 0x0D))
		return 1;
	if (c == 0xC2 && i + 1 < s.size()) { // U+0085, U+00A0
		const unsigned char d = (unsigned char)s[i + 1];
		return (d == 0x85 || d == 0xA0) ? 2 : 0;
	}
	if (i + 2 < s.size()) {
		const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
		if (c == 0xE1 && d == 0x9A && e == 0x80) // U+1680
			return 3;
		if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) // U+2000-200A, 2028, 2029, 202F
			return 3;
		if (c == 0xE2 && d == 0x81 && e == 0x9F) // U+205F
			return 3;
		if (c == 0xE3 && d == 0x80 && e == 0x80) // U+3000
			return 3;
	}
	return 0;
}

std::vector<std::string> split_whitespace(const std::string &s)
{
	std::vector<std::string> out;
	std::string cur;
	for (size_t i = 0; i < s.size();) {
		const size_t w = whitespace_at(s, i);
		if (w) {
			if (!cur.empty())
				out.push_back(cur);
			cur.clear();
			i += w;
		} else {
			cur.push_back(s[i++]);
		}
	}
	if (!cur.empty())
		out.push_back(cur);
	return out;
}

} // namespace

ParsedFontName parse_font_name(const std::string &family, const std::string &ps_name)
{
	ParsedFontName r;
	// what follows the last '-' of the PostScript name carries style and weight (:221-239)
	const size_t dash = ps_name.rfind('-');
	const std::string suffix = ascii_lower(dash == std::string::npos ? ps_name : ps_name.substr(dash + 1));
	if (contains(suffix, "italic"))
		r.style = "italic";
	const uint16_t ps_weight = find_weight(suffix);
	if (ps_weight != 400)
		r.weight = ps_weight;

	// family words: widths, script names and weight words are taken out (:242-286)
	const std::vector<std::string> tokens = split_whitespace(family);
	std::vector<const std::string *> kept;
	for (size_t i = 0; i < tokens.size(); i++) {
		const std::string t = ascii_lower(tokens[i]);
		if (i + 1 < tokens.size() && t == "extra" && ascii_lower(tokens[i + 1]) == "condensed") {
			r.width = "extra-condensed";
			i++;
			continue;
		}
		if (t == "semicondensed" || t == "semi-condensed") {
			r.width = "semi-condensed";
			continue;
		}
		if (t == "condensed") {
			r.width = "condensed";
			continue;
		}
		if (is_script_word(t))
			continue;
		const uint16_t w = find_weight(t);
		if (w != 400) {
			if (ps_weight == 400) // the PostScript suffix wins
				r.weight = w;
			continue;
		}
		kept.push_back(&tokens[i]);
	}
	for (size_t i = 0; i < kept.size(); i++) {
		if (i)
			r.family.push_back(' ');
		r.family += *kept[i];
	}
	return r;
}

std::string FontMetadata::generate_name() const
{
	std::string n = family;
	if (width != "normal")
		n += " " + width;
	const char *w = "Unknown";
	switch (weight) {
	case 100: w = "Thin"; break;
	case 200: w = "ExtraLight"; break;
	case 300: w = "Light"; break;
	case 400: w = "Regular"; break;
	case 500: w = "Medium"; break;
	case 600: w = "SemiBold"; break;
	case 700: w = "Bold"; break;
	case 800: w = "ExtraBold"; break;
	case 900: w = "Black"; break;
	default: break;
	}
	n += std::string(" ") + w;
	if (style != "normal")
		n += " " + style;
	return n;
}

} // namespace vg
