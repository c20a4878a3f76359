// index_files.cpp — see index_files.hpp.
#include "index_files.hpp"

#include <algorithm>
#include <cstdio>
#include <map>
#include <stdexcept>

#include "font_manager.hpp"

namespace vg {

std::string encode_codeblocks(const std::vector<uint32_t> &codepoints)
{
	std::vector<uint32_t> blocks;
	blocks.reserve(codepoints.size());
	for (uint32_t cp : codepoints)
		blocks.push_back(cp >> 4);
	std::sort(blocks.begin(), blocks.end());
	blocks.erase(std::unique(blocks.begin(), blocks.end()), blocks.end());
	std::string out;
	char buf[32];
	for (size_t i = 0; i < blocks.size();) {
		size_t j = i;
		while (j + 1 < blocks.size() && blocks[j + 1] == blocks[j] + 1)
			j++;
		if (!out.empty())
			out.push_back(',');
		if (i == j)
			std::snprintf(buf, sizeof buf, "%X", blocks[i]);
		else
			std::snprintf(buf, sizeof buf, "%X-%X", blocks[i], blocks[j]);
		out += buf;
		i = j + 1;
	}
	return out;
}

std::string json_quote(const std::string &s)
{
	std::string o = "\"";
	char buf[8];
	for (unsigned char c : s) {
		switch (c) {
		case '"': o += "\\\""; break;
		case '\\': o += "\\\\"; break;
		case '\b': o += "\\b"; break;
		case '\f': o += "\\f"; break;
		case '\n': o += "\\n"; break;
		case '\r': o += "\\r"; break;
		case '\t': o += "\\t"; break;
		default:
			if (c < 0x20) {
				std::snprintf(buf, sizeof buf, "\\u%04x", c);
				o += buf;
			} else {
				o.push_back((char)c);
			}
		}
	}
	o.push_back('"');
	return o;
}

std::vector<uint8_t> build_index_json(const FontManager &m)
{
	std::string s;
	if (m.fonts().empty()) {
		s = "[]";
	} else {
		s = "[\n";
		size_t i = 0;
		for (const auto &kv : m.fonts()) { // std::map: already sorted by id (index_files.rs:115)
			s += "  " + json_quote(kv.first);
			s += ++i < m.fonts().size() ? ",\n" : "\n";
		}
		s += "]";
	}
	return std::vector<uint8_t>(s.begin(), s.end());
}

std::vector<uint8_t> build_font_families_json(const FontManager &m)
{
	struct Face {
		std::string id, style, width, codeblocks;
		uint16_t weight;
	};
	// family name -> faces.  The reference walks a HashMap (arbitrary order) and sorts only the
	// families (index_files.rs:141); faces inside a family are listed here in ascending id.
	std::map<std::string, std::vector<Face>> families;
	for (const auto &kv : m.fonts()) {
		if (kv.second.files().empty())
			throw std::runtime_error("FontWrapper has no files");
		const FontFileEntry &first = *kv.second.files().front(); // wrapper.rs:84-90: the first file speaks for the font
		const FontMetadata &meta = first.metadata();
		families[meta.family].push_back(Face{kv.first, meta.style, meta.width, encode_codeblocks(first.codepoints()), meta.weight});
	}
	std::string s;
	if (families.empty()) {
		s = "[]";
	} else {
		s = "[\n";
		size_t fi = 0;
		for (const auto &fam : families) {
			s += "  {\n    \"name\": " + json_quote(fam.first) + ",\n    \"faces\": [\n";
			for (size_t k = 0; k < fam.second.size(); k++) {
				const Face &f = fam.second[k];
				s += "      {\n";
				s += "        \"id\": " + json_quote(f.id) + ",\n";
				s += "        \"style\": " + json_quote(f.style) + ",\n";
				s += "        \"weight\": " + std::to_string(f.weight) + ",\n";
				s += "        \"width\": " + json_quote(f.width) + ",\n";
				s += "        \"codeblocks\": " + json_quote(f.codeblocks) + "\n";
				s += k + 1 < fam.second.size() ? "      },\n" : "      }\n";
			}
			s += "    ]\n";
			s += ++fi < families.size() ? "  },\n" : "  }\n";
		}
		s += "]";
	}
	return std::vector<uint8_t>(s.begin(), s.end());
}

// ---- a small JSON reader (RFC 8259) for fonts.json ---------------------------------------
namespace {

struct JsonReader {
	const std::string &t;
	size_t i = 0;
	std::string err;
	int depth = 0;
	explicit JsonReader(const std::string &text) : t(text) {}

	bool fail(const std::string &m)
	{
		if (err.empty())
			err = m + " at byte " + std::to_string(i);
		return false;
	}
	void ws()
	{
		while (i < t.size() && (t[i] == ' ' || t[i] == '\t' || t[i] == '\n' || t[i] == '\r'))
			i++;
	}
	bool lit(const char *w)
	{
		const size_t n = std::char_traits<char>::length(w);
		if (t.compare(i, n, w) != 0)
			return fail("invalid literal");
		i += n;
		return true;
	}
	static void put_utf8(std::string &o, uint32_t c)
	{
		if (c < 0x80) {
			o.push_back((char)c);
		} else if (c < 0x800) {
			o.push_back((char)(0xC0 | (c >> 6)));
			o.push_back((char)(0x80 | (c & 63)));
		} else if (c < 0x10000) {
			o.push_back((char)(0xE0 | (c >> 12)));
			o.push_back((char)(0x80 | ((c >> 6) & 63)));
			o.push_back((char)(0x80 | (c & 63)));
		} else {
			o.push_back((char)(0xF0 | (c >> 18)));
			o.push_back((char)(0x80 | ((c >> 12) & 63)));
			o.push_back((char)(0x80 | ((c >> 6) & 63)));
			o.push_back((char)(0x80 | (c & 63)));
		}
	}
	bool hex4(uint32_t &v)
	{
		if (i + 4 > t.size())
			return fail("short \\u escape");
		v = 0;
		for (int k = 0; k < 4; k++) {
			const char c = t[i++];
			v <<= 4;
			if (c >= '0' && c <= '9')
				v |= (uint32_t)(c - '0');
			else if (c >= 'a' && c <= 'f')
				v |= (uint32_t)(c - 'a' + 10);
			else if (c >= 'A' && c <= 'F')
				v |= (uint32_t)(c - 'A' + 10);
			else
				return fail("bad \\u escape");
		}
		return true;
	}
	bool string(std::string *out)
	{
		if (i >= t.size() || t[i] != '"')
			return fail("expected a string");
		i++;
		std::string o;
		while (true) {
			if (i >= t.size())
				return fail("unterminated string");
			const unsigned char c = (unsigned char)t[i++];
			if (c == '"')
				break;
			if (c < 0x20)
				return fail("control character in string");
			if (c != '\\') {
				o.push_back((char)c);
				continue;
			}
			if (i >= t.size())
				return fail("unterminated escape");
			const char e = t[i++];
			switch (e) {
			case '"': o.push_back('"'); break;
			case '\\': o.push_back('\\'); break;
			case '/': o.push_back('/'); break;
			case 'b': o.push_back('\b'); break;
			case 'f': o.push_back('\f'); break;
			case 'n': o.push_back('\n'); break;
			case 'r': o.push_back('\r'); break;
			case 't': o.push_back('\t'); break;
			case 'u': {
				uint32_t v;
				if (!hex4(v))
					return false;
				if (v >= 0xD800 && v <= 0xDBFF) {
					uint32_t lo;
					if (t.compare(i, 2, "\\u") != 0)
						return fail("lone surrogate");
					i += 2;
					if (!hex4(lo))
						return false;
					if (lo < 0xDC00 || lo > 0xDFFF)
						return fail("lone surrogate");
					v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
				} else if (v >= 0xDC00 && v <= 0xDFFF) {
					return fail("lone surrogate");
				}
				put_utf8(o, v);
				break;
			}
			default:
				return fail("invalid escape");
			}
		}
		if (out)
			*out = std::move(o);
		return true;
	}
	bool number()
	{
		const size_t s = i;
		if (i < t.size() && t[i] == '-')
			i++;
		if (i >= t.size() || t[i] < '0' || t[i] > '9')
			return fail("invalid number");
		if (t[i] == '0')
			i++;
		else
			while (i < t.size() && t[i] >= '0' && t[i] <= '9')
				i++;
		if (i < t.size() && t[i] == '.') {
			i++;
			if (i >= t.size() || t[i] < '0' || t[i] > '9')
				return fail("invalid number");
			while (i < t.size() && t[i] >= '0' && t[i] <= '9')
				i++;
		}
		if (i < t.size() && (t[i] == 'e' || t[i] == 'E')) {
			i++;
			if (i < t.size() && (t[i] == '+' || t[i] == '-'))
				i++;
			if (i >= t.size() || t[i] < '0' || t[i] > '9')
				return fail("invalid number");
			while (i < t.size() && t[i] >= '0' && t[i] <= '9')
				i++;
		}
		return i > s;
	}
	// any value, discarded (unknown keys)
	bool skip()
	{
		if (++depth > 128)
			return fail("recursion limit exceeded");
		ws();
		bool ok;
		if (i >= t.size()) {
			ok = fail("unexpected end of input");
		} else if (t[i] == '"') {
			ok = string(nullptr);
		} else if (t[i] == '{') {
			i++;
			ws();
			ok = true;
			if (i < t.size() && t[i] == '}') {
				i++;
			} else {
				while (ok) {
					ws();
					ok = string(nullptr);
					if (!ok)
						break;
					ws();
					if (i >= t.size() || t[i] != ':') {
						ok = fail("expected ':'");
						break;
					}
					i++;
					ok = skip();
					if (!ok)
						break;
					ws();
					if (i < t.size() && t[i] == ',') {
						i++;
						continue;
					}
					if (i < t.size() && t[i] == '}') {
						i++;
						break;
					}
					ok = fail("expected ',' or '}'");
				}
			}
		} else if (t[i] == '[') {
			i++;
			ws();
			ok = true;
			if (i < t.size() && t[i] == ']') {
				i++;
			} else {
				while (ok) {
					ok = skip();
					if (!ok)
						break;
					ws();
					if (i < t.size() && t[i] == ',') {
						i++;
						continue;
					}
					if (i < t.size() && t[i] == ']') {
						i++;
						break;
					}
					ok = fail("expected ',' or ']'");
				}
			}
		} else if (t[i] == 't') {
			ok = lit("true");
		} else if (t[i] == 'f') {
			ok = lit("false");
		} else if (t[i] == 'n') {
			ok = lit("null");
		} else {
			ok = number();
		}
		depth--;
		return ok;
	}
};

} // namespace

bool parse_fonts_json(const std::string &text, std::vector<FontConfig> &out, std::string *err)
{
	JsonReader r(text);
	auto bail = [&](const std::string &m) {
		r.fail(m);
		if (err)
			*err = "fonts.json: " + r.err;
		return false;
	};
	out.clear();
	r.ws();
	if (r.i >= text.size() || text[r.i] != '[')
		return bail("expected an array of font configurations");
	r.i++;
	r.ws();
	if (r.i < text.size() && text[r.i] == ']') {
		r.i++;
	} else {
		while (true) {
			r.ws();
			if (r.i >= text.size() || text[r.i] != '{')
				return bail("expected an object {name, sources}");
			r.i++;
			FontConfig c;
			bool have_name = false, have_sources = false;
			r.ws();
			if (r.i < text.size() && text[r.i] == '}') {
				r.i++;
			} else {
				while (true) {
					r.ws();
					std::string key;
					if (!r.string(&key))
						return bail("expected a key");
					r.ws();
					if (r.i >= text.size() || text[r.i] != ':')
						return bail("expected ':'");
					r.i++;
					r.ws();
					if (key == "name") {
						if (have_name)
							return bail("duplicate field `name`");
						if (!r.string(&c.name))
							return bail("`name` must be a string");
						have_name = true;
					} else if (key == "sources") {
						if (have_sources)
							return bail("duplicate field `sources`");
						if (r.i >= text.size() || text[r.i] != '[')
							return bail("`sources` must be an array of strings");
						r.i++;
						r.ws();
						if (r.i < text.size() && text[r.i] == ']') {
							r.i++;
						} else {
							while (true) {
								r.ws();
								std::string s;
								if (!r.string(&s))
									return bail("`sources` must be an array of strings");
								c.sources.push_back(std::move(s));
								r.ws();
								if (r.i < text.size() && text[r.i] == ',') {
									r.i++;
									continue;
								}
								if (r.i < text.size() && text[r.i] == ']') {
									r.i++;
									break;
								}
								return bail("expected ',' or ']'");
							}
						}
						have_sources = true;
					} else if (!r.skip()) {
						return bail("invalid value");
					}
					r.ws();
					if (r.i < text.size() && text[r.i] == ',') {
						r.i++;
						continue;
					}
					if (r.i < text.size() && text[r.i] == '}') {
						r.i++;
						break;
					}
					return bail("expected ',' or '}'");
				}
			}
			if (!have_name)
				return bail("missing field `name`");
			if (!have_sources)
				return bail("missing field `sources`");
			out.push_back(std::move(c));
			r.ws();
			if (r.i < text.size() && text[r.i] == ',') {
				r.i++;
				continue;
			}
			if (r.i < text.size() && text[r.i] == ']') {
				r.i++;
				break;
			}
			return bail("expected ',' or ']'");
		}
	}
	r.ws();
	if (r.i != text.size())
		return bail("trailing characters");
	return true;
}

} // namespace vg
