/*
 * vg_oracle.c — CPU ORACLE (test infrastructure only; see vg_oracle.h).
 *
 * Scalar f64 restatement of the reference's per-glyph SDF path.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference/).  Compile
 * with -ffp-contract=off (Rust never fuses a*b+c).  The `CFF ` / `CFF2` readers further down restate a
 * third-party crate's behaviour from Adobe's technical notes and the OpenType specification; no
 * reference fixture pins them (parity unpinned, see their header comments).
 */
#define _POSIX_C_SOURCE 200809L
#include "vg_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* src/render/mod.rs:52-68 */
#define GLYPH_SIZE 24
#define BUFFER 3
#define SDF_RADIUS 8.0
#define CUTOFF (0.25 * 256.0)
#define MAX_COMPONENTS 32 /* ttf-parser glyf.rs */

/* ------------------------------------------------------------------------------------
 * big-endian stream helpers
 * ---------------------------------------------------------------------------------- */
static inline int rd_ok(size_t len, size_t off, size_t n) { return off <= len && n <= len - off; }
static inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }
static inline uint32_t be32(const uint8_t *p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

typedef struct {
	const uint8_t *p;
	size_t len;
} span;

struct vgo_font {
	uint8_t *data;
	size_t len;
	span head, maxp, hhea, hmtx, loca, glyf, cmap, cff, cff2, fvar;
	int units_per_em, num_glyphs, num_hmetrics, loca_long;
	size_t loca_count;
};

/* table directory of face 0: the file, or the first face of a font collection ('ttcf'); (size_t)-1 for another magic
 * (ttf-parser RawFace::parse: 0x00010000, 'true', 'OTTO', 'ttcf') */
static size_t face_dir(const uint8_t *d, size_t len)
{
	if (len < 4)
		return (size_t)-1;
	uint32_t m = be32(d);
	if (m == 0x00010000u || m == 0x74727565u || m == 0x4F54544Fu)
		return 0;
	if (m != 0x74746366u || len < 16 || be32(d + 8) == 0)
		return (size_t)-1;
	/* the crate reads all numFonts offsets (a truncated array: no face) and subtracts the end of the array from the
	 * face's offset with checked_sub (a face inside header or array: no face) */
	uint64_t n_fonts = be32(d + 8);
	if (4 * n_fonts > len || !rd_ok(len, 12, (size_t)(4 * n_fonts)))
		return (size_t)-1;
	size_t at = be32(d + 12);
	if (at < 12 + 4 * n_fonts || !rd_ok(len, at, 4))
		return (size_t)-1;
	m = be32(d + at);
	return (m == 0x00010000u || m == 0x74727565u || m == 0x4F54544Fu) ? at : (size_t)-1;
}

static span find_table(const uint8_t *d, size_t len, const char *tag)
{
	span s = {NULL, 0};
	size_t dir = face_dir(d, len);
	if (dir == (size_t)-1 || !rd_ok(len, dir, 12))
		return s;
	int n = be16(d + dir + 4);
	for (int i = 0; i < n; i++) {
		size_t r = dir + 12 + (size_t)i * 16;
		if (!rd_ok(len, r, 16))
			break;
		if (memcmp(d + r, tag, 4) == 0) {
			size_t off = be32(d + r + 8), l = be32(d + r + 12);
			if (rd_ok(len, off, l)) {
				s.p = d + off;
				s.len = l;
			}
			return s;
		}
	}
	return s;
}

/* ttf_parser::Face::parse — only the tables the render path touches */
vgo_font *vgo_font_open(const uint8_t *data, size_t len)
{
	vgo_font *f = (vgo_font *)calloc(1, sizeof *f);
	f->data = (uint8_t *)malloc(len ? len : 1);
	memcpy(f->data, data, len);
	f->len = len;
	f->head = find_table(f->data, len, "head");
	f->maxp = find_table(f->data, len, "maxp");
	f->hhea = find_table(f->data, len, "hhea");
	f->hmtx = find_table(f->data, len, "hmtx");
	f->loca = find_table(f->data, len, "loca");
	f->glyf = find_table(f->data, len, "glyf");
	f->cmap = find_table(f->data, len, "cmap");
	f->cff = find_table(f->data, len, "CFF ");
	f->cff2 = find_table(f->data, len, "CFF2");
	f->fvar = find_table(f->data, len, "fvar");
	if (!f->head.p || f->head.len < 54 || !f->maxp.p || f->maxp.len < 6 || !f->hhea.p ||
	    f->hhea.len < 36) {
		vgo_font_close(f);
		return NULL;
	}
	f->units_per_em = be16(f->head.p + 18);
	if (f->units_per_em < 16 || f->units_per_em > 16384) { /* ttf-parser head.rs */
		vgo_font_close(f);
		return NULL;
	}
	f->loca_long = (int16_t)be16(f->head.p + 50) != 0;
	f->num_glyphs = be16(f->maxp.p + 4);
	f->num_hmetrics = be16(f->hhea.p + 34);
	if (f->loca.p) {
		size_t total = (size_t)f->num_glyphs + 1; /* ttf-parser loca.rs */
		if (f->num_glyphs == 0xFFFF)
			total = 0xFFFF;
		size_t actual = f->loca.len / (f->loca_long ? 4 : 2);
		f->loca_count = actual < total ? actual : total;
	}
	return f;
}

void vgo_font_close(vgo_font *f)
{
	if (!f)
		return;
	free(f->data);
	free(f);
}

int vgo_font_units_per_em(const vgo_font *f) { return f->units_per_em; }
int vgo_font_num_glyphs(const vgo_font *f) { return f->num_glyphs; }

/* ttf-parser hmtx.rs Table::advance (call site renderer.rs:115) */
int vgo_font_hor_advance(const vgo_font *f, int gid)
{
	if (!f->hmtx.p || f->num_hmetrics == 0 || f->num_glyphs == 0)
		return -1;
	if (gid < 0 || gid >= f->num_glyphs)
		return -1;
	size_t have = f->hmtx.len / 4;
	size_t nm = (size_t)f->num_hmetrics;
	if (have < nm)
		return -1; /* read_array16 fails -> no hmtx table */
	size_t idx = (size_t)gid < nm ? (size_t)gid : nm - 1; /* last long metric repeats */
	return be16(f->hmtx.p + idx * 4);
}

/* ------------------------------------------------------------------------------------
 * cmap (ttf-parser tables/cmap/ *.rs; call sites renderer.rs:106, metadata.rs:105-117)
 * ---------------------------------------------------------------------------------- */
typedef struct {
	int platform, encoding, format;
	span d; /* from the subtable offset to the end of the cmap table */
} cmap_sub;

static int cmap_subtable(const vgo_font *f, int i, cmap_sub *out)
{
	span c = f->cmap;
	if (!c.p || c.len < 4)
		return 0;
	int n = be16(c.p + 2);
	if (i >= n)
		return 0;
	size_t r = 4 + (size_t)i * 8;
	if (!rd_ok(c.len, r, 8))
		return 0;
	out->platform = be16(c.p + r);
	out->encoding = be16(c.p + r + 2);
	size_t off = be32(c.p + r + 4);
	out->format = -1;
	out->d.p = NULL;
	out->d.len = 0;
	if (off <= c.len && c.len - off >= 2) {
		out->d.p = c.p + off;
		out->d.len = c.len - off;
		out->format = be16(out->d.p);
	}
	return 1;
}

/* cmap Subtable::is_unicode */
static int sub_is_unicode(const cmap_sub *s)
{
	if (s->platform == 0)
		return 1;
	if (s->platform == 3 && s->encoding == 1)
		return 1;
	if (s->platform == 3)
		return s->encoding == 10 && (s->format == 12 || s->format == 13);
	return 0;
}

/* returns glyph id or -1 (None) */
static int sub_glyph_index(const cmap_sub *s, uint32_t cp)
{
	const uint8_t *d = s->d.p;
	size_t len = s->d.len;
	switch (s->format) {
	case 0: {
		if (len < 6 + 256 || cp >= 256)
			return -1;
		int g = d[6 + cp];
		return g != 0 ? g : -1;
	}
	case 4: { /* format4.rs: custom binary search over end codes */
		if (cp > 0xFFFF || len < 16)
			return -1;
		uint16_t c16 = (uint16_t)cp;
		unsigned segx2 = be16(d + 6);
		if (segx2 < 2)
			return -1;
		size_t seg = segx2 / 2;
		size_t end_off = 14, start_off = end_off + seg * 2 + 2, delta_off = start_off + seg * 2,
		       ro_off = delta_off + seg * 2;
		if (!rd_ok(len, ro_off, seg * 2))
			return -1;
		size_t lo = 0, hi = seg;
		while (hi > lo) {
			size_t idx = (lo + hi) / 2;
			uint16_t endv = be16(d + end_off + idx * 2);
			if (endv >= c16) {
				uint16_t startv = be16(d + start_off + idx * 2);
				if (startv > c16) {
					hi = idx;
				} else {
					uint16_t ro = be16(d + ro_off + idx * 2);
					int16_t delta = (int16_t)be16(d + delta_off + idx * 2);
					if (ro == 0)
						return (uint16_t)(c16 + (uint16_t)delta);
					if (ro == 0xFFFF)
						return -1;
					uint32_t dl = ((uint32_t)c16 - (uint32_t)startv) * 2;
					if (dl > 0xFFFF)
						return -1;
					uint16_t pos = (uint16_t)(ro_off + idx * 2);
					pos = (uint16_t)(pos + (uint16_t)dl);
					pos = (uint16_t)(pos + ro);
					if (!rd_ok(len, pos, 2))
						return -1;
					uint16_t gv = be16(d + pos);
					if (gv == 0)
						return -1;
					int16_t gid = (int16_t)((int16_t)gv + delta); /* wrapping i16 add */
					return gid < 0 ? -1 : gid;
				}
			} else {
				lo = idx + 1;
			}
		}
		return -1;
	}
	case 6: {
		if (cp > 0xFFFF || len < 10)
			return -1;
		unsigned first = be16(d + 6), count = be16(d + 8);
		if (cp < first)
			return -1;
		uint32_t idx = cp - first;
		if (idx >= count || !rd_ok(len, 10 + (size_t)idx * 2, 2))
			return -1;
		return be16(d + 10 + idx * 2);
	}
	case 10: {
		if (len < 20)
			return -1;
		uint32_t first = be32(d + 12), count = be32(d + 16);
		if (cp < first)
			return -1;
		uint32_t idx = cp - first;
		if (idx >= count || !rd_ok(len, 20 + (size_t)idx * 2, 2))
			return -1;
		return be16(d + 20 + (size_t)idx * 2);
	}
	case 12:
	case 13: {
		if (len < 16)
			return -1;
		uint32_t n = be32(d + 12);
		if (!rd_ok(len, 16, (size_t)n * 12))
			return -1;
		size_t lo = 0, hi = n;
		while (lo < hi) { /* binary_search_by on [start,end] ranges */
			size_t mid = lo + (hi - lo) / 2;
			const uint8_t *g = d + 16 + mid * 12;
			uint32_t sc = be32(g), ec = be32(g + 4), sg = be32(g + 8);
			if (sc > cp)
				hi = mid;
			else if (ec < cp)
				lo = mid + 1;
			else {
				uint64_t id = s->format == 12 ? (uint64_t)sg + cp - sc : sg;
				if (s->format == 12 && (uint64_t)sg + cp > 0xFFFFFFFFull)
					return -1;
				return id > 0xFFFF ? -1 : (int)id;
			}
		}
		return -1;
	}
	default:
		return -1; /* formats 2, 8, 14: not restated (absent from all fixtures) */
	}
}

typedef void (*cp_fn)(uint32_t cp, void *ctx);

/* cmap Subtable::codepoints */
static void sub_codepoints(const cmap_sub *s, cp_fn fn, void *ctx)
{
	const uint8_t *d = s->d.p;
	size_t len = s->d.len;
	switch (s->format) {
	case 0:
		if (len < 6 + 256)
			return;
		for (uint32_t i = 0; i < 256; i++)
			if (d[6 + i] != 0)
				fn(i, ctx);
		return;
	case 4: {
		if (len < 16)
			return;
		unsigned segx2 = be16(d + 6);
		if (segx2 < 2)
			return;
		size_t seg = segx2 / 2, end_off = 14, start_off = end_off + seg * 2 + 2;
		if (!rd_ok(len, start_off + seg * 4, seg * 2))
			return;
		for (size_t i = 0; i < seg; i++) {
			uint32_t st = be16(d + start_off + i * 2), en = be16(d + end_off + i * 2);
			if (st == en && st == 0xFFFF)
				break;
			for (uint32_t c = st; c <= en; c++)
				fn(c, ctx);
		}
		return;
	}
	case 6: {
		if (len < 10)
			return;
		uint32_t first = be16(d + 6), count = be16(d + 8);
		for (uint32_t i = 0; i < count && first + i <= 0xFFFF; i++)
			fn(first + i, ctx);
		return;
	}
	case 10: {
		if (len < 20)
			return;
		uint32_t first = be32(d + 12), count = be32(d + 16);
		for (uint32_t i = 0; i < count && first + i >= first; i++)
			fn(first + i, ctx);
		return;
	}
	case 12:
	case 13: {
		if (len < 16)
			return;
		uint32_t n = be32(d + 12);
		if (!rd_ok(len, 16, (size_t)n * 12))
			return;
		for (uint32_t i = 0; i < n; i++) {
			const uint8_t *g = d + 16 + (size_t)i * 12;
			uint32_t sc = be32(g), ec = be32(g + 4);
			for (uint64_t c = sc; c <= ec; c++)
				fn((uint32_t)c, ctx);
		}
		return;
	}
	default:
		return;
	}
}

/* Face::glyph_index: first unicode subtable that maps the code point */
int vgo_font_glyph_index(const vgo_font *f, uint32_t cp)
{
	cmap_sub s;
	for (int i = 0; cmap_subtable(f, i, &s); i++) {
		if (s.format < 0 || !sub_is_unicode(&s))
			continue;
		int g = sub_glyph_index(&s, cp);
		if (g >= 0)
			return g;
	}
	return -1;
}

typedef struct {
	const cmap_sub *s;
	uint8_t *bits; /* 0x110000 bits */
} cp_ctx;

static void cp_mark(uint32_t cp, void *vctx)
{
	cp_ctx *c = (cp_ctx *)vctx;
	if (cp >= 0x110000)
		return;
	if (sub_glyph_index(c->s, cp) >= 0) /* metadata.rs:111-113 */
		c->bits[cp >> 3] |= (uint8_t)(1u << (cp & 7));
}

/* FontMetadata::try_from, code point collection (metadata.rs:105-119) */
int vgo_font_codepoints(const vgo_font *f, uint32_t *out, int cap)
{
	uint8_t *bits = (uint8_t *)calloc(0x110000 / 8, 1);
	cmap_sub s;
	for (int i = 0; cmap_subtable(f, i, &s); i++) {
		if (s.format < 0 || !sub_is_unicode(&s))
			continue;
		cp_ctx c = {&s, bits};
		sub_codepoints(&s, cp_mark, &c);
	}
	int n = 0;
	for (uint32_t cp = 0; cp < 0x110000; cp++)
		if (bits[cp >> 3] & (1u << (cp & 7))) {
			if (out && n < cap)
				out[n] = cp;
			n++;
		}
	free(bits);
	return n;
}

/* ------------------------------------------------------------------------------------
 * glyf outline emission (ttf-parser tables/glyf.rs; call site renderer.rs:110)
 * ---------------------------------------------------------------------------------- */
typedef struct {
	vgo_cmd *out;
	int cap, n;
} cmd_sink;

static void emit(cmd_sink *k, int kind, float x1, float y1, float x, float y)
{
	if (k->n < k->cap) {
		vgo_cmd *c = &k->out[k->n];
		c->kind = kind;
		c->x1 = x1;
		c->y1 = y1;
		c->x2 = 0;
		c->y2 = 0;
		c->x = x;
		c->y = y;
	}
	k->n++;
}

typedef struct {
	float a, b, c, d, e, f;
} xform;

static const xform XF_ID = {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};

static int xf_is_default(const xform *t)
{
	return t->a == 1.0f && t->b == 0.0f && t->c == 0.0f && t->d == 1.0f && t->e == 0.0f &&
	       t->f == 0.0f;
}

/* Transform::combine — f32, each op rounded (no contraction) */
static xform xf_combine(xform t1, xform t2)
{
	xform r;
	r.a = t1.a * t2.a + t1.c * t2.b;
	r.b = t1.b * t2.a + t1.d * t2.b;
	r.c = t1.a * t2.c + t1.c * t2.d;
	r.d = t1.b * t2.c + t1.d * t2.d;
	r.e = t1.a * t2.e + t1.c * t2.f + t1.e;
	r.f = t1.b * t2.e + t1.d * t2.f + t1.f;
	return r;
}

static void xf_apply(const xform *t, float *x, float *y)
{
	float tx = *x, ty = *y;
	*x = t->a * tx + t->c * ty + t->e;
	*y = t->b * tx + t->d * ty + t->f;
}

typedef struct {
	cmd_sink *sink;
	xform t;
	int has_first_on, has_first_off, has_last_off;
	float fon_x, fon_y, foff_x, foff_y, loff_x, loff_y;
} builder;

static void b_move(builder *b, float x, float y)
{
	if (!xf_is_default(&b->t))
		xf_apply(&b->t, &x, &y);
	emit(b->sink, VGO_MOVE, 0, 0, x, y);
}
static void b_line(builder *b, float x, float y)
{
	if (!xf_is_default(&b->t))
		xf_apply(&b->t, &x, &y);
	emit(b->sink, VGO_LINE, 0, 0, x, y);
}
static void b_quad(builder *b, float x1, float y1, float x, float y)
{
	if (!xf_is_default(&b->t)) {
		xf_apply(&b->t, &x1, &y1);
		xf_apply(&b->t, &x, &y);
	}
	emit(b->sink, VGO_QUAD, x1, y1, x, y);
}

/* Point::lerp(other, t) = self + t*(other - self), f32 */
static inline float lerp_half(float a, float b) { return a + 0.5f * (b - a); }

/* glyf.rs Builder::push_point / finish_contour */
static void b_push_point(builder *b, float x, float y, int on_curve, int last)
{
	if (!b->has_first_on) {
		if (on_curve) {
			b->has_first_on = 1;
			b->fon_x = x;
			b->fon_y = y;
			b_move(b, x, y);
		} else if (b->has_first_off) {
			float mx = lerp_half(b->foff_x, x), my = lerp_half(b->foff_y, y);
			b->has_first_on = 1;
			b->fon_x = mx;
			b->fon_y = my;
			b->has_last_off = 1;
			b->loff_x = x;
			b->loff_y = y;
			b_move(b, mx, my);
		} else {
			b->has_first_off = 1;
			b->foff_x = x;
			b->foff_y = y;
		}
	} else if (b->has_last_off && on_curve) {
		b->has_last_off = 0;
		b_quad(b, b->loff_x, b->loff_y, x, y);
	} else if (b->has_last_off && !on_curve) {
		float ox = b->loff_x, oy = b->loff_y;
		b->loff_x = x;
		b->loff_y = y;
		b_quad(b, ox, oy, lerp_half(ox, x), lerp_half(oy, y));
	} else if (on_curve) {
		b_line(b, x, y);
	} else {
		b->has_last_off = 1;
		b->loff_x = x;
		b->loff_y = y;
	}

	if (last) {
		if (b->has_first_off && b->has_last_off) {
			b->has_last_off = 0;
			float mx = lerp_half(b->loff_x, b->foff_x), my = lerp_half(b->loff_y, b->foff_y);
			b_quad(b, b->loff_x, b->loff_y, mx, my);
		}
		if (b->has_first_on && b->has_first_off)
			b_quad(b, b->foff_x, b->foff_y, b->fon_x, b->fon_y);
		else if (b->has_first_on && b->has_last_off)
			b_quad(b, b->loff_x, b->loff_y, b->fon_x, b->fon_y);
		else if (b->has_first_on)
			b_line(b, b->fon_x, b->fon_y);
		b->has_first_on = b->has_first_off = b->has_last_off = 0;
		emit(b->sink, VGO_CLOSE, 0, 0, 0, 0);
	}
}

/* loca.rs glyph_range + glyf.rs get */
static int glyph_data(const vgo_font *f, int gid, span *out)
{
	if (!f->loca.p || !f->glyf.p || gid < 0 || gid == 0xFFFF)
		return 0;
	if ((size_t)gid + 1 >= f->loca_count)
		return 0;
	size_t a, b;
	if (f->loca_long) {
		a = be32(f->loca.p + (size_t)gid * 4);
		b = be32(f->loca.p + (size_t)gid * 4 + 4);
	} else {
		a = (size_t)be16(f->loca.p + (size_t)gid * 2) * 2;
		b = (size_t)be16(f->loca.p + (size_t)gid * 2 + 2) * 2;
	}
	if (a >= b || b > f->glyf.len)
		return 0;
	out->p = f->glyf.p + a;
	out->len = b - a;
	return 1;
}

/* glyf.rs parse_simple_outline + GlyphPointsIter; returns 0 on parse failure (abort) */
static int simple_outline(span g, int n_contours, builder *b)
{
	const uint8_t *d = g.p;
	size_t len = g.len, off = 0;
	if (!rd_ok(len, off, (size_t)n_contours * 2))
		return 0;
	const uint8_t *endpts = d;
	off += (size_t)n_contours * 2;
	unsigned last_end = be16(endpts + ((size_t)n_contours - 1) * 2);
	if (last_end == 0xFFFF)
		return 0; /* checked_add(1) */
	unsigned total = last_end + 1;
	if (total == 1)
		return 1; /* single point: empty iterator */
	if (!rd_ok(len, off, 2))
		return 0;
	unsigned ilen = be16(d + off);
	off += 2 + ilen; /* s.advance never fails; later reads do */
	if (off > len)
		return 0;
	/* resolve_coords_len */
	size_t flags_off = off, p = off;
	unsigned left = total;
	size_t xlen = 0, ylen = 0;
	while (left > 0) {
		if (!rd_ok(len, p, 1))
			return 0;
		uint8_t fl = d[p++];
		unsigned rep = 1;
		if (fl & 0x08) {
			if (!rd_ok(len, p, 1))
				return 0;
			rep = (unsigned)d[p++] + 1;
		}
		if (rep > left)
			return 0;
		if (fl & 0x02)
			xlen += rep;
		else if (!(fl & 0x10))
			xlen += (size_t)rep * 2;
		if (fl & 0x04)
			ylen += rep;
		else if (!(fl & 0x20))
			ylen += (size_t)rep * 2;
		left -= rep;
	}
	size_t x_off = p, y_off = x_off + xlen, y_end = y_off + ylen;
	if (y_end > len)
		return 0;
	/* iterate points */
	size_t fp = flags_off, xp = x_off, yp = y_off;
	uint8_t fl = 0;
	unsigned repeats = 0;
	int16_t x = 0, y = 0;
	/* EndpointsIter */
	unsigned ep_index = 1;
	unsigned ep_left = be16(endpts);
	for (unsigned i = 0; i < total; i++) {
		int last_point;
		if (ep_left == 0) {
			if (ep_index < (unsigned)n_contours) {
				unsigned end = be16(endpts + (size_t)ep_index * 2);
				unsigned prev = be16(endpts + (size_t)(ep_index - 1) * 2);
				ep_left = end > prev ? end - prev : 0;
				ep_left = ep_left > 0 ? ep_left - 1 : 0;
			}
			ep_index++;
			last_point = 1;
		} else {
			ep_left--;
			last_point = 0;
		}
		if (repeats == 0) {
			fl = fp < x_off ? d[fp++] : 0;
			if (fl & 0x08)
				repeats = fp < x_off ? d[fp++] : 0;
		} else {
			repeats--;
		}
		int16_t dx = 0, dy = 0;
		if (fl & 0x02) {
			int v = xp < y_off ? d[xp++] : 0;
			dx = (int16_t)((fl & 0x10) ? v : -v);
		} else if (!(fl & 0x10)) {
			if (xp + 2 <= y_off) {
				dx = (int16_t)be16(d + xp);
				xp += 2;
			}
		}
		if (fl & 0x04) {
			int v = yp < y_end ? d[yp++] : 0;
			dy = (int16_t)((fl & 0x20) ? v : -v);
		} else if (!(fl & 0x20)) {
			if (yp + 2 <= y_end) {
				dy = (int16_t)be16(d + yp);
				yp += 2;
			}
		}
		x = (int16_t)(x + dx); /* wrapping_add */
		y = (int16_t)(y + dy);
		b_push_point(b, (float)x, (float)y, fl & 0x01, last_point);
	}
	return 1;
}

/* glyf.rs outline_impl; returns 0 when ttf-parser would bail out with None */
static int outline_impl(const vgo_font *f, span g, int depth, cmd_sink *sink, xform t)
{
	if (depth >= MAX_COMPONENTS)
		return 0;
	if (g.len < 2)
		return 0;
	int16_t nc = (int16_t)be16(g.p);
	size_t off = 10; /* numberOfContours + bbox */
	if (nc > 0) {
		if (off > g.len)
			return 0;
		builder b;
		memset(&b, 0, sizeof b);
		b.sink = sink;
		b.t = t;
		span body = {g.p + off, g.len - off};
		return simple_outline(body, nc, &b);
	} else if (nc < 0) {
		if (off > g.len)
			return 0;
		const uint8_t *d = g.p + off;
		size_t len = g.len - off, p = 0;
		for (;;) { /* CompositeGlyphIter */
			if (!rd_ok(len, p, 4))
				break;
			unsigned flags = be16(d + p), cg = be16(d + p + 2);
			p += 4;
			xform ct = XF_ID;
			if (flags & 0x0002) { /* ARGS_ARE_XY_VALUES */
				if (flags & 0x0001) {
					if (!rd_ok(len, p, 4))
						break;
					ct.e = (float)(int16_t)be16(d + p);
					ct.f = (float)(int16_t)be16(d + p + 2);
					p += 4;
				} else {
					if (!rd_ok(len, p, 2))
						break;
					ct.e = (float)(int8_t)d[p];
					ct.f = (float)(int8_t)d[p + 1];
					p += 2;
				}
			} /* point-matching args are NOT skipped by ttf-parser 0.25 (parity unpinned) */
			if (flags & 0x0080) {
				if (!rd_ok(len, p, 8))
					break;
				ct.a = (float)(int16_t)be16(d + p) / 16384.0f;
				ct.b = (float)(int16_t)be16(d + p + 2) / 16384.0f;
				ct.c = (float)(int16_t)be16(d + p + 4) / 16384.0f;
				ct.d = (float)(int16_t)be16(d + p + 6) / 16384.0f;
				p += 8;
			} else if (flags & 0x0040) {
				if (!rd_ok(len, p, 4))
					break;
				ct.a = (float)(int16_t)be16(d + p) / 16384.0f;
				ct.d = (float)(int16_t)be16(d + p + 2) / 16384.0f;
				p += 4;
			} else if (flags & 0x0008) {
				if (!rd_ok(len, p, 2))
					break;
				ct.a = (float)(int16_t)be16(d + p) / 16384.0f;
				ct.d = ct.a;
				p += 2;
			}
			int more = (flags & 0x0020) != 0;
			span cgd;
			if (glyph_data(f, (int)cg, &cgd)) {
				if (!outline_impl(f, cgd, depth + 1, sink, xf_combine(t, ct)))
					return 0;
			}
			if (!more)
				break;
		}
		return 1;
	}
	return 1;
}

/* ------------------------------------------------------------------------------------
 * `CFF ` (version 1) outline emission — what ttf-parser's cff1 table does behind
 * Face::outline_glyph (call site renderer.rs:110) for OpenType fonts with Type 2 charstrings.
 * Restated from Adobe Technical Notes #5176 (CFF) and #5177 (Type 2 charstrings) and the crate's
 * observable rules: callbacks in f32, one f32 addition per coordinate in operand order, close()
 * before every further move_to and at endchar, the width operand taken once, at most 48 operands
 * and 10 nested calls, no outline for a glyph without points or with a bbox beyond i16.
 * PARITY UNPINNED: no fixture of the reference holds a CFF font; tests compare this reader, the
 * product's reader and fontTools on fonts synthesised with fontTools.
 * ---------------------------------------------------------------------------------- */
typedef struct {
	span tab;        /* the whole table */
	uint32_t count;  /* objects */
	int off_size;
	size_t offs, base; /* position of the offset array / of the byte before the object data */
} cff_index;

static uint32_t cff_off(const cff_index *ix, uint32_t i)
{
	uint32_t v = 0;
	for (int k = 0; k < ix->off_size; k++)
		v = (v << 8) | ix->tab.p[ix->offs + (size_t)i * ix->off_size + k];
	return v;
}

/* parses the INDEX at `at` (count of `cw` = 2 bytes, or 4 in CFF2); returns the position behind it, 0 on malformed data */
static size_t cff_index_wide(span t, size_t at, cff_index *ix, size_t cw)
{
	memset(ix, 0, sizeof *ix);
	ix->tab = t;
	if (!rd_ok(t.len, at, cw))
		return 0;
	ix->count = cw == 4 ? be32(t.p + at) : be16(t.p + at);
	if (ix->count == 0 || ix->count == 0xFFFFFFFFu) {
		ix->count = 0;
		return at + cw;
	}
	if (!rd_ok(t.len, at, cw + 1))
		return 0;
	ix->off_size = t.p[at + cw];
	if (ix->off_size < 1 || ix->off_size > 4)
		return 0;
	ix->offs = at + cw + 1;
	size_t n_off = ((size_t)ix->count + 1) * ix->off_size;
	if (!rd_ok(t.len, ix->offs, n_off))
		return 0;
	ix->base = ix->offs + n_off - 1;
	uint32_t last = cff_off(ix, ix->count);
	if (last < 1 || !rd_ok(t.len, ix->base + 1, (size_t)last - 1))
		return 0;
	return ix->base + last;
}

static size_t cff_index_at(span t, size_t at, cff_index *ix) { return cff_index_wide(t, at, ix, 2); }

static int cff_get(const cff_index *ix, uint32_t i, span *out)
{
	if (i >= ix->count)
		return 0;
	uint32_t a = cff_off(ix, i), b = cff_off(ix, i + 1);
	if (a < 1 || b < a || !rd_ok(ix->tab.len, ix->base + a, b - a))
		return 0;
	out->p = ix->tab.p + ix->base + a;
	out->len = b - a;
	return 1;
}

/* DICT walk: calls back with (operator, operands); two-byte operators are 1200 + second byte */
typedef struct {
	double v[48];
	int n;
} cff_operands;
typedef void (*cff_dict_fn)(int op, const cff_operands *a, void *ctx);

/* v2: the DICTs of a CFF2 table, where every byte that is not a number is an operator (22 vsindex, 23 blend, 24 vstore) */
static int cff_dict_any(span d, cff_dict_fn fn, void *ctx, int v2)
{
	cff_operands a;
	a.n = 0;
	size_t p = 0;
	while (p < d.len) {
		uint8_t b = d.p[p];
		if (b <= 21 || (v2 && (b <= 27 || b == 31 || b == 255))) {
			int op = b;
			p++;
			if (b == 12) {
				if (p >= d.len)
					return 0;
				op = 1200 + d.p[p++];
			}
			fn(op, &a, ctx);
			a.n = 0;
			continue;
		}
		double v = 0;
		if (b == 28) {
			if (!rd_ok(d.len, p, 3))
				return 0;
			v = (int16_t)be16(d.p + p + 1);
			p += 3;
		} else if (b == 29) {
			if (!rd_ok(d.len, p, 5))
				return 0;
			v = (int32_t)be32(d.p + p + 1);
			p += 5;
		} else if (b == 30) { /* real: skipped nibble by nibble */
			p++;
			for (;;) {
				if (p >= d.len)
					return 0;
				uint8_t q = d.p[p++];
				if ((q >> 4) == 0xF || (q & 0xF) == 0xF)
					break;
			}
		} else if (b >= 32 && b <= 246) {
			v = (int)b - 139;
			p++;
		} else if (b >= 247 && b <= 250) {
			if (!rd_ok(d.len, p, 2))
				return 0;
			v = ((int)b - 247) * 256 + d.p[p + 1] + 108;
			p += 2;
		} else if (b >= 251 && b <= 254) {
			if (!rd_ok(d.len, p, 2))
				return 0;
			v = -((int)b - 251) * 256 - d.p[p + 1] - 108;
			p += 2;
		} else {
			return 0;
		}
		if (a.n < 48)
			a.v[a.n++] = v;
	}
	return 1;
}

static int cff_dict(span d, cff_dict_fn fn, void *ctx) { return cff_dict_any(d, fn, ctx, 0); }

typedef struct {
	span tab;
	cff_index gsubrs, chars, fd_array;
	cff_index lsubrs;            /* name-keyed fonts */
	int have_lsubrs, is_cid;
	size_t charset, fd_select;
	size_t priv_size, priv_at;
	int have_priv, bad;
	size_t chars_at, fd_array_at;
	/* CFF2 */
	int v2, have_fd_array, have_vstore, n_coords;
	size_t vstore_at;
	span store;              /* the ItemVariationStore (behind its u16 length) */
	size_t regions_at;       /* of the VariationRegionList inside `store` */
	uint32_t n_axes, n_region_records, n_data;
} cff_font;

static void cff_top_op(int op, const cff_operands *a, void *vctx)
{
	cff_font *c = (cff_font *)vctx;
	double last = a->n ? a->v[a->n - 1] : 0;
	if (op == 15 && a->n == 1)
		c->charset = (size_t)last;
	else if (op == 17 && a->n == 1)
		c->chars_at = (size_t)last;
	else if (op == 18 && a->n == 2) {
		c->priv_size = (size_t)a->v[0];
		c->priv_at = (size_t)a->v[1];
		c->have_priv = 1;
	} else if (op == 1206 && a->n == 1 && last != 2.0)
		c->bad = 1;
	else if (op == 1230)
		c->is_cid = 1;
	else if (op == 1236 && a->n == 1)
		c->fd_array_at = (size_t)last;
	else if (op == 1237 && a->n == 1)
		c->fd_select = (size_t)last;
}

typedef struct {
	size_t subrs_rel;
	int have;
} cff_priv_ctx;

static void cff_priv_op(int op, const cff_operands *a, void *vctx)
{
	cff_priv_ctx *p = (cff_priv_ctx *)vctx;
	if (op == 19 && a->n == 1) {
		p->subrs_rel = (size_t)a->v[0];
		p->have = 1;
	}
}

/* local subroutines of a Private DICT at (at, size); 1 if an INDEX was found */
static int cff_local_subrs(span t, size_t at, size_t size, cff_index *out)
{
	if (!rd_ok(t.len, at, size))
		return 0;
	span d = {t.p + at, size};
	cff_priv_ctx pc = {0, 0};
	if (!cff_dict(d, cff_priv_op, &pc) || !pc.have)
		return 0;
	return cff_index_at(t, at + pc.subrs_rel, out) != 0;
}

static int cff_open(span t, cff_font *c)
{
	memset(c, 0, sizeof *c);
	c->tab = t;
	if (t.len < 4 || t.p[0] != 1)
		return 0;
	size_t p = t.p[2];
	cff_index names, tops, strings;
	if (!(p = cff_index_at(t, p, &names)) || !(p = cff_index_at(t, p, &tops)) || !(p = cff_index_at(t, p, &strings)) ||
	    !(p = cff_index_at(t, p, &c->gsubrs)))
		return 0;
	span top;
	if (!cff_get(&tops, 0, &top) || !cff_dict(top, cff_top_op, c) || c->bad)
		return 0;
	if (!c->chars_at || !cff_index_at(t, c->chars_at, &c->chars) || c->chars.count == 0)
		return 0;
	if (c->is_cid) {
		if (!c->fd_array_at || !c->fd_select || !cff_index_at(t, c->fd_array_at, &c->fd_array) || !rd_ok(t.len, c->fd_select, 1))
			return 0;
	} else if (c->have_priv) {
		c->have_lsubrs = cff_local_subrs(t, c->priv_at, c->priv_size, &c->lsubrs);
	}
	return 1;
}

/* ------------------------------------------------------------------------------------
 * `CFF2` — the crate's cff2 table (used when the face has neither glyf nor a readable `CFF `), at the variation
 * coordinates the reference leaves untouched: zero on every fvar axis, NO coordinates without fvar.
 * Restated from the OpenType specification (CFF2, ItemVariationStore) and the crate's observable rules
 * (see the product's csrc/host/cff.hpp for the list).  PARITY UNPINNED: no fixture; tests compare this reader,
 * the product's and fontTools at the default position.
 * ---------------------------------------------------------------------------------- */
static void cff2_top_op(int op, const cff_operands *a, void *vctx)
{
	cff_font *c = (cff_font *)vctx;
	int one = a->n == 1 && a->v[0] >= 0 && a->v[0] <= 4294967295.0;
	if (op == 17) {
		if (one)
			c->chars_at = (size_t)a->v[0];
		else
			c->bad = 1;
	} else if (op == 1236) {
		c->have_fd_array = one;
		c->fd_array_at = one ? (size_t)a->v[0] : 0;
	} else if (op == 24) {
		c->have_vstore = one;
		c->vstore_at = one ? (size_t)a->v[0] : 0;
	}
}

typedef struct {
	size_t a, b;
	int n_ops, done;
	int want_op, want_n;
} cff2_first_ctx;

/* remembers the operands of the FIRST occurrence of operator want_op (whether or not their number fits) */
static void cff2_first_op(int op, const cff_operands *a, void *vctx)
{
	cff2_first_ctx *f = (cff2_first_ctx *)vctx;
	if (f->done || op != f->want_op)
		return;
	f->done = 1;
	f->n_ops = 0;
	if (a->n != f->want_n)
		return;
	for (int i = 0; i < a->n; i++)
		if (!(a->v[i] >= 0 && a->v[i] <= 4294967295.0))
			return;
	f->a = (size_t)a->v[0];
	f->b = a->n > 1 ? (size_t)a->v[1] : 0;
	f->n_ops = a->n;
}

static int cff2_open(span t, int n_coords, cff_font *c)
{
	memset(c, 0, sizeof *c);
	c->tab = t;
	c->v2 = 1;
	c->n_coords = n_coords;
	if (t.len < 5 || t.p[0] != 2)
		return 0;
	size_t top_at = t.p[2] > 5 ? t.p[2] : 5, top_len = be16(t.p + 3);
	if (!rd_ok(t.len, top_at, top_len))
		return 0;
	span top = {t.p + top_at, top_len};
	(void)cff_dict_any(top, cff2_top_op, c, 1);
	if (c->bad || !c->chars_at)
		return 0;
	if (!cff_index_wide(t, top_at + top_len, &c->gsubrs, 4) || !cff_index_wide(t, c->chars_at, &c->chars, 4))
		return 0;
	if (c->have_vstore) {
		if (!rd_ok(t.len, c->vstore_at, 2))
			return 0;
		c->store.p = t.p + c->vstore_at + 2;
		c->store.len = t.len - c->vstore_at - 2;
		if (c->store.len < 8 || be16(c->store.p) != 1)
			return 0;
		c->regions_at = be32(c->store.p + 2);
		c->n_data = be16(c->store.p + 6);
		if (!rd_ok(c->store.len, 8, (size_t)c->n_data * 4) || !rd_ok(c->store.len, c->regions_at, 4))
			return 0;
		c->n_axes = be16(c->store.p + c->regions_at);
		c->n_region_records = c->n_axes * (uint32_t)be16(c->store.p + c->regions_at + 2);
		if (c->n_region_records > 0xFFFF || !rd_ok(c->store.len, c->regions_at + 4, (size_t)c->n_region_records * 6))
			return 0;
	}
	if (c->have_fd_array) {
		if (!cff_index_wide(t, c->fd_array_at, &c->fd_array, 4))
			return 0;
		for (uint32_t i = 0; i < c->fd_array.count; i++) {
			span fd;
			if (!cff_get(&c->fd_array, i, &fd))
				continue;
			cff2_first_ctx pv = {0, 0, 0, 0, 18, 2};
			(void)cff_dict_any(fd, cff2_first_op, &pv, 1);
			if (pv.n_ops != 2)
				continue;
			if (!rd_ok(t.len, pv.b, pv.a)) /* Private DICT (size, offset) outside the table: no table */
				return 0;
			span priv = {t.p + pv.b, pv.a};
			cff2_first_ctx sb = {0, 0, 0, 0, 19, 1};
			(void)cff_dict_any(priv, cff2_first_op, &sb, 1);
			if (sb.n_ops != 1)
				continue;
			if (!cff_index_wide(t, pv.b + sb.a, &c->lsubrs, 4))
				return 0;
			c->have_lsubrs = 1;
			break;
		}
	}
	return 1;
}

/* factors of the regions of ItemVariationData `index` at the default position -> out[0 .. *n); 0: no such subtable,
 * or more regions than the crate keeps (64) */
static int cff2_scalars(const cff_font *c, uint32_t index, float *out, int *n)
{
	if (!c->have_vstore || index >= c->n_data)
		return 0;
	const span st = c->store;
	size_t at = be32(st.p + 8 + (size_t)index * 4);
	if (at > st.len || !rd_ok(st.len, at + 4, 2))
		return 0;
	uint32_t cnt = be16(st.p + at + 4);
	if (cnt > 64 || !rd_ok(st.len, at + 6, (size_t)cnt * 2))
		return 0;
	for (uint32_t k = 0; k < cnt; k++) {
		uint32_t region = be16(st.p + at + 6 + (size_t)k * 2);
		float v = 1.0f;
		for (int i = 0; i < c->n_coords && v != 0.0f; i++) {
			uint32_t base = region * c->n_axes, rec = base + (uint32_t)i; /* 16-bit arithmetic in the crate: overflow = no record */
			if (base > 0xFFFF || rec > 0xFFFF || rec >= c->n_region_records) {
				v = 0.0f;
				break;
			}
			const uint8_t *q = st.p + c->regions_at + 4 + (size_t)rec * 6;
			int start = (int16_t)be16(q), peak = (int16_t)be16(q + 2), end = (int16_t)be16(q + 4);
			/* evaluate_axis at coordinate 0 */
			float f;
			if (start > peak || peak > end)
				f = 1.0f;
			else if (start < 0 && end > 0 && peak != 0)
				f = 1.0f;
			else if (peak == 0)
				f = 1.0f;
			else
				f = 0.0f; /* 0 <= start or end <= 0 */
			v = f == 0.0f ? 0.0f : v * f;
		}
		out[k] = v;
	}
	*n = (int)cnt;
	return 1;
}

typedef struct {
	size_t size, at;
	int have;
} cff_fd_ctx;

static void cff_fd_op(int op, const cff_operands *a, void *vctx)
{
	cff_fd_ctx *f = (cff_fd_ctx *)vctx;
	if (op == 18 && a->n == 2) {
		f->size = (size_t)a->v[0];
		f->at = (size_t)a->v[1];
		f->have = 1;
	}
}

/* local subroutines that apply to glyph gid; 0: none */
static int cff_glyph_lsubrs(const cff_font *c, int gid, cff_index *out)
{
	if (!c->is_cid) {
		*out = c->lsubrs;
		return c->have_lsubrs;
	}
	const uint8_t *s = c->tab.p + c->fd_select;
	size_t left = c->tab.len - c->fd_select;
	int fd = -1;
	if (s[0] == 0) {
		if (rd_ok(left, 1 + (size_t)gid, 1))
			fd = s[1 + gid];
	} else if (s[0] == 3 && left >= 3) {
		int n = be16(s + 1);
		for (int i = 0; i < n; i++) {
			size_t r = 3 + (size_t)i * 3;
			if (!rd_ok(left, r, 5))
				break;
			if (gid >= be16(s + r) && gid < be16(s + r + 3)) {
				fd = s[r + 2];
				break;
			}
		}
	}
	span dict;
	if (fd < 0 || !cff_get(&c->fd_array, (uint32_t)fd, &dict))
		return 0;
	cff_fd_ctx fc = {0, 0, 0};
	if (!cff_dict(dict, cff_fd_op, &fc) || !fc.have)
		return 0;
	return cff_local_subrs(c->tab, fc.at, fc.size, out);
}

/* StandardEncoding code -> SID -> glyph (seac); -1: none */
static int cff_std_glyph(const cff_font *c, int code)
{
	static const uint8_t lo[] = {32, 161, 177, 182, 191, 193, 202, 205}, hi[] = {126, 175, 180, 189, 191, 200, 203, 208};
	static const uint8_t sub[] = {31, 65, 66, 67, 68, 69, 70, 71};
	static const uint8_t single[][2] = {{225, 138}, {227, 139}, {232, 140}, {233, 141}, {234, 142}, {235, 143},
	                                    {241, 144}, {245, 145}, {248, 146}, {249, 147}, {250, 148}, {251, 149}};
	int sid = 0;
	for (size_t i = 0; i < sizeof lo; i++)
		if (code >= lo[i] && code <= hi[i])
			sid = code - sub[i];
	for (size_t i = 0; i < sizeof single / 2; i++)
		if (code == single[i][0])
			sid = single[i][1];
	if (!sid || c->is_cid)
		return -1;
	int n = (int)c->chars.count;
	if (c->charset == 0)
		return sid <= 228 && sid < n ? sid : -1;
	if (c->charset <= 2 || !rd_ok(c->tab.len, c->charset, 1))
		return -1;
	const uint8_t *s = c->tab.p + c->charset;
	size_t left = c->tab.len - c->charset;
	if (s[0] == 0) {
		for (int g = 1; g < n; g++) {
			if (!rd_ok(left, 1 + 2 * (size_t)(g - 1), 2))
				return -1;
			if (be16(s + 1 + 2 * (g - 1)) == sid)
				return g;
		}
		return -1;
	}
	if (s[0] != 1 && s[0] != 2)
		return -1;
	size_t rec = s[0] == 1 ? 3 : 4;
	int g = 1;
	for (size_t at = 1; g < n; at += rec) {
		if (!rd_ok(left, at, rec))
			return -1;
		int first = be16(s + at), cnt = s[0] == 1 ? s[at + 2] : be16(s + at + 2);
		if (sid >= first && sid <= first + cnt)
			return g + (sid - first) < n ? g + (sid - first) : -1;
		g += cnt + 1;
	}
	return -1;
}

typedef struct {
	const cff_font *font;
	cmd_sink *sink;
	cff_index lsubrs;
	int have_lsubrs;
	float st[513];
	int sp;
	float x, y;
	int moved, path_open, width_seen, ended;
	int had_vsindex, had_blend, n_scal; /* CFF2 */
	float scal[64];
	uint32_t stems;
	float bx0, by0, bx1, by1;
	int any_point;
} cff_run;

static void cff_emit(cff_run *r, int kind, float x1, float y1, float x2, float y2, float x, float y)
{
	cmd_sink *k = r->sink;
	if (k->n < k->cap) {
		vgo_cmd *c = &k->out[k->n];
		c->kind = kind;
		c->x1 = x1, c->y1 = y1, c->x2 = x2, c->y2 = y2, c->x = x, c->y = y;
	}
	k->n++;
}

static void cff_bb(cff_run *r, float x, float y)
{
	if (!r->any_point) {
		r->bx0 = r->bx1 = x;
		r->by0 = r->by1 = y;
		r->any_point = 1;
		return;
	}
	if (x < r->bx0) r->bx0 = x;
	if (x > r->bx1) r->bx1 = x;
	if (y < r->by0) r->by0 = y;
	if (y > r->by1) r->by1 = y;
}

static void cff_line(cff_run *r)
{
	cff_bb(r, r->x, r->y);
	cff_emit(r, VGO_LINE, 0, 0, 0, 0, r->x, r->y);
}

static void cff_curve(cff_run *r, float x1, float y1, float x2, float y2)
{
	cff_bb(r, x1, y1);
	cff_bb(r, x2, y2);
	cff_bb(r, r->x, r->y);
	cff_emit(r, VGO_CURVE, x1, y1, x2, y2, r->x, r->y);
}

/* six operands from a[0..5]: relative curve */
static void cff_rcurve(cff_run *r, const float *a)
{
	float x1 = r->x + a[0], y1 = r->y + a[1];
	float x2 = x1 + a[2], y2 = y1 + a[3];
	r->x = x2 + a[4];
	r->y = y2 + a[5];
	cff_curve(r, x1, y1, x2, y2);
}

static int cff_moveto(cff_run *r, int skip, int use_x, int use_y)
{
	if (r->sp != skip + use_x + use_y)
		return 0;
	if (r->path_open)
		cff_emit(r, VGO_CLOSE, 0, 0, 0, 0, 0, 0);
	r->path_open = 1;
	r->moved = 1;
	int i = skip;
	if (use_x)
		r->x += r->st[i++];
	if (use_y)
		r->y += r->st[i++];
	cff_bb(r, r->x, r->y);
	cff_emit(r, VGO_MOVE, 0, 0, 0, 0, r->x, r->y);
	r->sp = 0;
	return 1;
}

static int cff_exec(cff_run *r, span cs, int depth);

static int cff_component(cff_run *parent, int gid, float ox, float oy, int depth)
{
	span cs;
	if (gid < 0 || !cff_get(&parent->font->chars, (uint32_t)gid, &cs))
		return 0;
	cff_run sub;
	memset(&sub, 0, sizeof sub);
	sub.font = parent->font;
	sub.sink = parent->sink;
	sub.lsubrs = parent->lsubrs;
	sub.have_lsubrs = parent->have_lsubrs;
	sub.x = ox;
	sub.y = oy;
	sub.any_point = parent->any_point;
	sub.bx0 = parent->bx0, sub.by0 = parent->by0, sub.bx1 = parent->bx1, sub.by1 = parent->by1;
	if (!cff_exec(&sub, cs, depth + 1) || !sub.ended)
		return 0;
	parent->any_point = sub.any_point;
	parent->bx0 = sub.bx0, parent->by0 = sub.by0, parent->bx1 = sub.bx1, parent->by1 = sub.by1;
	return 1;
}

static int cff_exec(cff_run *r, span cs, int depth)
{
	size_t p = 0;
	while (p < cs.len) {
		uint8_t op = cs.p[p++];
		if (op == 28 || op >= 32) {
			float v;
			if (op == 28) {
				if (!rd_ok(cs.len, p, 2))
					return 0;
				v = (float)(int16_t)be16(cs.p + p);
				p += 2;
			} else if (op <= 246) {
				v = (float)((int)op - 139);
			} else if (op <= 250) {
				if (!rd_ok(cs.len, p, 1))
					return 0;
				v = (float)(((int)op - 247) * 256 + cs.p[p++] + 108);
			} else if (op <= 254) {
				if (!rd_ok(cs.len, p, 1))
					return 0;
				v = (float)(-((int)op - 251) * 256 - cs.p[p++] - 108);
			} else {
				if (!rd_ok(cs.len, p, 4))
					return 0;
				v = (float)(int32_t)be32(cs.p + p) / 65536.0f;
				p += 4;
			}
			if (r->sp >= (r->font->v2 ? 513 : 48))
				return 0;
			r->st[r->sp++] = v;
			continue;
		}
		float *s = r->st;
		int n = r->sp;
		switch (op) {
		case 1: case 3: case 18: case 23: /* stems */
			if ((n & 1) && !r->width_seen) {
				r->width_seen = 1;
				n--;
			}
			r->stems += (uint32_t)n >> 1;
			r->sp = 0;
			break;
		case 19: case 20: /* hintmask, cntrmask */
			r->sp = 0;
			if (n & 1) {
				n--;
				r->width_seen = 1;
			}
			r->stems += (uint32_t)n >> 1;
			p += (r->stems + 7) >> 3;
			if (p > cs.len) {
				if (!r->font->v2)
					return 0;
				p = cs.len;
			}
			break;
		case 21: { /* rmoveto */
			int skip = n == 3 && !r->width_seen; /* the width operand exists once per charstring */
			if (skip)
				r->width_seen = 1;
			if (!cff_moveto(r, skip, 1, 1))
				return 0;
			break;
		}
		case 22: { /* hmoveto */
			int skip = n == 2 && !r->width_seen;
			if (skip)
				r->width_seen = 1;
			if (!cff_moveto(r, skip, 1, 0))
				return 0;
			break;
		}
		case 4: { /* vmoveto */
			int skip = n == 2 && !r->width_seen;
			if (skip)
				r->width_seen = 1;
			if (!cff_moveto(r, skip, 0, 1))
				return 0;
			break;
		}
		case 5: /* rlineto */
			if (!r->moved || (n & 1))
				return 0;
			for (int i = 0; i < n; i += 2) {
				r->x += s[i];
				r->y += s[i + 1];
				cff_line(r);
			}
			r->sp = 0;
			break;
		case 6: case 7: { /* hlineto, vlineto: alternating */
			if (!r->moved || n == 0)
				return 0;
			int horizontal = op == 6;
			for (int i = 0; i < n; i++, horizontal = !horizontal) {
				if (horizontal)
					r->x += s[i];
				else
					r->y += s[i];
				cff_line(r);
			}
			r->sp = 0;
			break;
		}
		case 8: /* rrcurveto */
			if (!r->moved || n % 6)
				return 0;
			for (int i = 0; i < n; i += 6)
				cff_rcurve(r, s + i);
			r->sp = 0;
			break;
		case 24: { /* rcurveline */
			if (!r->moved || n < 8 || (n - 2) % 6)
				return 0;
			int i = 0;
			for (; i + 6 <= n - 2; i += 6)
				cff_rcurve(r, s + i);
			r->x += s[i];
			r->y += s[i + 1];
			cff_line(r);
			r->sp = 0;
			break;
		}
		case 25: { /* rlinecurve */
			if (!r->moved || n < 8 || ((n - 6) & 1))
				return 0;
			int i = 0;
			for (; i + 2 <= n - 6; i += 2) {
				r->x += s[i];
				r->y += s[i + 1];
				cff_line(r);
			}
			cff_rcurve(r, s + i);
			r->sp = 0;
			break;
		}
		case 26: case 27: { /* vvcurveto, hhcurveto */
			if (!r->moved)
				return 0;
			int i = 0, vertical = op == 26;
			if (n & 1) {
				if (vertical)
					r->x += s[0];
				else
					r->y += s[0];
				i = 1;
			}
			if ((n - i) % 4)
				return 0;
			for (; i < n; i += 4) {
				float x1, y1, x2, y2;
				if (vertical) {
					x1 = r->x, y1 = r->y + s[i];
					x2 = x1 + s[i + 1], y2 = y1 + s[i + 2];
					r->x = x2;
					r->y = y2 + s[i + 3];
				} else {
					x1 = r->x + s[i], y1 = r->y;
					x2 = x1 + s[i + 1], y2 = y1 + s[i + 2];
					r->x = x2 + s[i + 3];
					r->y = y2;
				}
				cff_curve(r, x1, y1, x2, y2);
			}
			r->sp = 0;
			break;
		}
		case 30: case 31: { /* vhcurveto, hvcurveto */
			if (!r->moved || n < 4)
				return 0;
			int i = 0, horizontal = op == 31;
			while (i < n) {
				int left = n - i;
				if (left < 4)
					return 0;
				float extra = left == 5 ? s[i + 4] : 0.0f, x1, y1, x2, y2;
				if (horizontal) {
					x1 = r->x + s[i], y1 = r->y;
					x2 = x1 + s[i + 1], y2 = y1 + s[i + 2];
					r->y = y2 + s[i + 3];
					r->x = x2 + extra;
				} else {
					x1 = r->x, y1 = r->y + s[i];
					x2 = x1 + s[i + 1], y2 = y1 + s[i + 2];
					r->x = x2 + s[i + 3];
					r->y = y2 + extra;
				}
				cff_curve(r, x1, y1, x2, y2);
				i += left == 5 ? 5 : 4;
				horizontal = !horizontal;
			}
			r->sp = 0;
			break;
		}
		case 10: case 29: { /* callsubr, callgsubr */
			if (n == 0 || depth == 10)
				return 0;
			const cff_index *ix = op == 29 ? &r->font->gsubrs : (r->have_lsubrs ? &r->lsubrs : NULL);
			if (!ix)
				return 0;
			float fi = s[--r->sp];
			long bias = ix->count < 1240 ? 107 : (ix->count < 33900 ? 1131 : 32768);
			long idx = (long)fi + bias;
			span sub;
			if ((float)(long)fi != fi || idx < 0 || !cff_get(ix, (uint32_t)idx, &sub) || !cff_exec(r, sub, depth + 1))
				return 0;
			if (r->ended)
				return p == cs.len;
			break;
		}
		case 15: { /* vsindex (CFF2) */
			if (!r->font->v2 || r->had_blend || r->had_vsindex || n != 1)
				return 0;
			if (!(s[0] >= 0.0f && s[0] <= 65535.0f) || !cff2_scalars(r->font, (uint32_t)s[0], r->scal, &r->n_scal))
				return 0;
			r->had_vsindex = 1;
			r->sp = 0;
			break;
		}
		case 16: { /* blend (CFF2): value_0..value_{m-1}, k deltas for each value, m */
			if (!r->font->v2 || n == 0)
				return 0;
			r->had_blend = 1;
			float fm = s[--r->sp];
			if (!(fm >= 0.0f && fm <= 65535.0f))
				return 0;
			size_t m = (size_t)fm, k = (size_t)r->n_scal, need = m * (k + 1);
			if ((size_t)r->sp < need)
				return 0;
			float *val = s + ((size_t)r->sp - need), *del = val + m;
			for (size_t i = m; i-- > 0;)       /* the crate pops: last value first, */
				for (size_t j = k; j-- > 0;)   /* last region first */
					val[i] += del[i * k + j] * r->scal[j];
			r->sp -= (int)(m * k);
			break;
		}
		case 11: /* return */
			return r->font->v2 ? 0 : 1;
		case 14: /* endchar */
			if (r->font->v2)
				return 0;
			if (n == 4 || (!r->width_seen && n == 5)) { /* seac: base glyph, then the accent at (adx, ady) */
				const float *a = s + (n - 4);
				r->width_seen = 1;
				r->sp = 0;
				if (depth == 10 || !(a[2] >= 0 && a[2] <= 255 && a[3] >= 0 && a[3] <= 255))
					return 0;
				if (r->path_open) {
					r->path_open = 0;
					cff_emit(r, VGO_CLOSE, 0, 0, 0, 0, 0, 0);
				}
				if (!cff_component(r, cff_std_glyph(r->font, (int)a[2]), 0.0f, 0.0f, depth) ||
				    !cff_component(r, cff_std_glyph(r->font, (int)a[3]), a[0], a[1], depth))
					return 0;
				r->ended = 1;
				return p == cs.len;
			}
			if (n == 1 && !r->width_seen)
				r->width_seen = 1;
			r->sp = 0;
			if (r->path_open) {
				r->path_open = 0;
				cff_emit(r, VGO_CLOSE, 0, 0, 0, 0, 0, 0);
			}
			r->ended = 1;
			return p == cs.len;
		case 12: {
			if (p >= cs.len || !r->moved)
				return 0;
			uint8_t op2 = cs.p[p++];
			float x0 = r->x, y0 = r->y;
			if (op2 == 35 && n == 13) { /* flex */
				cff_rcurve(r, s);
				cff_rcurve(r, s + 6);
			} else if (op2 == 34 && n == 7) { /* hflex */
				float a1[6] = {s[0], 0, s[1], s[2], s[3], 0};
				cff_rcurve(r, a1);
				float x1 = r->x + s[4], x2 = x1 + s[5];
				r->x = x2 + s[6];
				float yy = r->y;
				r->y = y0;
				cff_curve(r, x1, yy, x2, y0);
			} else if (op2 == 36 && n == 9) { /* hflex1 */
				float a1[6] = {s[0], s[1], s[2], s[3], s[4], 0};
				cff_rcurve(r, a1);
				float x1 = r->x + s[5], y1 = r->y, x2 = x1 + s[6], y2 = y1 + s[7];
				r->x = x2 + s[8];
				r->y = y0;
				cff_curve(r, x1, y1, x2, y2);
			} else if (op2 == 37 && n == 11) { /* flex1 */
				cff_rcurve(r, s);
				float x1 = r->x + s[6], y1 = r->y + s[7], x2 = x1 + s[8], y2 = y1 + s[9];
				if (fabsf(x2 - x0) > fabsf(y2 - y0)) {
					r->x = x2 + s[10];
					r->y = y0;
				} else {
					r->x = x0;
					r->y = y2 + s[10];
				}
				cff_curve(r, x1, y1, x2, y2);
			} else {
				return 0;
			}
			r->sp = 0;
			break;
		}
		default:
			return 0;
		}
	}
	return 1;
}

/* number of commands written to the sink (whatever ttf-parser's Option says: the reference ignores it) */
static void cff_outline(const vgo_font *f, int gid, cmd_sink *sink)
{
	cff_font c;
	span cs;
	if (!cff_open(f->cff, &c) || gid < 0 || !cff_get(&c.chars, (uint32_t)gid, &cs))
		return;
	cff_run r;
	memset(&r, 0, sizeof r);
	r.font = &c;
	r.sink = sink;
	r.have_lsubrs = cff_glyph_lsubrs(&c, gid, &r.lsubrs);
	(void)cff_exec(&r, cs, 0);
}

static void cff2_outline(const vgo_font *f, int gid, cmd_sink *sink)
{
	/* coordinates of the face: one per fvar axis (<= 64) when fvar reads (version 1.0, axes > 0, records inside) */
	int n_coords = 0;
	if (f->fvar.p && f->fvar.len >= 10 && be32(f->fvar.p) == 0x00010000u) {
		size_t at = be16(f->fvar.p + 4), n = be16(f->fvar.p + 8);
		if (n && rd_ok(f->fvar.len, at, n * 20))
			n_coords = n > 64 ? 64 : (int)n;
	}
	cff_font c;
	span cs;
	if (!cff2_open(f->cff2, n_coords, &c) || gid < 0 || !cff_get(&c.chars, (uint32_t)gid, &cs))
		return;
	cff_run r;
	memset(&r, 0, sizeof r);
	r.font = &c;
	r.sink = sink;
	r.lsubrs = c.lsubrs;
	r.have_lsubrs = 1; /* (an empty INDEX when the font has none: the call fails on the index) */
	r.width_seen = 1;  /* no width operand in CFF2 */
	if (!cff2_scalars(&c, 0, r.scal, &r.n_scal))
		return; /* the scalars of subtable 0 are loaded first: no store, no outline */
	(void)cff_exec(&r, cs, 0);
}

int vgo_font_outline(const vgo_font *f, int gid, vgo_cmd *out, int cap)
{
	cmd_sink sink = {out, cap, 0};
	span g;
	if (!(f->glyf.p && f->loca.p) && f->cff.p) { /* ttf-parser: glyf first, then cff, then cff2 */
		cff_font probe;
		if (cff_open(f->cff, &probe) || !f->cff2.p) {
			cff_outline(f, gid, &sink);
			return sink.n;
		}
	}
	if (!(f->glyf.p && f->loca.p) && f->cff2.p) {
		cff2_outline(f, gid, &sink);
		return sink.n;
	}
	if (!glyph_data(f, gid, &g))
		return 0;
	outline_impl(f, g, 0, &sink, XF_ID);
	return sink.n;
}

/* ------------------------------------------------------------------------------------
 * RingBuilder + flattening (render/ring_builder.rs, geometry/ring.rs, geometry/point.rs)
 * ---------------------------------------------------------------------------------- */
typedef struct {
	double *pts; /* x,y interleaved */
	int n, cap;
} ptvec;

static void pv_push(ptvec *v, double x, double y)
{
	if (v->n == v->cap) {
		v->cap = v->cap ? v->cap * 2 : 256;
		v->pts = (double *)realloc(v->pts, sizeof(double) * 2 * (size_t)v->cap);
	}
	v->pts[2 * v->n] = x;
	v->pts[2 * v->n + 1] = y;
	v->n++;
}

typedef struct {
	ptvec all;      /* points of saved rings, concatenated */
	int *ring_off;  /* n_rings+1 */
	int n_rings, ring_cap;
	ptvec ring;     /* the active ring */
} ringset;

/* ring_builder.rs:33-54 save_ring, ring.rs:53-63 close */
static void save_ring(ringset *rs)
{
	ptvec *r = &rs->ring;
	if (r->n < 3) {
		r->n = 0;
		return;
	}
	double fx = r->pts[0], fy = r->pts[1];
	double lx = r->pts[2 * (r->n - 1)], ly = r->pts[2 * (r->n - 1) + 1];
	if (fabs(fx - lx) > 2.220446049250313e-16 || fabs(fy - ly) > 2.220446049250313e-16)
		pv_push(r, fx, fy);
	if (r->n < 4) {
		r->n = 0;
		return;
	}
	if (rs->n_rings + 2 > rs->ring_cap) {
		rs->ring_cap = rs->ring_cap ? rs->ring_cap * 2 : 16;
		rs->ring_off = (int *)realloc(rs->ring_off, sizeof(int) * (size_t)rs->ring_cap);
	}
	if (rs->n_rings == 0)
		rs->ring_off[0] = 0;
	for (int i = 0; i < r->n; i++)
		pv_push(&rs->all, r->pts[2 * i], r->pts[2 * i + 1]);
	rs->n_rings++;
	rs->ring_off[rs->n_rings] = rs->all.n;
	r->n = 0;
}

typedef struct {
	double sx, sy, cx, cy, ex, ey;
} quad;
typedef struct {
	double sx, sy, ax, ay, bx, by, ex, ey;
} cubic;

/* ring.rs:119-144 add_quadratic_bezier — explicit LIFO stack, right half pushed first */
static void add_quad(ptvec *r, double sx, double sy, double cx, double cy, double ex, double ey,
                     double tol_sq)
{
	int cap = 64, n = 0;
	quad *st = (quad *)malloc(sizeof(quad) * (size_t)cap);
	st[n++] = (quad){sx, sy, cx, cy, ex, ey};
	while (n > 0) {
		quad q = st[--n];
		double dx = q.sx + q.ex - q.cx * 2.0;
		double dy = q.sy + q.ey - q.cy * 2.0;
		if (dx * dx + dy * dy <= tol_sq) {
			pv_push(r, q.ex, q.ey);
			continue;
		}
		double m1x = (q.sx + q.cx) / 2.0, m1y = (q.sy + q.cy) / 2.0; /* point.rs:29-31 */
		double m2x = (q.cx + q.ex) / 2.0, m2y = (q.cy + q.ey) / 2.0;
		double mx = (m1x + m2x) / 2.0, my = (m1y + m2y) / 2.0;
		if (n + 2 > cap) {
			cap *= 2;
			st = (quad *)realloc(st, sizeof(quad) * (size_t)cap);
		}
		st[n++] = (quad){mx, my, m2x, m2y, q.ex, q.ey};
		st[n++] = (quad){q.sx, q.sy, m1x, m1y, mx, my};
	}
	free(st);
}

/* ring.rs:159-187 add_cubic_bezier */
static void add_cubic(ptvec *r, double sx, double sy, double ax, double ay, double bx, double by,
                      double ex, double ey, double tol_sq)
{
	int cap = 64, n = 0;
	cubic *st = (cubic *)malloc(sizeof(cubic) * (size_t)cap);
	st[n++] = (cubic){sx, sy, ax, ay, bx, by, ex, ey};
	while (n > 0) {
		cubic c = st[--n];
		double dx = (c.bx + c.ax) - (c.sx + c.ex);
		double dy = (c.by + c.ay) - (c.sy + c.ey);
		if (dx * dx + dy * dy <= tol_sq) {
			pv_push(r, c.ex, c.ey);
			continue;
		}
		double p01x = (c.sx + c.ax) / 2.0, p01y = (c.sy + c.ay) / 2.0;
		double p12x = (c.ax + c.bx) / 2.0, p12y = (c.ay + c.by) / 2.0;
		double p23x = (c.bx + c.ex) / 2.0, p23y = (c.by + c.ey) / 2.0;
		double p012x = (p01x + p12x) / 2.0, p012y = (p01y + p12y) / 2.0;
		double p123x = (p12x + p23x) / 2.0, p123y = (p12y + p23y) / 2.0;
		double mx = (p012x + p123x) / 2.0, my = (p012y + p123y) / 2.0;
		if (n + 2 > cap) {
			cap *= 2;
			st = (cubic *)realloc(st, sizeof(cubic) * (size_t)cap);
		}
		st[n++] = (cubic){mx, my, p123x, p123y, p23x, p23y, c.ex, c.ey};
		st[n++] = (cubic){c.sx, c.sy, p01x, p01y, p012x, p012y, mx, my};
	}
	free(st);
}

/* ring_builder.rs:67-117 (OutlineBuilder impl) + :26-29 into_rings */
static void build_rings(const vgo_cmd *cmds, int n_cmds, ringset *rs)
{
	const double precision = 0.01; /* ring_builder.rs:62 — used as tolerance_sq */
	memset(rs, 0, sizeof *rs);
	for (int i = 0; i < n_cmds; i++) {
		const vgo_cmd *c = &cmds[i];
		switch (c->kind) {
		case VGO_MOVE:
			save_ring(rs);
			pv_push(&rs->ring, (double)c->x, (double)c->y);
			break;
		case VGO_LINE:
			pv_push(&rs->ring, (double)c->x, (double)c->y);
			break;
		case VGO_QUAD:
			if (rs->ring.n == 0)
				break;
			add_quad(&rs->ring, rs->ring.pts[2 * (rs->ring.n - 1)],
			         rs->ring.pts[2 * (rs->ring.n - 1) + 1], (double)c->x1, (double)c->y1,
			         (double)c->x, (double)c->y, precision);
			break;
		case VGO_CURVE:
			if (rs->ring.n == 0)
				break;
			add_cubic(&rs->ring, rs->ring.pts[2 * (rs->ring.n - 1)],
			          rs->ring.pts[2 * (rs->ring.n - 1) + 1], (double)c->x1, (double)c->y1,
			          (double)c->x2, (double)c->y2, (double)c->x, (double)c->y, precision);
			break;
		case VGO_CLOSE:
			save_ring(rs);
			break;
		}
	}
	save_ring(rs); /* into_rings */
}

static void ringset_free(ringset *rs)
{
	free(rs->all.pts);
	free(rs->ring.pts);
	free(rs->ring_off);
}

int vgo_build_rings(const vgo_cmd *cmds, int n_cmds, double *pts, int pts_cap, int *ring_off,
                    int ring_cap, int *n_pts_out)
{
	ringset rs;
	build_rings(cmds, n_cmds, &rs);
	int ret = rs.n_rings;
	if (n_pts_out)
		*n_pts_out = rs.all.n;
	if (rs.all.n > pts_cap || rs.n_rings + 1 > ring_cap) {
		ret = -(rs.all.n > 0 ? rs.all.n : 1);
	} else {
		memcpy(pts, rs.all.pts, sizeof(double) * 2 * (size_t)rs.all.n);
		if (rs.n_rings == 0)
			ring_off[0] = 0;
		else
			memcpy(ring_off, rs.ring_off, sizeof(int) * (size_t)(rs.n_rings + 1));
	}
	ringset_free(&rs);
	return ret;
}

/* Rust `f as i32`: truncate toward zero, saturating, NaN -> 0 */
static int32_t f64_as_i32(double v)
{
	if (v != v)
		return 0;
	if (v >= 2147483647.0)
		return INT32_MAX;
	if (v <= -2147483648.0)
		return INT32_MIN;
	return (int32_t)v;
}
static uint32_t f64_as_u32(double v)
{
	if (v != v || v <= 0.0)
		return 0;
	if (v >= 4294967295.0)
		return UINT32_MAX;
	return (uint32_t)v;
}

/* Renderer::render_glyph up to and including prepare_glyph (renderer.rs:103-137, 64-91).
 * On has_bitmap the scaled+shifted ring points stay in *rs_out for the caller. */
static int prepare(const vgo_font *f, uint32_t cp, vgo_glyph_info *info, ringset *rs_out)
{
	memset(info, 0, sizeof *info);
	memset(rs_out, 0, sizeof *rs_out);
	info->id = cp;
	if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) /* char::from_u32 (:104) */
		return 0;
	int gid = vgo_font_glyph_index(f, cp); /* :106 */
	if (gid < 0)
		return 0;
	double scale = (double)GLYPH_SIZE / (double)f->units_per_em; /* :107 */

	int ncmd = vgo_font_outline(f, gid, NULL, 0);
	vgo_cmd *cmds = (vgo_cmd *)malloc(sizeof(vgo_cmd) * (size_t)(ncmd ? ncmd : 1));
	vgo_font_outline(f, gid, cmds, ncmd);
	build_rings(cmds, ncmd, rs_out); /* :109-111 */
	free(cmds);

	int adv = vgo_font_hor_advance(f, gid);
	if (adv < 0)
		adv = 0;
	double advance_float = (double)adv * scale * 0.95; /* :115, left to right */
	uint32_t advance = f64_as_u32(round(advance_float)); /* :116 */
	info->advance = advance;

	if (rs_out->n_rings == 0) /* :118-120 */
		return 1;

	double *p = rs_out->all.pts;
	int np = rs_out->all.n;
	for (int i = 0; i < np; i++) { /* :122 rings.scale (point.rs:96-99) */
		p[2 * i] *= scale;
		p[2 * i + 1] *= scale;
	}
	double dx = ((double)advance - advance_float) / 2.0; /* :130 */
	for (int i = 0; i < np; i++) {                       /* :131 translate (point.rs:83-86) */
		p[2 * i] += dx;
		p[2 * i + 1] += 0.0;
	}
	/* prepare_glyph (:64-91); bbox.rs:26-31,56-58,64-69 */
	double minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
	for (int i = 0; i < np; i++) {
		minx = fmin(minx, p[2 * i]);
		miny = fmin(miny, p[2 * i + 1]);
		maxx = fmax(maxx, p[2 * i]);
		maxy = fmax(maxy, p[2 * i + 1]);
	}
	if (maxx <= minx && maxy <= miny)
		return 1; /* PbfGlyph::empty */
	int32_t x0 = f64_as_i32(floor(minx)) - BUFFER;
	int32_t y0 = f64_as_i32(floor(miny)) - BUFFER;
	int32_t x1 = f64_as_i32(ceil(maxx)) + BUFFER;
	int32_t y1 = f64_as_i32(ceil(maxy)) + BUFFER;
	info->x0 = x0;
	info->y0 = y0;
	info->w = (uint32_t)(x1 - x0);
	info->h = (uint32_t)(y1 - y0);
	info->has_bitmap = 1;
	int nseg = 0;
	for (int r = 0; r < rs_out->n_rings; r++)
		nseg += rs_out->ring_off[r + 1] - rs_out->ring_off[r] - 1;
	info->n_segments = nseg;
	/* :146 y1 -= 24; result.rs:66-76 into_pbf_glyph */
	y1 -= GLYPH_SIZE;
	info->width = info->w - 2 * BUFFER;
	info->height = info->h - 2 * BUFFER;
	info->left = x0 + BUFFER;
	info->top = y1 - BUFFER;
	return 1;
}

/* Rings::get_segments (rings.rs:75-81, ring.rs:97-104) -> AoS sx,sy,ex,ey */
static void rings_segments(const ringset *rs, double *segs)
{
	int k = 0;
	for (int r = 0; r < rs->n_rings; r++)
		for (int i = rs->ring_off[r]; i + 1 < rs->ring_off[r + 1]; i++) {
			segs[4 * k] = rs->all.pts[2 * i];
			segs[4 * k + 1] = rs->all.pts[2 * i + 1];
			segs[4 * k + 2] = rs->all.pts[2 * i + 2];
			segs[4 * k + 3] = rs->all.pts[2 * i + 3];
			k++;
		}
}

int vgo_prepare_glyph(const vgo_font *f, uint32_t cp, vgo_glyph_info *info, double *segs,
                      int seg_cap)
{
	ringset rs;
	int some = prepare(f, cp, info, &rs);
	if (some && info->has_bitmap && segs && info->n_segments <= seg_cap)
		rings_segments(&rs, segs);
	ringset_free(&rs);
	return some;
}

/* ------------------------------------------------------------------------------------
 * the SDF raster
 * ---------------------------------------------------------------------------------- */

/* Segment::squared_distance_to_point (segment.rs:54-72,96-99; point.rs:38-42) */
static inline double seg_dist_sq(double vx, double vy, double wx, double wy, double px, double py)
{
	double dx = wx - vx, dy = wy - vy;
	double l2 = dx * dx + dy * dy;
	double qx, qy;
	if (l2 == 0.0) {
		qx = vx;
		qy = vy;
	} else {
		double t = ((px - vx) * (wx - vx) + (py - vy) * (wy - vy)) / l2;
		if (t < 0.0) {
			qx = vx;
			qy = vy;
		} else if (t > 1.0) {
			qx = wx;
			qy = wy;
		} else {
			qx = vx + t * (wx - vx);
			qy = vy + t * (wy - vy);
		}
	}
	double ex = qx - px, ey = qy - py; /* p.squared_distance_to(&proj): other - self */
	return ex * ex + ey * ey;
}

typedef struct {
	double x;
	int sign;
} crossing;

static int cmp_crossing(const void *a, const void *b)
{
	double xa = ((const crossing *)a)->x, xb = ((const crossing *)b)->x;
	return (xa > xb) - (xa < xb);
}

/* renderer_precise (renderer_precise.rs:8-84) + min_distance_to_line_segment
 * (rtree_segments.rs:40-68).  mode VGO_PRECISE applies the R-tree's candidate rule
 * (segment AABB intersects the closed box [p-8, p+8]^2, rtree_segments.rs:25-31,47-53);
 * VGO_BRUTE takes the min over all segments. */
void vgo_sdf_render(const double *segs, int n, int x0i, int y0i, int w, int h, int mode,
                    uint8_t *out)
{
	if (mode == VGO_DUMMY) { /* renderer_dummy.rs:3-5 */
		memset(out, 0, (size_t)w * (size_t)h);
		return;
	}
	const double radius_by_256 = 256.0 / SDF_RADIUS; /* :25 */
	const double x0 = (double)x0i + 0.5, y0 = (double)y0i + 0.5; /* :27-28 */
	crossing *cr = (crossing *)malloc(sizeof(crossing) * (size_t)(n ? n : 1));
	double *aabb = NULL;
	if (mode == VGO_PRECISE) {
		aabb = (double *)malloc(sizeof(double) * 4 * (size_t)(n ? n : 1));
		for (int i = 0; i < n; i++) {
			const double *s = segs + 4 * i;
			aabb[4 * i] = fmin(s[0], s[2]);
			aabb[4 * i + 1] = fmin(s[1], s[3]);
			aabb[4 * i + 2] = fmax(s[0], s[2]);
			aabb[4 * i + 3] = fmax(s[1], s[3]);
		}
	}
	for (int y = 0; y < h; y++) {
		double py = (double)y + y0; /* :34 */
		int nc = 0;
		for (int i = 0; i < n; i++) { /* :41-51 */
			double sx = segs[4 * i], sy = segs[4 * i + 1], ex = segs[4 * i + 2],
			       ey = segs[4 * i + 3];
			if (sy <= py && ey > py) {
				double t = (py - sy) / (ey - sy);
				cr[nc].x = sx + t * (ex - sx);
				cr[nc++].sign = 1;
			} else if (sy > py && ey <= py) {
				double t = (py - sy) / (ey - sy);
				cr[nc].x = sx + t * (ex - sx);
				cr[nc++].sign = -1;
			}
		}
		qsort(cr, (size_t)nc, sizeof(crossing), cmp_crossing); /* :52 */
		int wn = 0, idx = 0;
		for (int x = 0; x < w; x++) {
			double px = (double)x + x0; /* :62 */
			while (idx < nc && cr[idx].x <= px) { /* :63-66 */
				wn -= cr[idx].sign;
				idx++;
			}
			int inside = wn != 0;
			double best = INFINITY;
			if (mode == VGO_PRECISE) {
				double qx0 = px - SDF_RADIUS, qx1 = px + SDF_RADIUS;
				double qy0 = py - SDF_RADIUS, qy1 = py + SDF_RADIUS;
				for (int i = 0; i < n; i++) {
					const double *a = aabb + 4 * i;
					if (a[0] <= qx1 && a[2] >= qx0 && a[1] <= qy1 && a[3] >= qy0) {
						const double *s = segs + 4 * i;
						double d2 = seg_dist_sq(s[0], s[1], s[2], s[3], px, py);
						if (d2 < best)
							best = d2;
					}
				}
			} else {
				for (int i = 0; i < n; i++) {
					const double *s = segs + 4 * i;
					double d2 = seg_dist_sq(s[0], s[1], s[2], s[3], px, py);
					if (d2 < best)
						best = d2;
				}
			}
			double d = sqrt(best); /* rtree_segments.rs:67 */
			if (inside)
				d = -d;
			d = d * radius_by_256 + CUTOFF; /* :75 */
			double nn = 255.0 - d;          /* :76 clamp(0,255) */
			if (nn < 0.0)
				nn = 0.0;
			if (nn > 255.0)
				nn = 255.0;
			out[(size_t)(h - 1 - y) * (size_t)w + (size_t)x] = (uint8_t)round(nn); /* :78-79 */
		}
	}
	free(cr);
	free(aabb);
}

int vgo_render_glyph(const vgo_font *f, uint32_t cp, int mode, vgo_glyph_info *info,
                     uint8_t *bitmap, size_t cap)
{
	ringset rs;
	int some = prepare(f, cp, info, &rs);
	int ret = some;
	if (some && info->has_bitmap) {
		size_t need = (size_t)info->w * info->h;
		if (need > cap) {
			ret = -1;
		} else {
			double *segs = (double *)malloc(sizeof(double) * 4 * (size_t)(info->n_segments + 1));
			rings_segments(&rs, segs);
			vgo_sdf_render(segs, info->n_segments, info->x0, info->y0, (int)info->w,
			               (int)info->h, mode, bitmap);
			free(segs);
		}
	}
	ringset_free(&rs);
	return ret;
}

/* ------------------------------------------------------------------------------------
 * PBF encoding (prost derives at protobuf/glyph.rs:10-41, fontstack.rs:9-25,
 * glyphs.rs:11-16)
 * ---------------------------------------------------------------------------------- */
static size_t varint_len(uint64_t v)
{
	size_t n = 1;
	while (v >= 0x80) {
		v >>= 7;
		n++;
	}
	return n;
}
static uint8_t *put_varint(uint8_t *p, uint64_t v)
{
	while (v >= 0x80) {
		*p++ = (uint8_t)(v | 0x80);
		v >>= 7;
	}
	*p++ = (uint8_t)v;
	return p;
}
static inline uint32_t zigzag32(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }

static size_t glyph_body_len(const vgo_glyph_info *g)
{
	size_t n = 1 + varint_len(g->id);
	if (g->has_bitmap) {
		size_t bl = (size_t)g->w * g->h;
		n += 1 + varint_len(bl) + bl;
	}
	n += 1 + varint_len(g->width) + 1 + varint_len(g->height);
	n += 1 + varint_len(zigzag32(g->left)) + 1 + varint_len(zigzag32(g->top));
	n += 1 + varint_len(g->advance);
	return n;
}

size_t vgo_pbf_encode(const char *name, uint32_t start, const vgo_glyph_info *glyphs,
                      const uint8_t *const *bitmaps, int n, uint8_t *out, size_t cap)
{
	char range[32];
	snprintf(range, sizeof range, "%u-%u", start, start + 255); /* glyph_block.rs:53-59 */
	size_t nl = strlen(name), rl = strlen(range);
	size_t stack = 1 + varint_len(nl) + nl + 1 + varint_len(rl) + rl;
	for (int i = 0; i < n; i++) {
		size_t gl = glyph_body_len(&glyphs[i]);
		stack += 1 + varint_len(gl) + gl;
	}
	size_t total = 1 + varint_len(stack) + stack;
	if (total > cap || !out)
		return total;
	uint8_t *p = out;
	*p++ = 0x0A;
	p = put_varint(p, stack);
	*p++ = 0x0A;
	p = put_varint(p, nl);
	memcpy(p, name, nl);
	p += nl;
	*p++ = 0x12;
	p = put_varint(p, rl);
	memcpy(p, range, rl);
	p += rl;
	for (int i = 0; i < n; i++) {
		const vgo_glyph_info *g = &glyphs[i];
		*p++ = 0x1A;
		p = put_varint(p, glyph_body_len(g));
		*p++ = 0x08;
		p = put_varint(p, g->id);
		if (g->has_bitmap) {
			size_t bl = (size_t)g->w * g->h;
			*p++ = 0x12;
			p = put_varint(p, bl);
			memcpy(p, bitmaps[i], bl);
			p += bl;
		}
		*p++ = 0x18;
		p = put_varint(p, g->width);
		*p++ = 0x20;
		p = put_varint(p, g->height);
		*p++ = 0x28;
		p = put_varint(p, zigzag32(g->left));
		*p++ = 0x30;
		p = put_varint(p, zigzag32(g->top));
		*p++ = 0x38;
		p = put_varint(p, g->advance);
	}
	return (size_t)(p - out);
}

/* ------------------------------------------------------------------------------------
 * GlyphBlock::render / FontWrapper::get_blocks / FontManager::render_glyphs
 * (font/glyph_block.rs:34-36,69-80; wrapper.rs:53-76; manager.rs:81-125)
 * ---------------------------------------------------------------------------------- */

/* provider[cp] = index of the first font (in precedence order) covering cp, or -1 */
static int16_t *build_providers(const vgo_font *const *fonts, int n_fonts)
{
	int16_t *prov = (int16_t *)malloc(sizeof(int16_t) * 0x10000);
	for (int i = 0; i < 0x10000; i++)
		prov[i] = -1;
	for (int fi = 0; fi < n_fonts; fi++) {
		int n = vgo_font_codepoints(fonts[fi], NULL, 0);
		uint32_t *cps = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
		vgo_font_codepoints(fonts[fi], cps, n);
		for (int i = 0; i < n; i++)
			if (cps[i] <= 0xFFFF && prov[cps[i]] < 0) /* wrapper.rs:66-71, glyph_block.rs:35 */
				prov[cps[i]] = (int16_t)fi;
		free(cps);
	}
	return prov;
}

static size_t render_block_prov(const vgo_font *const *fonts, const int16_t *prov,
                                const char *name, uint32_t start, int mode, uint8_t **out,
                                int *n_glyphs, uint64_t *n_pixels, uint64_t *n_segs)
{
	vgo_glyph_info infos[256];
	uint8_t *bitmaps[256];
	int n = 0;
	uint64_t pix = 0, segs = 0;
	for (uint32_t ci = 0; ci < 256; ci++) { /* canonical order: ascending id */
		uint32_t cp = start + ci;
		int fi = prov[cp];
		if (fi < 0)
			continue;
		vgo_glyph_info info;
		ringset rs;
		int some = prepare(fonts[fi], cp, &info, &rs);
		if (some) {
			bitmaps[n] = NULL;
			if (info.has_bitmap) {
				size_t need = (size_t)info.w * info.h;
				bitmaps[n] = (uint8_t *)malloc(need);
				double *sg = (double *)malloc(sizeof(double) * 4 * (size_t)(info.n_segments + 1));
				rings_segments(&rs, sg);
				vgo_sdf_render(sg, info.n_segments, info.x0, info.y0, (int)info.w, (int)info.h,
				               mode, bitmaps[n]);
				free(sg);
				pix += need;
				segs += (uint64_t)info.n_segments;
			}
			infos[n++] = info;
		}
		ringset_free(&rs);
	}
	size_t need = vgo_pbf_encode(name, start, infos, (const uint8_t *const *)bitmaps, n, NULL, 0);
	*out = (uint8_t *)malloc(need ? need : 1);
	vgo_pbf_encode(name, start, infos, (const uint8_t *const *)bitmaps, n, *out, need);
	int nb = 0;
	for (int i = 0; i < n; i++) {
		if (infos[i].has_bitmap)
			nb++;
		free(bitmaps[i]);
	}
	if (n_glyphs)
		*n_glyphs = n;
	if (n_pixels)
		*n_pixels = pix;
	if (n_segs)
		*n_segs = segs;
	(void)nb;
	return need;
}

size_t vgo_render_block(const vgo_font *const *fonts, int n_fonts, const char *name,
                        uint32_t start, int mode, uint8_t *out, size_t cap, int *n_glyphs,
                        uint64_t *n_pixels)
{
	int16_t *prov = build_providers(fonts, n_fonts);
	uint8_t *buf = NULL;
	size_t need = render_block_prov(fonts, prov, name, start, mode, &buf, n_glyphs, n_pixels, NULL);
	if (out && need <= cap)
		memcpy(out, buf, need);
	free(buf);
	free(prov);
	return need;
}

typedef struct {
	const vgo_font *const *fonts;
	const int16_t *prov;
	const char *name;
	int mode;
	int only_block;
	int next; /* block cursor, guarded by mu */
	pthread_mutex_t mu;
	uint64_t blocks, glyphs, pixels, segs, bytes;
	uint64_t block_hash[256];
} all_job;

static uint64_t fnv1a(uint64_t h, const uint8_t *p, size_t n)
{
	for (size_t i = 0; i < n; i++) {
		h ^= p[i];
		h *= 0x100000001b3ull;
	}
	return h;
}

static void *all_worker(void *arg)
{
	all_job *j = (all_job *)arg;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		int b = j->next++;
		pthread_mutex_unlock(&j->mu);
		if (b >= 256)
			break;
		uint32_t start = (uint32_t)b * 256;
		if (j->only_block >= 0 && (uint32_t)j->only_block != start)
			continue;
		uint8_t *buf = NULL;
		int ng = 0;
		uint64_t px = 0, sg = 0;
		size_t len = render_block_prov(j->fonts, j->prov, j->name, start, j->mode, &buf, &ng, &px, &sg);
		uint8_t st[4] = {(uint8_t)(start >> 24), (uint8_t)(start >> 16), (uint8_t)(start >> 8),
		                 (uint8_t)start};
		uint64_t h = fnv1a(0xcbf29ce484222325ull, st, 4);
		h = fnv1a(h, buf, len);
		free(buf);
		pthread_mutex_lock(&j->mu); /* the writer mutex of manager.rs:108-111 */
		j->blocks++;
		j->glyphs += (uint64_t)ng;
		j->pixels += px;
		j->segs += sg;
		j->bytes += len;
		j->block_hash[b] = h;
		pthread_mutex_unlock(&j->mu);
	}
	return NULL;
}

double vgo_render_all(const vgo_font *const *fonts, int n_fonts, const char *name, int mode,
                      int threads, int only_block, uint64_t counters[6])
{
	all_job j;
	memset(&j, 0, sizeof j);
	j.fonts = fonts;
	j.name = name;
	j.mode = mode;
	j.only_block = only_block;
	pthread_mutex_init(&j.mu, NULL);
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	int16_t *prov = build_providers(fonts, n_fonts);
	j.prov = prov;
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	pthread_t th[256];
	for (int i = 0; i < threads; i++)
		pthread_create(&th[i], NULL, all_worker, &j);
	for (int i = 0; i < threads; i++)
		pthread_join(th[i], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(prov);
	pthread_mutex_destroy(&j.mu);
	uint64_t h = 0xcbf29ce484222325ull;
	for (int b = 0; b < 256; b++)
		h = fnv1a(h, (const uint8_t *)&j.block_hash[b], 8);
	if (counters) {
		counters[0] = j.blocks;
		counters[1] = j.glyphs;
		counters[2] = j.pixels;
		counters[3] = j.segs;
		counters[4] = j.bytes;
		counters[5] = h;
	}
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------
 * batch raster (test + baseline helper; same per-glyph routine as above)
 * ---------------------------------------------------------------------------------- */
typedef struct {
	uint32_t n;
	const uint32_t *seg_off;
	const double *sx, *sy, *ex, *ey;
	const int32_t *x0, *y0;
	const uint32_t *w, *h;
	const uint64_t *out_off;
	int mode;
	uint8_t *out;
	uint32_t next;
	pthread_mutex_t mu;
} batch_job;

static void *batch_worker(void *arg)
{
	batch_job *j = (batch_job *)arg;
	double *segs = NULL;
	size_t cap = 0;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		uint32_t g0 = j->next;
		j->next += 4;
		pthread_mutex_unlock(&j->mu);
		if (g0 >= j->n)
			break;
		for (uint32_t g = g0; g < g0 + 4 && g < j->n; g++) {
			uint32_t a = j->seg_off[g], n = j->seg_off[g + 1] - a;
			if (n > cap) {
				cap = n;
				segs = (double *)realloc(segs, sizeof(double) * 4 * cap);
			}
			for (uint32_t i = 0; i < n; i++) {
				segs[4 * i] = j->sx[a + i];
				segs[4 * i + 1] = j->sy[a + i];
				segs[4 * i + 2] = j->ex[a + i];
				segs[4 * i + 3] = j->ey[a + i];
			}
			vgo_sdf_render(segs, (int)n, j->x0[g], j->y0[g], (int)j->w[g], (int)j->h[g], j->mode,
			               j->out + j->out_off[g]);
		}
	}
	free(segs);
	return NULL;
}

double vgo_sdf_render_batch(uint32_t n_glyphs, const uint32_t *seg_off, const double *sx,
                            const double *sy, const double *ex, const double *ey,
                            const int32_t *x0, const int32_t *y0, const uint32_t *w,
                            const uint32_t *h, const uint64_t *out_off, int mode, int threads,
                            uint8_t *out)
{
	batch_job j = {n_glyphs, seg_off, sx, sy, ex, ey, x0, y0, w, h, out_off, mode, out, 0,
	               PTHREAD_MUTEX_INITIALIZER};
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	pthread_t th[256];
	for (int i = 0; i < threads; i++)
		pthread_create(&th[i], NULL, batch_worker, &j);
	for (int i = 0; i < threads; i++)
		pthread_join(th[i], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
