/*
 * vgsdf.h — C ABI of the MI355X SDF raster (libvgsdf.so).
 *
 * Drop-in boundary for the per-glyph hot loop of versatiles-glyphs-rs: one call renders
 * the signed-distance bitmaps of a whole batch of glyphs on the GPU, bit-exact with
 *   renderer_precise()               src/render/renderer_precise.rs:8-84
 *   min_distance_to_line_segment()   src/render/rtree_segments.rs:40-68
 *   Segment::squared_distance_to_point / project_point_on   src/geometry/segment.rs:54-99
 * which the reference runs once per glyph from Renderer::render_glyph
 * (src/render/renderer.rs:140-143, the `match self.mode` arm a new back-end slots into).
 * The caller is the GPU batch dispatcher that replaces the rayon block loop of
 * FontManager::render_glyphs (src/font/manager.rs:104-121); see vgfont.h.
 *
 * Plain C: pointers + sizes, no C++/torch types, no exceptions across the boundary.
 * Every entry point returns VGSDF_OK (0) or a negative vgsdf_status; the message is
 * available from vgsdf_last_error().  There is NO CPU fallback: without a usable HIP
 * device vgsdf_create() fails with VGSDF_E_HIP.
 *
 * Reference-side binding (Rust `extern "C"` block, untested here — no rustc in this
 * image): INTEGRATION.md.
 */
#ifndef VGSDF_H
#define VGSDF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	VGSDF_OK = 0,
	VGSDF_E_ARG = -1, /* NULL / inconsistent batch */
	VGSDF_E_HIP = -2, /* HIP runtime error (no device, launch failure, ...) */
	VGSDF_E_OOM = -3, /* device or pinned-host allocation failed */
	VGSDF_E_GLYF = -4 /* vgsdf_outlines_submit_glyf: a malformed `glyf` entry in the batch — record it with a host reader instead */
} vgsdf_status;

/* One context per (host thread, GPU): owns a HIP stream, device buffers and pinned
 * staging.  Not thread-safe; use one per thread. */
typedef struct vgsdf_ctx vgsdf_ctx;
/* A batch resident in HBM (segments, descriptors, tile list, output bitmaps). */
typedef struct vgsdf_dbatch vgsdf_dbatch;

/*
 * Host-side SoA batch, caller-owned, read-only during the call.
 *
 * Segments are what Rings::get_segments() yields (src/geometry/rings.rs:75-81) AFTER
 * rings.scale() and rings.translate() (renderer.rs:122,131): consecutive point pairs per
 * ring, rings in order, f64 pixel units.  Glyph g owns segments
 * [seg_off[g], seg_off[g+1]).  (x0,y0,w,h) is RenderResult{x0,y0,width,height}
 * (src/render/result.rs:7-29), i.e. including the 3 px buffer on every side.
 * Output bitmap of glyph g: out[out_off[g] + (h-1-y)*w + x], row-major, top row first —
 * exactly renderer_precise.rs:78.  out_off is ascending with out_off[g] + w[g]*h[g] <= out_off[g+1]: normally the prefix
 * sums of w*h (bitmaps packed back to back); a caller that assembles its output in place (the host façade lays the
 * bitmaps out where the finished PBF blocks have them) leaves gaps, whose bytes are not written.  out_off[n_glyphs] is
 * the size of the output buffer.
 */
typedef struct {
	uint32_t n_glyphs;
	const uint32_t *seg_off; /* [n_glyphs+1] prefix sums */
	const double *seg_sx;    /* [seg_off[n_glyphs]] */
	const double *seg_sy;
	const double *seg_ex;
	const double *seg_ey;
	const int32_t *x0; /* [n_glyphs] */
	const int32_t *y0;
	const uint32_t *w;
	const uint32_t *h;
	const uint64_t *out_off; /* [n_glyphs+1] ascending, see above */
} vgsdf_batch;

/* Per-launch statistics (filled by vgsdf_batch_stats). */
typedef struct {
	uint64_t n_glyphs;
	uint64_t n_segments;
	uint64_t n_pixels;
	uint64_t n_pairs;   /* sum over glyphs of w*h*n_segments (pixel x segment evaluations) */
	uint64_t n_tiles;   /* workgroups launched */
	uint64_t alg_bytes; /* 32*segments + 32*glyphs + pixels   */
} vgsdf_stats;

int vgsdf_device_count(void);
int vgsdf_create(int device_ordinal, vgsdf_ctx **out);
void vgsdf_destroy(vgsdf_ctx *ctx);
const char *vgsdf_last_error(const vgsdf_ctx *ctx); /* ctx may be NULL: last create error */

/* Page-locked host memory (hipHostMalloc).  Batch arrays / output buffers allocated with it
 * are DMA'd directly by vgsdf_render_batch / vgsdf_batch_upload / _download (no staging
 * copy).  Returns NULL when HIP is unavailable.  Optional: plain memory works too. */
void *vgsdf_host_alloc(size_t bytes);
void vgsdf_host_free(void *p);

/* Synchronous whole-batch render: H2D, kernel, D2H; out_bitmaps is host memory of
 * out_off[n_glyphs] bytes. */
int vgsdf_render_batch(vgsdf_ctx *ctx, const vgsdf_batch *in, uint8_t *out_bitmaps);

/* Split form, for callers that keep batches resident / overlap transfers:
 *   upload   : validates, builds the tile list, copies everything to HBM (async on the
 *              context stream, staged through pinned memory).
 *   launch   : enqueues the SDF kernel on the context stream (asynchronous).
 *   download : enqueues D2H of all bitmaps and waits for it.
 *   sync     : waits for the context stream. */
int vgsdf_batch_upload(vgsdf_ctx *ctx, const vgsdf_batch *in, vgsdf_dbatch **out);
int vgsdf_batch_launch(vgsdf_ctx *ctx, vgsdf_dbatch *b);
int vgsdf_batch_download(vgsdf_ctx *ctx, vgsdf_dbatch *b, uint8_t *out_bitmaps);
int vgsdf_batch_free(vgsdf_ctx *ctx, vgsdf_dbatch *b);
int vgsdf_sync(vgsdf_ctx *ctx);
int vgsdf_batch_stats(const vgsdf_dbatch *b, vgsdf_stats *out);

/* Times `iters` back-to-back launches of the SDF kernel with HIP events recorded on the
 * context stream (the stream the kernel runs on); *total_ms = elapsed for all of them. */
int vgsdf_batch_time(vgsdf_ctx *ctx, vgsdf_dbatch *b, int iters, float *total_ms);

/* Selects the kernel variant: 0 = default (bounded-group span kernel), 1 = brute force (every pixel
 * against every segment in f64; A/B reference).  Both are bit-exact with the reference.  Any other
 * value fails with VGSDF_E_ARG (development builds of the library, `make dev`, accept more ids for
 * kernel experiments; those are not part of the product).  Set it BEFORE uploading / preparing a
 * batch: the work list layout depends on it, and launching a resident batch under a different
 * variant fails with VGSDF_E_ARG. */
int vgsdf_set_variant(vgsdf_ctx *ctx, int variant);

/*
 * Device front-end: the host only records the OutlineBuilder callbacks of
 * every glyph (f32 font units); flattening (ring_builder.rs:62-110, ring.rs:119-187), ring
 * closing (ring.rs:53-63), scale / shift (renderer.rs:122-131), bbox + buffer
 * (renderer.rs:64-91) and Rings::get_segments run on the GPU, bit-exact with the host path.
 *   prepare : commands in -> per-glyph rects out (needed for the PBF metrics and to size the
 *             output); leaves segments, descriptors and tiles resident on the device.
 *   render  : SDF raster of the prepared batch; bitmaps of the glyphs with has_raster != 0
 *             are packed back to back in glyph order (w*h bytes each).
 */
typedef struct {
	float x1, y1;  /* quad control / first cubic control */
	float x2, y2;  /* second cubic control */
	float x, y;    /* end point */
	uint32_t kind; /* 0 move_to, 1 line_to, 2 quad_to, 3 curve_to, 4 close */
} vgsdf_outline_cmd;

typedef struct {
	uint32_t n_glyphs;
	const uint32_t *cmd_off;       /* [n_glyphs+1] prefix sums */
	const vgsdf_outline_cmd *cmds; /* [cmd_off[n_glyphs]] */
	const double *scale;           /* [n_glyphs] GLYPH_SIZE / units_per_em (renderer.rs:107) */
	const double *shift_x;         /* [n_glyphs] (advance - advance_float) / 2 (renderer.rs:130) */
} vgsdf_outlines;

typedef struct {
	int32_t x0, y0;      /* RenderResult.x0 / y0 (incl. -3 buffer) */
	uint32_t w, h;       /* RenderResult.width / height (incl. +6) */
	uint32_t n_segments;
	uint32_t has_raster; /* 0 => PbfGlyph::empty (no ring survived, or empty bbox) */
} vgsdf_rect;

int vgsdf_outlines_prepare(vgsdf_ctx *ctx, const vgsdf_outlines *in, vgsdf_rect *rects_out, uint64_t *out_bytes,
                           uint64_t *n_segments);
int vgsdf_outlines_render(vgsdf_ctx *ctx, uint8_t *out_bitmaps);
/* prepare + render as ONE submission: the raster is enqueued right behind the front-end kernels, before the host
 * has seen the sizes; the device checks the grid and every capacity it was launched against.  `out_bitmaps` holds
 * `out_capacity` bytes (from earlier batches, or a guess).  If it comes from vgsdf_host_alloc() the raster writes
 * the bitmaps straight into it (no copy, one synchronisation per call).  On return *out_bytes is the size needed;
 * *rendered = 1: the bitmaps are in out_bitmaps (packed in glyph order, as vgsdf_outlines_render leaves them);
 * *rendered = 0 (only when *out_bytes > out_capacity): the batch stays prepared — grow the buffer and call
 * vgsdf_outlines_render().  Results are identical to prepare + render. */
int vgsdf_outlines_render_into(vgsdf_ctx *ctx, const vgsdf_outlines *in, vgsdf_rect *rects_out, uint8_t *out_bitmaps,
                               size_t out_capacity, uint64_t *out_bytes, uint64_t *n_segments, int *rendered);
/* The same in two halves, for callers that overlap their own work with the device (one batch in flight per context;
 * use two contexts to keep the GPU busy while the host prepares the next batch and encodes the previous one):
 * submit enqueues the upload, the front-end and the raster and returns at once; wait synchronises and reports as
 * vgsdf_outlines_render_into does.  in->cmds and out_bitmaps must stay valid and untouched in between
 * (out_bitmaps may be NULL: no raster is enqueued, wait then equals vgsdf_outlines_prepare). */
int vgsdf_outlines_submit(vgsdf_ctx *ctx, const vgsdf_outlines *in, uint8_t *out_bitmaps, size_t out_capacity);
/* The same commands in their compact form for the upload: one kind byte per command and only the coordinates the kind
 * carries, in callback order (move_to / line_to: x y; quad_to: x1 y1 x y; curve_to: x1 y1 x2 y2 x y; close: none) —
 * about 12 bytes per command of a TrueType font instead of 28.  dat_off[g] .. dat_off[g + 1] is glyph g's range of
 * `coords` (dat_off[0] = 0); it must match the kinds (VGSDF_E_ARG otherwise).  Everything else as
 * vgsdf_outlines_submit; collect with vgsdf_outlines_wait.  When the arrays sit back to back in ONE block from
 * vgsdf_host_alloc() in the order scale | shift_x | cmd_off | dat_off | (pad to a multiple of 8 bytes) | coords | kinds
 * the library uploads them with a single copy. */
typedef struct {
	uint32_t n_glyphs;
	const uint32_t *cmd_off; /* [n_glyphs + 1] into kinds */
	const uint32_t *dat_off; /* [n_glyphs + 1] into coords */
	const uint8_t *kinds;    /* [cmd_off[n_glyphs]] 0..4 = move / line / quad / curve / close */
	const float *coords;     /* [dat_off[n_glyphs]] */
	const double *scale;     /* [n_glyphs] */
	const double *shift_x;   /* [n_glyphs] */
	/* In-place PBF assembly (both NULL: the bitmaps are packed back to back).  The output buffer becomes an ARENA of
	 * finished glyphs-PBF blocks (src/protobuf/glyphs.rs:66-70): the device lays the glyphs out as the `glyphs` entries
	 * of their fontstack message —
	 *   0x1A varint(len) | 0x08 varint(id) | [0x12 varint(w h) BITMAP] | 0x18 width 0x20 height 0x28 left 0x30 top 0x38 advance
	 * (glyph.rs:10-41; width = w - 6, height = h - 6, left = x0 + 3, top = y0 + h - 27: result.rs:66-76 after
	 * renderer.rs:146; a glyph without a raster is PbfGlyph::empty, glyph.rs:60-70) — one after the other, and the raster
	 * stores every BITMAP where the finished file has it; all other bytes are left for the caller, who knows id, advance and
	 * the block headers and gets w, h, x0, y0 back in the rects (the host façade writes them: csrc/host/pbf.hpp,
	 * write_pbf_entry_headers).  Glyph g starts at the running sum of what the glyphs before it occupy plus pbf_pre[g]:
	 *   pbf_pre[g]  bytes reserved in front of glyph g's entry (the file + fontstack header of the block it opens, else 0)
	 *   pbf_fix[g]  (1 + varint_len(id)) | (1 + varint_len(advance)) << 4
	 * *out_bytes of vgsdf_outlines_wait is the size of the arena.  In the single-copy block of vgsdf_host_alloc() the two
	 * arrays follow `kinds`: ... | kinds | (pad to a multiple of 4 bytes) | pbf_pre | pbf_fix. */
	const uint32_t *pbf_pre; /* [n_glyphs] or NULL */
	const uint8_t *pbf_fix;  /* [n_glyphs] or NULL */
} vgsdf_outlines_packed;
int vgsdf_outlines_submit_packed(vgsdf_ctx *ctx, const vgsdf_outlines_packed *in, uint8_t *out_bitmaps, size_t out_capacity);
/* The same front-end fed with the glyphs' `glyf` entries themselves: the host only looks glyphs up (cmap, loca, the
 * component records of composite glyphs) and copies bytes; the device replays ttf-parser's walk of every simple glyph
 * (glyf.rs parse_simple_outline + Builder: flag runs, short / repeated coordinates, wrapping i16 sums, implied on-curve
 * midpoints, the closing curve of a contour) and produces the callbacks Face::outline_glyph would deliver
 * (/root/reference/src/render/renderer.rs:110), then runs on as above.  One PART per simple glyph:
 *   bytes[byte_off .. +byte_len)  endPtsOfContours[n_contours] (big-endian u16, as in the font) followed by the entry's
 *                                 flags / xCoordinates / yCoordinates exactly as they stand in the font (the bytes from behind
 *                                 the instructions to the end of the entry); byte_off a multiple of 4
 *   a b c d e f                   the transform ttf-parser has accumulated for the component (x' = a x + c y + e,
 *                                 y' = b x + d y + f, in f32); plain = 1 for the identity (a simple glyph drawn as itself)
 *   cmd_at, cmd_cap               its command slots: cmd_cap >= points + 2 * contours of the entry (what its end points say);
 *                                 the parts tile [0, cmd_off[n_glyphs]) in order, glyph g owns [cmd_off[g], cmd_off[g + 1])
 * A glyph without outline has no part and no slots.  Slots a part does not need are filled with close() callbacks, which
 * do nothing on the empty ring behind a contour's own close().  An entry whose arrays do not fit its bytes or its slots
 * (ttf-parser returns None for such a glyph and, in a composite, skips the components behind it) fails the whole batch with
 * VGSDF_E_GLYF in vgsdf_outlines_wait: the caller records that batch with its host reader and submits commands instead.
 * In ONE block from vgsdf_host_alloc() in the order scale | shift_x | cmd_off | (pad to a multiple of 8 bytes) | parts | bytes
 * [| pbf_pre | pbf_fix] the batch is uploaded with a single copy — by a kernel reading the block itself (it is device-mapped), so no
 * copy-engine hand-over sits in front of the decoder — and its offsets are validated under that copy.  When every scale is positive
 * and finite and no part's slots straddle two glyphs the decoder also notes the ring state in front of every callback (what
 * ring_builder.rs:83-85,99-101 ask) and the separate context pass is skipped.  pbf_pre / pbf_fix as in vgsdf_outlines_packed. */
typedef struct {
	uint32_t byte_off, byte_len;
	uint32_t cmd_at, cmd_cap;
	uint32_t n_contours; /* > 0 */
	uint32_t plain;
	float a, b, c, d, e, f;
} vgsdf_glyf_part;
typedef struct {
	uint32_t n_glyphs, n_parts, n_bytes; /* n_bytes a multiple of 4 */
	const uint32_t *cmd_off;      /* [n_glyphs + 1] command slots, cmd_off[0] = 0 */
	const vgsdf_glyf_part *parts; /* [n_parts] */
	const uint8_t *bytes;         /* [n_bytes] */
	const double *scale;          /* [n_glyphs] */
	const double *shift_x;        /* [n_glyphs] */
	const uint32_t *pbf_pre;      /* [n_glyphs] or NULL */
	const uint8_t *pbf_fix;       /* [n_glyphs] or NULL */
} vgsdf_outlines_glyf;
int vgsdf_outlines_submit_glyf(vgsdf_ctx *ctx, const vgsdf_outlines_glyf *in, uint8_t *out_bitmaps, size_t out_capacity);
int vgsdf_outlines_wait(vgsdf_ctx *ctx, vgsdf_rect *rects_out, uint64_t *out_bytes, uint64_t *n_segments, int *rendered);
/* Between submit and wait: blocks until the front-end's results are on the host — they leave the device right behind the
 * plan kernel, on a stream of their own, while flattening and raster are still running — and reports the rects and
 * *out_bytes as vgsdf_outlines_wait will.  *in_place = 1: the raster enqueued with the submission is storing the bitmaps
 * straight into the caller's page-locked out_bitmaps and nothing else will touch that buffer, so the caller may write the
 * bytes BETWEEN the bitmaps (in-place PBF assembly: the headers) while it runs; 0: the bitmaps arrive only in
 * vgsdf_outlines_wait (pageable destination, a capacity guess that did not hold, a batch in error).  Optional. */
int vgsdf_outlines_peek(vgsdf_ctx *ctx, vgsdf_rect *rects_out, uint64_t *out_bytes, int *in_place);
/* after vgsdf_outlines_wait (or _peek) on a batch submitted with pbf_pre / pbf_fix: bitmap_at[g] = where in the arena the device placed
 * glyph g's bitmap (for a glyph without a raster: the byte behind its id field, where the `width` tag goes) */
int vgsdf_outlines_pbf_positions(vgsdf_ctx *ctx, uint64_t *bitmap_at);
/* test / inspection: download the segments the front-end produced (seg_off[n_glyphs+1]) */
int vgsdf_outlines_segments(vgsdf_ctx *ctx, uint32_t *seg_off, double *sx, double *sy, double *ex, double *ey);

/*
 * Multi-GPU: ONE process drives N devices, one context (+ its own host thread) per device; glyph batches
 * are independent (renderer.rs:103 has no shared state), so results need no exchange and the only collective of the
 * path is the sum of the run counters {blocks, glyphs, pixels} over the contexts.
 *   vgsdf_add_counters    : the dispatcher credits a context with the work it rendered there
 *   vgsdf_reduce_counters : counters[3] = sum over ctxs[0..n).  n >= 2 contexts on DISTINCT devices: an RCCL all-reduce
 *                           (sum, 3 x u64) over a communicator of exactly those devices, on the contexts' streams; every
 *                           rank's result is checked against the others.  RCCL is loaded at first use (dlopen of
 *                           librccl.so.1: no link-time dependency, shared with a host that already mapped it).  The
 *                           payload is 24 bytes the host already holds, so a finished render is never lost to the
 *                           collective: if RCCL cannot be loaded, cannot form the communicator or fails, the sum is taken
 *                           on the host, a line goes to stderr and vgsdf_reduce_path() says so.  One context, and contexts
 *                           that share a device (n lanes rehearsed on one GPU: RCCL refuses two ranks on one device), are
 *                           summed on the host.
 *   vgsdf_reduce_counters_rccl : the same through RCCL or not at all (n >= 1 distinct devices; VGSDF_E_HIP when the
 *                           collective is unavailable or fails, VGSDF_E_ARG when two contexts share a device) — for
 *                           callers and tests that want the failure instead of the fallback.
 *   vgsdf_reduce_path     : how the last reduce whose FIRST context was `ctx` took its sum: "rccl", "host: one context",
 *                           "host: contexts share a device" or "host: RCCL fallback: <reason>".
 * The all-reduce over distinct devices has not run on hardware yet (rounds 1-4 had one-GPU boxes; a one-rank communicator
 * has).  Replaces nothing in the reference (single process, rayon threads: src/font/manager.rs:81-125 counts nothing); it is
 * the north star's "RCCL only for the final block-count reduce".
 */
void vgsdf_add_counters(vgsdf_ctx *ctx, uint64_t blocks, uint64_t glyphs, uint64_t pixels);
void vgsdf_reset_counters(vgsdf_ctx *ctx);
int vgsdf_reduce_counters(vgsdf_ctx **ctxs, int n, uint64_t counters[3]);
int vgsdf_reduce_counters_rccl(vgsdf_ctx **ctxs, int n, uint64_t counters[3]);
const char *vgsdf_reduce_path(const vgsdf_ctx *ctx);

/* Raw device pointer of the resident output bitmaps (for zero-copy consumers on the same
 * device, e.g. a torch tensor wrapping it); valid until vgsdf_batch_free. */
void *vgsdf_batch_device_output(const vgsdf_dbatch *b);

#ifdef __cplusplus
}
#endif
#endif
