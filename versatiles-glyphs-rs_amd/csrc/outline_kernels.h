// outline_kernels.h — data layout of the device front-end (outline commands -> segments).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vgsdf {

// ttf_parser::OutlineBuilder callbacks as recorded by the host (f32 font units)
enum : uint32_t { CMD_MOVE = 0, CMD_LINE = 1, CMD_QUAD = 2, CMD_CURVE = 3, CMD_CLOSE = 4 };

struct OutlineCmd {
	float x1, y1; // control point (quad) / first control (curve)
	float x2, y2; // second control (curve)
	float x, y;   // end point
	uint32_t kind;
};
static_assert(sizeof(OutlineCmd) == 28, "OutlineCmd layout");

// One ring (contour) of a glyph; stored at the index of the command that opened it.
struct RingRec {
	uint32_t pt_first, pt_count; // raw points [pt_first, pt_first + pt_count)
	uint32_t append;             // Ring::close appends a copy of the first point
	uint32_t accepted;           // survives RingBuilder::save_ring
	uint32_t seg_local;          // first segment of the ring inside its glyph
	uint32_t glyph;
};

// RenderResult of one glyph (/root/reference/src/render/result.rs:7-29) + segment count.
// Mirrors vgsdf_rect of include/vgsdf.h.
struct OutlineRect {
	int32_t x0, y0;
	uint32_t w, h;
	uint32_t n_segments;
	uint32_t has_raster; // 0 => PbfGlyph::empty (no rings / empty bbox)
};

// totals of a planned batch, read back by the host together with the rects
struct PlanHeader {
	unsigned long long n_segments; // sum of the rasterised glyphs' segments
	unsigned long long out_bytes;  // sum of w * h
	uint32_t n_spans;              // entries of the work list (may exceed the capacity it was planned with)
	uint32_t n_main;               // of which for the main kernel (they come first)
	uint32_t error;                // bit 0: absurd input (a glyph beyond 2^28 points / 2^32 pixels, more than 2^32 - 1 segments);
	                               // bit 1: a command kind that is none of the five callbacks; bit 2: a cubic broke its depth bound;
	                               // bit 3: dat_off does not match the kinds; bit 4: a malformed `glyf` entry (vgsdf_outlines_glyf)
	uint32_t ok;                   // the raster launch enqueued behind the plan may run: no error, everything within the
	                               // capacities and the grid it was planned against (outline_plan's last arguments)
};

struct GlyphDesc;

} // namespace vgsdf

extern "C" {
// parts: vgsdf_glyf_part records (include/vgsdf.h); writes the commands of every part into its slots of `cmds`;
// error_flag bit 4: a malformed entry
// max_cmd_cap / max_byte_len: the largest cmd_cap / byte_len among the parts (they size the launch's LDS)
// cmd_open (may be NULL): the context pass's byte per command, written by the decoder itself — only for batches whose scales
// are all positive and finite (bit 1 of that byte is never set here)
int vgsdf_glyf_decode(const void *parts, uint32_t n_parts, const uint8_t *bytes, vgsdf::OutlineCmd *cmds, uint32_t *error_flag,
                      uint32_t max_cmd_cap, uint32_t max_byte_len, uint8_t *cmd_open, hipStream_t stream);
// upload by a kernel: src_mapped = device address of a page-locked, device-mapped host block (16-byte aligned), dst 16-byte aligned
int vgsdf_copy_in(const void *src_mapped, void *dst, size_t bytes, hipStream_t stream);
// cmd_open: one byte per command (bit 0: ring open in front of it, bit 1: the glyph's scale is not positive finite)
// error_flag: one zeroed word; bit 1 is raised for a command kind that is none of the five callbacks
int vgsdf_outline_context(const vgsdf::OutlineCmd *cmds, const uint32_t *cmd_off, const double *scale, uint32_t n_glyphs,
                          uint8_t *cmd_open, uint32_t *error_flag, hipStream_t stream);
// the same for commands that arrive packed (one kind byte per command, coords: 2 / 4 / 6 / 0 floats per move or line / quad /
// curve / close, dat_off[n_glyphs + 1]: every glyph's range of coords); expands them into cmds_out; error_flag bit 3: the
// offsets do not match the kinds
int vgsdf_outline_context_packed(const uint8_t *kinds, const float *coords, const uint32_t *dat_off, const uint32_t *cmd_off,
                                 const double *scale, uint32_t n_glyphs, vgsdf::OutlineCmd *cmds_out, uint8_t *cmd_open,
                                 uint32_t *error_flag, hipStream_t stream);
// counts: one per command; cmd_box: double4 per command
// cmd_mask: one 64-bit word per command (cubics: first candidates of the leaves, read again by emit_segments);
// error_flag bit 2: a cubic broke its depth bound
int vgsdf_outline_count(const vgsdf::OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, const uint32_t *cmd_off,
                        uint32_t n_glyphs, const double *scale, const double *shift_x, uint32_t *counts, void *cmd_box,
                        unsigned long long *cmd_mask, uint32_t *error_flag, hipStream_t stream);
// pt_local: n_cmds + n_glyphs + 1 entries; error_flag: one zeroed word
int vgsdf_outline_rings(const vgsdf::OutlineCmd *cmds, const uint32_t *cmd_off, const uint8_t *cmd_open, const double *scale,
                        const double *shift_x, uint32_t n_glyphs, const uint32_t *counts, uint32_t *pt_local, const void *cmd_box,
                        vgsdf::RingRec *rings, uint32_t *cmd_ring, vgsdf::OutlineRect *rects, uint32_t *error_flag,
                        hipStream_t stream);
// seg_cap / out_cap / launch_spans: what PlanHeader::ok is decided against (capacity of the segment records, of the
// output buffer, grid of the raster launch enqueued behind the plan; launch_spans = 0: no such launch)
int vgsdf_outline_plan(const vgsdf::OutlineRect *rects, uint32_t n_glyphs, int span_list, uint32_t delta_cap, uint32_t span_max,
                       uint32_t span_budget, uint32_t tile_cap, vgsdf::GlyphDesc *descs, uint2 *tiles, vgsdf::PlanHeader *hdr,
                       const uint32_t *error_flag, unsigned long long seg_cap, unsigned long long out_cap, uint32_t launch_spans,
                       const uint32_t *pbf_pre /* NULL: bitmaps packed back to back */, const uint8_t *pbf_fix,
                       unsigned long long *pbf_at /* [n_glyphs] positions of the bitmaps in the arena */,
                       uint32_t *next_flag /* may be NULL: a word the kernel zeroes (the next submission's error word) */, hipStream_t stream);
int vgsdf_outline_emit_segments(const vgsdf::OutlineCmd *cmds, uint32_t n_cmds, const uint8_t *cmd_open, const double *scale,
                                const double *shift_x,
                                const uint32_t *pt_local, const vgsdf::RingRec *rings, const uint32_t *cmd_ring,
                                const vgsdf::GlyphDesc *descs, const vgsdf::PlanHeader *hdr, unsigned long long seg_cap,
                                double *seg /* records {sx, sy, ex, ey} */, const unsigned long long *cmd_mask,
                                // n_box_glyphs = n_glyphs: the first workgroups of the grid also fill the raster's chunk-box table
                                // (sdf_kernels.h, vgsdf_chunk_box_bytes) from the commands' boxes — boxes that CONTAIN the exact ones of
                                // vgsdf_launch_chunk_boxes; 0: no boxes (cmd_off / cmd_box / boxes unused)
                                uint32_t n_box_glyphs, const uint32_t *cmd_off, const void *cmd_box, void *boxes, hipStream_t stream);
}
