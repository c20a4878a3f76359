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

} // namespace vgsdf

extern "C" {
size_t vgsdf_outline_scan_temp_bytes(uint32_t n);
int vgsdf_outline_scan(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t stream);
int vgsdf_outline_context(const vgsdf::OutlineCmd *cmds, const uint32_t *cmd_off, uint32_t n_glyphs, uint8_t *cmd_open,
                          hipStream_t stream);
int vgsdf_outline_count(const vgsdf::OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, uint32_t *counts,
                        hipStream_t stream);
int vgsdf_outline_emit(const vgsdf::OutlineCmd *cmds, const uint8_t *cmd_open, uint32_t n_cmds, const uint32_t *pt_off,
                       double *ptx, double *pty, void *cmd_box /* double4 per command */, hipStream_t stream);
int vgsdf_outline_rings(const vgsdf::OutlineCmd *cmds, const uint32_t *cmd_off, const uint32_t *pt_off, const double *ptx,
                        const double *pty, const double *scale, const double *shift_x, uint32_t n_glyphs,
                        vgsdf::RingRec *rings, uint32_t *cmd_ring, vgsdf::OutlineRect *rects, uint32_t *seg_count,
                        const void *cmd_box, hipStream_t stream);
int vgsdf_outline_segments(const uint32_t *pt_off, uint32_t n_cmds, uint32_t n_points, const uint32_t *cmd_ring,
                           const vgsdf::RingRec *rings, const vgsdf::OutlineRect *rects, const uint32_t *seg_off,
                           const double *ptx, const double *pty, const double *scale, const double *shift_x, double *sx,
                           double *sy, double *ex, double *ey, hipStream_t stream);
}
