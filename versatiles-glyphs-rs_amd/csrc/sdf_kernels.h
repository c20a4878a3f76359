// sdf_kernels.h — device-side data layout shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VGSDF_TILE_PIXELS 256

namespace vgsdf {

// One glyph's raster job, 32 bytes (the "+32 descriptor" of SURVEY.md §8d).
// (x0,y0,w,h) = RenderResult{x0,y0,width,height}, /root/reference/src/render/result.rs:7-29.
struct GlyphDesc {
	uint32_t seg_off; // first segment in the SoA arrays
	uint32_t n_seg;
	int32_t x0, y0;
	uint32_t w, h;
	uint64_t out_off; // first byte of this glyph's bitmap in the output buffer
};
static_assert(sizeof(GlyphDesc) == 32, "GlyphDesc must stay 32 bytes");

// Index of the box of chunk c (256 segments) of glyph g in the chunk-box table: (seg_off[g] >> 8) + g + c — unique for a
// monotone seg_off, so no offset table is needed; the table has (total segments >> 8) + n_glyphs + 1 entries.
__device__ __forceinline__ uint32_t chunk_box_index(uint32_t seg_off, uint32_t glyph, uint32_t c) { return (seg_off >> 8) + glyph + c; }

} // namespace vgsdf

// largest rows*(w+1) a tile of the filtered kernel may need (winding histogram in LDS)
extern "C" int vgsdf_filtered_delta_cap(void);

// 1 when vgsdf_launch_tiles of this build knows the kernel id (50 = span kernel, 1 = brute force; more
// only with -DVGSDF_DEV_VARIANTS)
extern "C" int vgsdf_kernel_known(int kernel);

// Segment layout: seg_stride 1 = four SoA arrays as the C ABI hands them (vgsdf_batch); 4 = records {sx, sy, ex, ey} of
// 32 bytes with sx / sy / ex / ey pointing at the four fields of record 0 (what the device front-end writes).
// tiles[i] = (glyph index, first output byte of the tile inside that glyph's bitmap)
// list_order != 0: workgroups take tiles in list order; 0: per-XCD contiguous remap
extern "C" int vgsdf_launch_tiles(int variant, int list_order, const vgsdf::GlyphDesc *glyphs,
                                  const uint2 *tiles, uint32_t n_tiles, const double *sx,
                                  const double *sy, const double *ex, const double *ey, uint32_t seg_stride,
                                  uint8_t *out, const void *boxes, hipStream_t stream);

// the span kernel over a device-planned work list that may still be in the making when the launch is enqueued (see .hip)
extern "C" int vgsdf_launch_span_planned(const vgsdf::GlyphDesc *glyphs, const uint2 *tiles, uint32_t grid, const double *sx,
                                         const double *sy, const double *ex, const double *ey, uint32_t seg_stride, uint8_t *out,
                                         const void *boxes, const void *plan, hipStream_t stream);

// chunk boxes of a resident batch (sdf_chunk_boxes): table size, and the preparation launch.  `boxes`
// may be NULL in vgsdf_launch_tiles (no chunk is skipped then); only the span kernel reads it.
extern "C" size_t vgsdf_chunk_box_bytes(uint64_t n_segments, uint32_t n_glyphs);
// guard (may be NULL): a vgsdf::PlanHeader on the device; the pass does nothing when it reports an error or more
// segments than seg_cap (device front-end: the segment arrays are then empty)
extern "C" int vgsdf_launch_chunk_boxes(const vgsdf::GlyphDesc *glyphs, uint32_t n_glyphs, const double *sx, const double *sy,
                                        const double *ex, const double *ey, uint32_t seg_stride, void *boxes,
                                        const void *guard, unsigned long long seg_cap, hipStream_t stream);
