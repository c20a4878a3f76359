/*
 * vgfont.h — flat C view of the C++ host façade in libvgsdf.so (namespace vg):
 * FontManager / GlyphBlock / Renderer with the reference's names and semantics, driving
 * the GPU raster of vgsdf.h.  Replaces, for the render path only:
 *   FontManager::{new,add_font_with_name,render_glyphs}  /root/reference/src/font/manager.rs:28,66,81
 *   FontManager::{add_path,add_paths}, scan              manager.rs:39-61, src/commands/recurse.rs:104-133
 *   write_index_json / write_families_json               manager.rs:128-138, src/font/index_files.rs:65-143
 *   Writer::{new_tar,new_file}                           src/writer/mod.rs:27-41, tar.rs:30-157, file.rs:10-52
 *   parse_font_name / FontMetadata::generate_name        src/font/parse_font_name.rs:214-291, metadata.rs:43-68
 *   GlyphBlock::render                                   src/font/glyph_block.rs:69
 *   Renderer::{new,new_precise,new_dummy,render_glyph}   src/render/renderer.rs:25-43,103
 * This header exists so tests/bench (Python ctypes) and non-C++ callers can reach the
 * façade; C++ callers include csrc/host/font_manager.hpp directly.
 *
 * Return convention: >= 0 success, < 0 failure with the message in vg_last_error()
 * (thread-local).  Renderer mode VG_MODE_HIP has no CPU fallback.
 */
#ifndef VGFONT_H
#define VGFONT_H
#include <stddef.h>
#include <stdint.h>

#include "vgsdf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vg_manager vg_manager;
typedef struct vg_renderer vg_renderer;
typedef struct vg_glyph_batch vg_glyph_batch;

enum { VG_MODE_HIP = 0, VG_MODE_DUMMY = 1 }; /* renderer.rs:14-17 RendererMode {Precise,Dummy} */

/* PbfGlyph (src/protobuf/glyph.rs:10-41) without the pixels */
typedef struct {
	uint32_t id;
	int32_t has_bitmap;
	uint32_t width, height;
	int32_t left, top;
	uint32_t advance;
	uint32_t bitmap_len;
} vg_pbf_glyph;

/* phases of the last vg_manager_render_glyphs call */
typedef struct {
	double tessellate_s, pack_s, device_s, encode_s, write_s, total_s;
	uint64_t blocks, glyphs, rasters, pixels, segments, pbf_bytes;
	uint64_t glyf_groups;    /* submissions whose glyphs the device decoded from their `glyf` arrays */
	uint64_t glyf_fallbacks; /* of which the device refused (a malformed entry) and the host's reader recorded again */
} vg_timings;

/* Writer sink (src/writer/mod.rs:10-19): is_dir=1 for write_directory. Return 0, or
 * non-zero to abort the render (first error aborts, manager.rs:117-121). */
typedef int (*vg_write_cb)(void *user, const char *path, const uint8_t *data, size_t len, int is_dir);

const char *vg_last_error(void);

vg_renderer *vg_renderer_new(int mode, int device_ordinal);
/* ONE process, n devices (the reference is one process with a rayon pool, src/font/manager.rs:81-125):
 * a HIP renderer with one lane per entry of `devices` (own device contexts and streams each; an entry may repeat a
 * device).  vg_manager_render_glyphs / _to with such a renderer deal the run's (font, block) tasks — manager.rs:86-97's
 * unit — to the lanes, longest first; every lane renders its tasks on its own host thread and every file comes whole from
 * one lane.  Where whole tasks do not balance (ONE font's 20-45 unequal blocks on 8 devices) the glyphs of the few heaviest
 * blocks are split between lanes and those blocks' partial PBFs merged in this process's memory (the hybrid plan;
 * vg_manager_plan_lanes shows it, vg_manager_set_lane_form chooses another form) — there is no exchange step.  Output bytes
 * equal a single-device run's in every form.  The run
 * counters {blocks, glyphs, pixels} are summed over the lanes with vgsdf_reduce_counters (RCCL all-reduce when the
 * devices are distinct; the host's own sum, flagged in vg_renderer_reduce_path, if RCCL fails) and checked; vg_manager_reduced_counters returns them.  NULL + vg_last_error() on failure. */
vg_renderer *vg_renderer_new_multi(const int *devices, int n);
int vg_renderer_device_count(const vg_renderer *r);
/* sum of the lanes' run counters as they stand (vgsdf_reduce_counters over the renderer's contexts) */
int vg_renderer_reduce_counters(const vg_renderer *r, uint64_t counters[3]);
/* how the last reduce (vg_renderer_reduce_counters, or the one at the end of a multi-lane vg_manager_render_glyphs) took
 * its sum: vgsdf_reduce_path of lane 0 — "rccl", "host: contexts share a device", "host: RCCL fallback: <reason>" ...
 * (valid until the next call on this thread) */
const char *vg_renderer_reduce_path(const vg_renderer *r);
void vg_renderer_add_counters(const vg_renderer *r, int lane, uint64_t blocks, uint64_t glyphs, uint64_t pixels);
void vg_renderer_reset_counters(const vg_renderer *r);
void vg_renderer_free(vg_renderer *r);

vg_manager *vg_manager_new(int parallel);
void vg_manager_free(vg_manager *m);
void vg_manager_set_threads(vg_manager *m, unsigned threads, unsigned blocks_per_batch);
/* 1: flatten / close / scale / bbox on the GPU (device front-end of vgsdf.h; HIP renderer
 * only), 0: on host threads.  Same bytes either way. */
void vg_manager_set_device_front_end(vg_manager *m, int on);
/* 1 (default): blocks are assembled IN PLACE — the raster stores every bitmap where its block's finished PBF has it, the
 * host writes the ~20 bytes around it (src/protobuf/glyphs.rs:66-70 without a second copy of the bitmaps); 0: bitmaps
 * packed back to back, blocks encoded afterwards.  Same bytes either way. */
void vg_manager_set_in_place_pbf(vg_manager *m, int on);
/* glyf fonts through the device front-end: 1 (default) = the glyphs' `glyf` arrays are copied as they stand and decoded on
 * the device (vgsdf_outlines_submit_glyf: what ttf-parser's Face::outline_glyph does for renderer.rs:110, replayed there),
 * 0 = the host's reader records the callbacks (CFF / CFF2 fonts always take that way; so does a batch in which the device
 * finds a malformed entry).  Same bytes either way. */
void vg_manager_set_glyf_on_device(vg_manager *m, int on);
/* How a renderer of several device lanes (vg_renderer_new_multi) splits a run: -1 / 2 (default) the hybrid plan — whole
 * (font, block) tasks per lane, manager.rs:86-97's unit, and the heaviest blocks' glyphs split between lanes until the lanes'
 * estimated raster cost is within 4 % of the mean; 1 whole tasks only; 0 glyph-level shards of every font (every block
 * merged afterwards).  Same bytes. */
void vg_manager_set_lane_form(vg_manager *m, int form);
int vg_manager_add_font_with_name(vg_manager *m, const char *name, const char *const *paths, int n_paths);
int vg_manager_add_font_data(vg_manager *m, const char *name, const uint8_t *data, size_t len);
/* manager.rs:39-53: the file's name table decides the font id (family/width/weight/style ->
 * generate_name -> name_to_id); files with the same id merge in call order. */
int vg_manager_add_path(vg_manager *m, const char *path);
/* recurse.rs:104-133 `scan`: *.ttf / *.otf files are added by add_path; a directory holding a
 * fonts.json ([{name, sources[]}], paths relative to it) is read through that file; other
 * directories are walked, entries in ascending byte order of their names (canonical order). */
int vg_manager_scan(vg_manager *m, const char *path);
/* font ids (ascending), '\n' separated, NUL terminated; returns the needed size incl. NUL */
long vg_manager_font_ids(const vg_manager *m, char *out, size_t cap);
/* raw family names (name id 1) of the files of one font id in wrapper order, '\n' separated */
long vg_manager_font_file_names(const vg_manager *m, const char *font_id, char *out, size_t cap);
/* parse_font_name(family, ps_name): style_out / width_out need 16 bytes each; returns the
 * length of the family (written NUL terminated into family_out, truncated to cap). */
int vg_parse_font_name(const char *family, const char *ps_name, char *family_out, size_t cap, char *style_out,
                       uint16_t *weight_out, char *width_out);
/* FontMetadata::generate_name of file `file_index` of a font id -> out; returns the needed size */
long vg_manager_generate_name(const vg_manager *m, const char *font_id, int file_index, char *out, size_t cap);
/* encode_codeblocks (index_files.rs:65-103); returns the needed size incl. NUL */
long vg_encode_codeblocks(const uint32_t *codepoints, size_t n, char *out, size_t cap);
/* build_index_json / build_font_families_json: returns the needed size (no NUL) */
long vg_manager_index_json(const vg_manager *m, uint8_t *out, size_t cap);
long vg_manager_families_json(const vg_manager *m, uint8_t *out, size_t cap);
/* name_to_id (manager.rs:141-147); writes a NUL-terminated id, returns its length */
int vg_name_to_id(const char *name, char *out, size_t cap);
/* number of code points (<= 0xFFFF) the font id maps after first-provider-wins merging */
int vg_manager_block_counts(const vg_manager *m, const char *font_id, uint32_t counts[256]);

/* Native sinks (src/writer): a ustar stream into a file or an open descriptor (e.g. 1 = stdout,
 * `recurse --tar`), or a directory tree.  mtime < 0 stamps every tar header with the wall clock as
 * the reference does (tar.rs:68-72); tests pass a fixed time to get reproducible bytes. */
typedef struct vg_writer vg_writer;
vg_writer *vg_writer_new_tar_path(const char *path, int64_t mtime);
vg_writer *vg_writer_new_tar_fd(int fd, int64_t mtime); /* the descriptor is NOT closed */
vg_writer *vg_writer_new_dir(const char *folder);
int vg_writer_write_file(vg_writer *w, const char *path, const uint8_t *data, size_t len);
int vg_writer_write_directory(vg_writer *w, const char *path);
int vg_writer_finish(vg_writer *w); /* idempotent (writer/mod.rs:71-77) */
void vg_writer_free(vg_writer *w);  /* finishes first, like Drop (mod.rs:84-96) */

int vg_manager_render_glyphs(vg_manager *m, vg_renderer *r, vg_write_cb cb, void *user);
/* the same into a native sink; index.json / font_families.json as manager.rs:128-138 */
int vg_manager_render_glyphs_to(vg_manager *m, vg_renderer *r, vg_writer *w);
int vg_manager_write_index_json(const vg_manager *m, vg_writer *w);
int vg_manager_write_families_json(const vg_manager *m, vg_writer *w);
/* Glyph-level shard of a font over `world` ranks: longest-processing-time-first on the
 * estimated raster cost w*h*N of every glyph (from its recorded outline: exact point counts of the
 * quadratic flattening, control-box area), identical on every rank.  owner[65536]: rank per code point,
 * 0xFF = unmapped; cost[65536] (may be NULL): the estimates. */
int vg_manager_shard_glyphs(const vg_manager *m, const char *font_id, uint32_t world, uint8_t *owner, double *cost);
/* The lane plan a run on `world` device lanes would use with the present lane form (1 or hybrid), for one font of the
 * manager: owner[65536] = lane per code point (0xFF = unmapped), *n_split_blocks = blocks of this font whose glyphs are
 * split between lanes, *est_max_over_mean = fullest lane / mean lane in the plan's own weights (all fonts of the manager).
 * Host arithmetic only (no device needed); the plan is kept until a font is added.  Any output pointer may be NULL. */
int vg_manager_plan_lanes(vg_manager *m, const char *font_id, uint32_t world, uint8_t *owner, uint32_t *n_split_blocks, double *est_max_over_mean);
/* From now on every render / build_batch / record_outlines call of this manager sees only the glyphs
 * that rank `rank` of `world` owns; every block is still emitted and its PBF holds this rank's glyphs
 * only (a partial).  world <= 1 switches sharding off. */
int vg_manager_set_glyph_shard(vg_manager *m, uint32_t rank, uint32_t world); /* -1: rank >= world or world > 254 */
/* Merges partial PBFs of ONE block (disjoint glyph subsets, same name and range) into the block's PBF:
 * glyphs in ascending id, byte for byte what a single process encodes.  Returns the needed size. */
long vg_pbf_merge(const uint8_t *const *parts, const size_t *lens, int n, uint8_t *out, size_t cap);
/* the same for parts that hold CONSECUTIVE runs of the block's code points, given in order (the split blocks of the hybrid
 * lane plan): header + the parts' entries as they are, no walk over the glyph messages; parts that are not in that form
 * are handed to vg_pbf_merge's code */
long vg_pbf_concat(const uint8_t *const *parts, const size_t *lens, int n, uint8_t *out, size_t cap);
/* A rank's shard: only the listed block starts (multiples of 256) of one font id. */
int vg_manager_render_blocks(vg_manager *m, vg_renderer *r, const char *font_id, const uint32_t *starts, int n,
                             vg_write_cb cb, void *user);
int vg_manager_timings(const vg_manager *m, vg_timings *out);
/* {blocks, glyphs, pixels} of the last render with a multi-device renderer, as reduced over its lanes (zeros otherwise) */
void vg_manager_reduced_counters(const vg_manager *m, uint64_t counters[3]);
/* GlyphBlock::render for one block of one font -> PBF bytes; returns needed size */
long vg_manager_render_block(vg_manager *m, vg_renderer *r, const char *font_id, uint32_t start, uint8_t *out,
                             size_t cap);

/* Renderer::render_glyph(face, index): 1 = Some, 0 = None.  file_index selects the file
 * inside the font id (wrapper.files order). */
int vg_render_glyph(vg_renderer *r, const vg_manager *m, const char *font_id, int file_index, uint32_t index,
                    vg_pbf_glyph *out, uint8_t *bitmap, size_t cap);

/* Host stage only (cmap -> outline -> flatten -> scale/shift -> bbox) for every glyph of a
 * font id: the SoA batch to hand to vgsdf_batch_upload.  Pointers in *view stay valid
 * until vg_glyph_batch_free.  ids[i] = code point of rasterised glyph i. */
vg_glyph_batch *vg_manager_build_batch(vg_manager *m, const char *font_id);
int vg_glyph_batch_view(const vg_glyph_batch *b, vgsdf_batch *view, const uint32_t **ids, uint32_t *n_jobs);
void vg_glyph_batch_free(vg_glyph_batch *b);

/* Host half of the DEVICE front-end for every glyph of a font id: the recorded outline
 * commands + scale / shift per glyph, i.e. the argument of vgsdf_outlines_prepare.  ids[i] /
 * advances[i] = code point / PBF advance of glyph i.  Valid until vg_outline_batch_free. */
typedef struct vg_outline_batch vg_outline_batch;
vg_outline_batch *vg_manager_record_outlines(const vg_manager *m, const char *font_id);
int vg_outline_batch_view(const vg_outline_batch *b, vgsdf_outlines *view, const uint32_t **ids, const uint32_t **advances);
void vg_outline_batch_free(vg_outline_batch *b);
/* The same for the device's glyf decoder (fonts whose glyphs all have `glyf` outlines; NULL otherwise): nothing is decoded
 * on the host, every glyph's simple glyphs are listed as parts with their arrays copied as they stand — the argument of
 * vgsdf_outlines_submit_glyf (pbf_pre / pbf_fix NULL).  Valid until vg_glyf_batch_free. */
typedef struct vg_glyf_batch vg_glyf_batch;
vg_glyf_batch *vg_manager_record_glyf_parts(const vg_manager *m, const char *font_id);
int vg_glyf_batch_view(const vg_glyf_batch *b, vgsdf_outlines_glyf *view, const uint32_t **ids, const uint32_t **advances);
void vg_glyf_batch_free(vg_glyf_batch *b);

/* Hand-encoder of the glyphs PBF (src/protobuf/glyphs.rs:66-70) for already rendered
 * glyphs; bitmaps[i] may be NULL when !has_bitmap. Returns needed size. */
long vg_pbf_encode(const char *name, const char *range, const vg_pbf_glyph *glyphs, const uint8_t *const *bitmaps,
                   int n, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
