"""ctypes binding of include/vgfont.h: FontManager / Renderer / GlyphBlock façade (C++).

Mirrors the reference's names (src/font/manager.rs, src/render/renderer.rs) so tests read
like the reference's own tests.  HIP mode has no CPU fallback; Renderer.new_precise()
raises when no MI355X/HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from .device import OUTLINE_CMD_DTYPE, Batch, VgsdfError, _CBatch, _COutlines, load_library

MODE_HIP, MODE_DUMMY = 0, 1


class PbfGlyph(C.Structure):
    _fields_ = [("id", C.c_uint32), ("has_bitmap", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("left", C.c_int32), ("top", C.c_int32), ("advance", C.c_uint32), ("bitmap_len", C.c_uint32)]
    bitmap = None  # numpy u8 [height+6, width+6] or None

    def metrics(self):
        return (self.width, self.height, self.left, self.top, self.advance)


class Timings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("tessellate_s", "pack_s", "device_s", "encode_s", "write_s", "total_s")] + \
               [(k, C.c_uint64) for k in ("blocks", "glyphs", "rasters", "pixels", "segments", "pbf_bytes", "glyf_groups", "glyf_fallbacks")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_int)

VGFONT_SYMBOLS = [
    "vg_last_error", "vg_renderer_new", "vg_renderer_free", "vg_manager_new", "vg_manager_free",
    "vg_manager_set_threads", "vg_manager_set_device_front_end", "vg_manager_add_font_with_name", "vg_manager_add_font_data", "vg_manager_add_path",
    "vg_name_to_id", "vg_manager_block_counts", "vg_manager_render_glyphs", "vg_manager_timings",
    "vg_manager_render_block", "vg_manager_render_blocks", "vg_render_glyph", "vg_manager_build_batch", "vg_glyph_batch_view",
    "vg_glyph_batch_free", "vg_manager_record_outlines", "vg_outline_batch_view", "vg_outline_batch_free", "vg_pbf_encode",
    "vg_manager_record_glyf_parts", "vg_glyf_batch_view", "vg_glyf_batch_free",
    "vg_manager_scan", "vg_manager_font_ids", "vg_manager_font_file_names", "vg_parse_font_name", "vg_manager_generate_name",
    "vg_encode_codeblocks", "vg_manager_index_json", "vg_manager_families_json", "vg_writer_new_tar_path",
    "vg_writer_new_tar_fd", "vg_writer_new_dir", "vg_writer_write_file", "vg_writer_write_directory", "vg_writer_finish",
    "vg_writer_free", "vg_manager_render_glyphs_to", "vg_manager_write_index_json", "vg_manager_write_families_json",
    "vg_manager_shard_glyphs", "vg_manager_set_glyph_shard", "vg_pbf_merge", "vg_pbf_concat",
    "vg_renderer_new_multi", "vg_renderer_device_count", "vg_renderer_reduce_counters", "vg_renderer_reduce_path", "vg_renderer_add_counters",
    "vg_renderer_reset_counters", "vg_manager_reduced_counters", "vg_manager_set_in_place_pbf", "vg_manager_set_glyf_on_device", "vg_manager_set_lane_form", "vg_manager_plan_lanes",
]

_bound = False


def _L():
    global _bound
    L = load_library()
    if not _bound:
        vp = C.c_void_p
        L.vg_last_error.restype = C.c_char_p
        L.vg_renderer_new.restype = vp
        L.vg_renderer_new.argtypes = [C.c_int, C.c_int]
        L.vg_renderer_free.argtypes = [vp]
        L.vg_manager_new.restype = vp
        L.vg_manager_new.argtypes = [C.c_int]
        L.vg_manager_free.argtypes = [vp]
        L.vg_manager_set_threads.argtypes = [vp, C.c_uint, C.c_uint]
        L.vg_manager_set_device_front_end.argtypes = [vp, C.c_int]
        L.vg_manager_set_device_front_end.restype = None
        L.vg_manager_set_in_place_pbf.argtypes = [vp, C.c_int]
        L.vg_manager_set_in_place_pbf.restype = None
        L.vg_manager_set_glyf_on_device.argtypes = [vp, C.c_int]
        L.vg_manager_set_glyf_on_device.restype = None
        L.vg_manager_set_lane_form.argtypes = [vp, C.c_int]
        L.vg_manager_set_lane_form.restype = None
        L.vg_manager_plan_lanes.argtypes = [vp, C.c_char_p, C.c_uint32, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.vg_manager_add_font_with_name.argtypes = [vp, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
        L.vg_manager_add_font_data.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
        L.vg_manager_add_path.argtypes = [vp, C.c_char_p]
        L.vg_name_to_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.vg_manager_block_counts.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint32)]
        L.vg_manager_render_glyphs.argtypes = [vp, vp, WRITE_CB, vp]
        L.vg_manager_render_blocks.argtypes = [vp, vp, C.c_char_p, C.POINTER(C.c_uint32), C.c_int, WRITE_CB, vp]
        L.vg_manager_timings.argtypes = [vp, C.POINTER(Timings)]
        L.vg_manager_render_block.restype = C.c_long
        L.vg_manager_render_block.argtypes = [vp, vp, C.c_char_p, C.c_uint32, vp, C.c_size_t]
        L.vg_render_glyph.argtypes = [vp, vp, C.c_char_p, C.c_int, C.c_uint32, C.POINTER(PbfGlyph), vp, C.c_size_t]
        L.vg_manager_build_batch.restype = vp
        L.vg_manager_build_batch.argtypes = [vp, C.c_char_p]
        L.vg_glyph_batch_view.argtypes = [vp, C.POINTER(_CBatch), C.POINTER(C.POINTER(C.c_uint32)),
                                          C.POINTER(C.c_uint32)]
        L.vg_glyph_batch_free.argtypes = [vp]
        L.vg_manager_record_outlines.restype = vp
        L.vg_manager_record_outlines.argtypes = [vp, C.c_char_p]
        L.vg_outline_batch_view.argtypes = [vp, C.POINTER(_COutlines), C.POINTER(C.POINTER(C.c_uint32)),
                                            C.POINTER(C.POINTER(C.c_uint32))]
        L.vg_outline_batch_free.argtypes = [vp]
        L.vg_pbf_encode.restype = C.c_long
        L.vg_pbf_encode.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PbfGlyph), C.POINTER(C.c_void_p), C.c_int,
                                    vp, C.c_size_t]
        L.vg_manager_scan.argtypes = [vp, C.c_char_p]
        L.vg_manager_shard_glyphs.argtypes = [vp, C.c_char_p, C.c_uint32, vp, vp]
        L.vg_manager_set_glyph_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.vg_renderer_new_multi.restype = vp
        L.vg_renderer_new_multi.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.vg_renderer_device_count.argtypes = [vp]
        L.vg_renderer_reduce_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.vg_renderer_reduce_path.argtypes = [vp]
        L.vg_renderer_reduce_path.restype = C.c_char_p
        L.vg_renderer_add_counters.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
        L.vg_renderer_add_counters.restype = None
        L.vg_renderer_reset_counters.argtypes = [vp]
        L.vg_renderer_reset_counters.restype = None
        L.vg_manager_reduced_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.vg_manager_reduced_counters.restype = None
        L.vg_pbf_merge.restype = C.c_long
        L.vg_pbf_merge.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, vp, C.c_size_t]
        L.vg_pbf_concat.restype = C.c_long
        L.vg_pbf_concat.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, vp, C.c_size_t]
        for f in (L.vg_manager_font_ids, L.vg_manager_index_json, L.vg_manager_families_json):
            f.restype = C.c_long
            f.argtypes = [vp, vp, C.c_size_t]
        L.vg_manager_font_file_names.restype = C.c_long
        L.vg_manager_font_file_names.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.vg_parse_font_name.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint16),
                                         C.c_char_p]
        L.vg_manager_generate_name.restype = C.c_long
        L.vg_manager_generate_name.argtypes = [vp, C.c_char_p, C.c_int, vp, C.c_size_t]
        L.vg_encode_codeblocks.restype = C.c_long
        L.vg_encode_codeblocks.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.vg_writer_new_tar_path.restype = vp
        L.vg_writer_new_tar_path.argtypes = [C.c_char_p, C.c_int64]
        L.vg_writer_new_tar_fd.restype = vp
        L.vg_writer_new_tar_fd.argtypes = [C.c_int, C.c_int64]
        L.vg_writer_new_dir.restype = vp
        L.vg_writer_new_dir.argtypes = [C.c_char_p]
        L.vg_writer_write_file.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
        L.vg_writer_write_directory.argtypes = [vp, C.c_char_p]
        L.vg_writer_finish.argtypes = [vp]
        L.vg_writer_free.argtypes = [vp]
        L.vg_writer_free.restype = None
        L.vg_manager_render_glyphs_to.argtypes = [vp, vp, vp]
        L.vg_manager_write_index_json.argtypes = [vp, vp]
        L.vg_manager_write_families_json.argtypes = [vp, vp]
        _bound = True
    return L


def _err() -> str:
    return (_L().vg_last_error() or b"").decode()


def name_to_id(name: str) -> str:
    buf = C.create_string_buffer(1024)
    _L().vg_name_to_id(name.encode(), buf, len(buf))
    return buf.value.decode()


class Renderer:
    """src/render/renderer.rs Renderer; Precise == the HIP back-end."""

    def __init__(self, handle, mode):
        self._h, self.mode = handle, mode

    @classmethod
    def new(cls, dummy: bool, device: int = 0):
        return cls.new_dummy() if dummy else cls.new_precise(device)

    @classmethod
    def new_precise(cls, device: int = 0):
        h = _L().vg_renderer_new(MODE_HIP, device)
        if not h:
            raise VgsdfError(-2, _err())
        return cls(h, MODE_HIP)

    @classmethod
    def new_dummy(cls):
        return cls(_L().vg_renderer_new(MODE_DUMMY, 0), MODE_DUMMY)

    @classmethod
    def new_multi(cls, devices):
        """ONE process, one lane per entry of `devices` (an entry may repeat a device): FontManager.render_glyphs deals a
        font's glyphs to the lanes and merges their partial PBFs in this process (include/vgfont.h)."""
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = _L().vg_renderer_new_multi(devs, len(devices))
        if not h:
            raise VgsdfError(-2, _err())
        return cls(h, MODE_HIP)

    @property
    def n_devices(self) -> int:
        return int(_L().vg_renderer_device_count(self._h))

    def add_counters(self, lane: int, blocks: int, glyphs: int, pixels: int):
        _L().vg_renderer_add_counters(self._h, lane, blocks, glyphs, pixels)

    def reset_counters(self):
        _L().vg_renderer_reset_counters(self._h)

    def reduce_counters(self):
        """(blocks, glyphs, pixels) summed over the lanes: vgsdf_reduce_counters (RCCL when the devices are distinct)"""
        out = (C.c_uint64 * 3)()
        if _L().vg_renderer_reduce_counters(self._h, out) != 0:
            raise RuntimeError(_err())
        return tuple(int(v) for v in out)

    def reduce_path(self) -> str:
        """how the last reduce took its sum: "rccl", "host: contexts share a device", "host: RCCL fallback: <reason>" ..."""
        return (_L().vg_renderer_reduce_path(self._h) or b"").decode()

    def render_glyph(self, manager: "FontManager", font_id: str, index: int, file_index: int = 0):
        """Renderer::render_glyph(&face, index) -> PbfGlyph | None"""
        g = PbfGlyph()
        buf = np.empty(1 << 20, dtype=np.uint8)
        rc = _L().vg_render_glyph(self._h, manager._h, font_id.encode(), file_index, index, C.byref(g),
                                  buf.ctypes.data, buf.size)
        if rc < 0:
            raise RuntimeError(_err())
        if rc == 0:
            return None
        if g.has_bitmap:
            g.bitmap = buf[:g.bitmap_len].reshape(g.height + 6, g.width + 6).copy()
        return g

    def close(self):
        if self._h:
            _L().vg_renderer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GlyphBatchHost:
    """Owner of a host SoA batch produced by the C++ tessellation stage."""

    def __init__(self, handle):
        self._h = handle
        cb = _CBatch()
        ids = C.POINTER(C.c_uint32)()
        nj = C.c_uint32(0)
        _L().vg_glyph_batch_view(handle, C.byref(cb), C.byref(ids), C.byref(nj))
        n = cb.n_glyphs
        self.n_jobs = nj.value

        def arr(ptr, count, dt):
            if count == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt, count=count)

        seg_off = arr(cb.seg_off, n + 1, np.uint32)
        s = int(seg_off[-1]) if n else 0
        self.batch = Batch(seg_off, arr(cb.seg_sx, s, np.float64), arr(cb.seg_sy, s, np.float64),
                           arr(cb.seg_ex, s, np.float64), arr(cb.seg_ey, s, np.float64), arr(cb.x0, n, np.int32),
                           arr(cb.y0, n, np.int32), arr(cb.w, n, np.uint32), arr(cb.h, n, np.uint32),
                           arr(cb.out_off, n + 1, np.uint64))
        self.ids = arr(C.cast(ids, C.c_void_p).value, n, np.uint32)

    def free(self):
        if self._h:
            _L().vg_glyph_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FontManager:
    """src/font/manager.rs FontManager (render path only)."""

    def __init__(self, parallel: bool = True):
        self._h = _L().vg_manager_new(1 if parallel else 0)
        self._cb_keepalive = None

    def set_threads(self, threads: int = 0, blocks_per_batch: int = 0):
        _L().vg_manager_set_threads(self._h, threads, blocks_per_batch)

    def set_in_place_pbf(self, on: bool):
        """True (default): the raster stores bitmaps where the finished PBF has them; False: blocks are encoded afterwards"""
        _L().vg_manager_set_in_place_pbf(self._h, 1 if on else 0)

    def set_glyf_on_device(self, on: bool):
        """True (default): glyf fonts are decoded on the device; False: the host reader records the outline callbacks"""
        _L().vg_manager_set_glyf_on_device(self._h, 1 if on else 0)

    def set_lane_form(self, form: int):
        """several device lanes: -1 / 2 hybrid (whole (font, block) tasks, the heaviest blocks split between lanes), 1 whole tasks
        only, 0 glyph-level shards of every font + merge"""
        _L().vg_manager_set_lane_form(self._h, int(form))

    def set_device_front_end(self, on: bool):
        """flatten / close / scale / bbox on the GPU instead of host threads (HIP renderer only)"""
        _L().vg_manager_set_device_front_end(self._h, 1 if on else 0)

    def add_font_with_name(self, name: str, sources):
        paths = [str(Path(p)).encode() for p in sources]
        arr = (C.c_char_p * len(paths))(*paths)
        if _L().vg_manager_add_font_with_name(self._h, name.encode(), arr, len(paths)) != 0:
            raise RuntimeError(_err())
        return name_to_id(name)

    def add_font_data(self, name: str, data: bytes):
        if _L().vg_manager_add_font_data(self._h, name.encode(), data, len(data)) != 0:
            raise RuntimeError(_err())
        return name_to_id(name)

    def add_path(self, path):
        """manager.rs:39-53: the file's own name table decides the font id."""
        if _L().vg_manager_add_path(self._h, str(path).encode()) != 0:
            raise RuntimeError(_err())

    def add_paths(self, paths):
        for p in paths:
            self.add_path(p)

    def scan(self, path):
        """recurse.rs:104-133 (fonts.json aware; directory entries in sorted order)."""
        if _L().vg_manager_scan(self._h, str(path).encode()) != 0:
            raise RuntimeError(_err())

    def _text(self, fn, *args) -> bytes:
        need = fn(self._h, *args, None, 0)
        if need < 0:
            raise RuntimeError(_err())
        buf = C.create_string_buffer(need + 1)
        fn(self._h, *args, buf, need + 1)
        return buf.raw[:need]

    def font_ids(self):
        t = self._text(_L().vg_manager_font_ids).rstrip(b"\0").decode()
        return t.split("\n") if t else []

    def font_file_names(self, font_id: str):
        t = self._text(_L().vg_manager_font_file_names, font_id.encode()).rstrip(b"\0").decode()
        return t.split("\n") if t else []

    def generate_name(self, font_id: str, file_index: int = 0) -> str:
        return self._text(_L().vg_manager_generate_name, font_id.encode(), file_index).rstrip(b"\0").decode()

    def index_json(self) -> bytes:
        return self._text(_L().vg_manager_index_json)

    def families_json(self) -> bytes:
        return self._text(_L().vg_manager_families_json)

    def render_glyphs_to(self, writer: "NativeWriter", renderer: Renderer):
        """render_glyphs into a native sink (tar stream / directory)."""
        if _L().vg_manager_render_glyphs_to(self._h, renderer._h, writer._h) != 0:
            raise RuntimeError(_err())

    def write_index_json(self, writer: "NativeWriter"):
        if _L().vg_manager_write_index_json(self._h, writer._h) != 0:
            raise RuntimeError(_err())

    def write_families_json(self, writer: "NativeWriter"):
        if _L().vg_manager_write_families_json(self._h, writer._h) != 0:
            raise RuntimeError(_err())

    def shard_glyphs(self, font_id: str, world: int):
        """-> (owner u8[65536] with 0xFF = unmapped, estimated cost f64[65536]); SURVEY.md §8e."""
        owner = np.empty(65536, dtype=np.uint8)
        cost = np.empty(65536, dtype=np.float64)
        if _L().vg_manager_shard_glyphs(self._h, font_id.encode(), world, owner.ctypes.data, cost.ctypes.data) != 0:
            raise RuntimeError(_err())
        return owner, cost

    def plan_lanes(self, font_id: str, world: int):
        """-> (lane per code point u8[65536] with 0xFF = unmapped, number of this font's blocks split between lanes, the plan's own
        max / mean of the lanes' weights): what a run on `world` device lanes would do (no device needed)"""
        owner = np.empty(65536, dtype=np.uint8)
        n_split, ratio = C.c_uint32(0), C.c_double(0.0)
        if _L().vg_manager_plan_lanes(self._h, font_id.encode(), world, owner.ctypes.data, C.byref(n_split), C.byref(ratio)) != 0:
            raise RuntimeError(_err())
        return owner, int(n_split.value), float(ratio.value)

    def set_glyph_shard(self, rank: int, world: int):
        """Later render / build_batch calls see only rank's glyphs (world <= 1: off)."""
        if _L().vg_manager_set_glyph_shard(self._h, rank, world) != 0:
            raise RuntimeError(_err())

    def reduced_counters(self):
        """(blocks, glyphs, pixels) of the last render with a multi-device renderer, as reduced over its lanes"""
        out = (C.c_uint64 * 3)()
        _L().vg_manager_reduced_counters(self._h, out)
        return tuple(int(v) for v in out)

    def block_counts(self, font_id: str) -> np.ndarray:
        out = np.zeros(256, dtype=np.uint32)
        if _L().vg_manager_block_counts(self._h, font_id.encode(), out.ctypes.data_as(C.POINTER(C.c_uint32))) != 0:
            raise KeyError(_err())
        return out

    def render_glyphs(self, writer, renderer: Renderer, font_id: str = None, block_starts=None):
        """FontManager::render_glyphs(&mut writer, &renderer).  `writer` needs
        write_directory(path) and write_file(path, bytes), or is None.  With font_id + block_starts only
        that shard of the (font, block) task list is rendered."""
        errors = []

        def cb(_user, path, data, n, is_dir):
            try:
                if is_dir:
                    writer.write_directory(path.decode())
                else:
                    writer.write_file(path.decode(), C.string_at(data, n) if n else b"")
                return 0
            except Exception as e:  # propagate through the C boundary as an abort
                errors.append(e)
                return 1

        # writer=None: NULL sink (the bytes are produced and counted, see timings()["pbf_bytes"], but not
        # handed to Python) -- what a native caller's in-memory writer costs
        ccb = WRITE_CB(cb) if writer is not None else C.cast(None, WRITE_CB)
        if block_starts is None:
            rc = _L().vg_manager_render_glyphs(self._h, renderer._h, ccb, None)
        else:
            arr = (C.c_uint32 * max(len(block_starts), 1))(*[int(b) for b in block_starts])
            rc = _L().vg_manager_render_blocks(self._h, renderer._h, font_id.encode(), arr, len(block_starts), ccb, None)
        if errors:
            raise errors[0]
        if rc != 0:
            raise RuntimeError(_err())

    def timings(self) -> dict:
        t = Timings()
        _L().vg_manager_timings(self._h, C.byref(t))
        return t.as_dict()

    def render_block(self, renderer: Renderer, font_id: str, start: int) -> bytes:
        """GlyphBlock::render(font_name, renderer) for block `start`."""
        need = _L().vg_manager_render_block(self._h, renderer._h, font_id.encode(), start, None, 0)
        if need < 0:
            raise RuntimeError(_err())
        out = np.empty(need, dtype=np.uint8)
        got = _L().vg_manager_render_block(self._h, renderer._h, font_id.encode(), start, out.ctypes.data, need)
        if got != need:
            raise RuntimeError(_err())
        return out.tobytes()

    def record_outlines(self, font_id: str) -> dict:
        """host half of the device front-end: {cmd_off, cmds, scale, shift_x, ids, advances} (numpy copies)"""
        h = _L().vg_manager_record_outlines(self._h, font_id.encode())
        if not h:
            raise RuntimeError(_err())
        try:
            co = _COutlines()
            ids, adv = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
            _L().vg_outline_batch_view(h, C.byref(co), C.byref(ids), C.byref(adv))
            n = co.n_glyphs

            def arr(ptr, count, dt):
                if count == 0 or not ptr:
                    return np.zeros(0, dtype=dt)
                buf = (C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dt, count=count).copy()

            cmd_off = arr(co.cmd_off, n + 1, np.uint32)
            return {"cmd_off": cmd_off, "cmds": arr(co.cmds, int(cmd_off[-1]) if n else 0, OUTLINE_CMD_DTYPE),
                    "scale": arr(co.scale, n, np.float64), "shift_x": arr(co.shift_x, n, np.float64),
                    "ids": arr(C.cast(ids, C.c_void_p).value, n, np.uint32),
                    "advances": arr(C.cast(adv, C.c_void_p).value, n, np.uint32)}
        finally:
            _L().vg_outline_batch_free(h)

    def record_glyf_parts(self, font_id: str) -> dict:
        """the same for the device's glyf decoder: {cmd_off (command slots), parts, bytes, scale, shift_x, ids, advances}"""
        from .device import GLYF_PART_DTYPE, _COutlinesGlyf
        L = _L()
        L.vg_manager_record_glyf_parts.restype = C.c_void_p
        L.vg_manager_record_glyf_parts.argtypes = [C.c_void_p, C.c_char_p]
        L.vg_glyf_batch_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vg_glyf_batch_free.argtypes = [C.c_void_p]
        L.vg_glyf_batch_free.restype = None
        h = L.vg_manager_record_glyf_parts(self._h, font_id.encode())
        if not h:
            raise RuntimeError(_err())
        try:
            co = _COutlinesGlyf()
            ids, adv = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
            L.vg_glyf_batch_view(h, C.byref(co), C.byref(ids), C.byref(adv))
            n = co.n_glyphs

            def arr(ptr, count, dt):
                if count == 0 or not ptr:
                    return np.zeros(0, dtype=dt)
                buf = (C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dt, count=count).copy()

            return {"cmd_off": arr(co.cmd_off, n + 1, np.uint32), "parts": arr(co.parts, co.n_parts, GLYF_PART_DTYPE),
                    "bytes": arr(co.bytes, co.n_bytes, np.uint8),
                    "scale": arr(co.scale, n, np.float64), "shift_x": arr(co.shift_x, n, np.float64),
                    "ids": arr(C.cast(ids, C.c_void_p).value, n, np.uint32),
                    "advances": arr(C.cast(adv, C.c_void_p).value, n, np.uint32)}
        finally:
            L.vg_glyf_batch_free(h)

    def build_batch(self, font_id: str) -> GlyphBatchHost:
        h = _L().vg_manager_build_batch(self._h, font_id.encode())
        if not h:
            raise RuntimeError(_err())
        return GlyphBatchHost(h)

    def close(self):
        if self._h:
            _L().vg_manager_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeWriter:
    """src/writer: Writer::new_tar / Writer::new_file, implemented natively (csrc/host/writers.cpp)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError(_err())
        self._h = handle

    @classmethod
    def new_tar(cls, path, mtime: int = -1):
        return cls(_L().vg_writer_new_tar_path(str(path).encode(), mtime))

    @classmethod
    def new_tar_fd(cls, fd: int, mtime: int = -1):
        return cls(_L().vg_writer_new_tar_fd(fd, mtime))

    @classmethod
    def new_file(cls, folder):
        return cls(_L().vg_writer_new_dir(str(folder).encode()))

    def write_file(self, path: str, data: bytes):
        if _L().vg_writer_write_file(self._h, path.encode(), data, len(data)) != 0:
            raise RuntimeError(_err())

    def write_directory(self, path: str):
        if _L().vg_writer_write_directory(self._h, path.encode()) != 0:
            raise RuntimeError(_err())

    def finish(self):
        if _L().vg_writer_finish(self._h) != 0:
            raise RuntimeError(_err())

    def close(self):
        if self._h:
            _L().vg_writer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pbf_merge(parts, consecutive: bool = False) -> bytes:
    """Partial PBFs of one block (disjoint glyph subsets) -> the block's PBF.  consecutive=True: vg_pbf_concat (parts that
    hold consecutive runs of the block's code points, in order)."""
    parts = [bytes(p) for p in parts]
    fn = _L().vg_pbf_concat if consecutive else _L().vg_pbf_merge
    arr = (C.c_char_p * len(parts))(*parts)
    lens = (C.c_size_t * len(parts))(*[len(p) for p in parts])
    need = fn(arr, lens, len(parts), None, 0)
    if need < 0:
        raise RuntimeError(_err())
    out = np.empty(need, dtype=np.uint8)
    fn(arr, lens, len(parts), out.ctypes.data, need)
    return out.tobytes()


def parse_font_name(family: str, ps_name: str):
    """(family, style, weight, width) — src/font/parse_font_name.rs:214-291"""
    fam = C.create_string_buffer(1024)
    style = C.create_string_buffer(16)
    width = C.create_string_buffer(16)
    weight = C.c_uint16(0)
    _L().vg_parse_font_name(family.encode(), ps_name.encode(), fam, len(fam), style, C.byref(weight), width)
    return fam.value.decode(), style.value.decode(), int(weight.value), width.value.decode()


def encode_codeblocks(codepoints) -> str:
    """src/font/index_files.rs:65-103"""
    cps = np.ascontiguousarray(codepoints, dtype=np.uint32)
    need = _L().vg_encode_codeblocks(cps.ctypes.data, cps.size, None, 0)
    buf = C.create_string_buffer(need)
    _L().vg_encode_codeblocks(cps.ctypes.data, cps.size, buf, need)
    return buf.value.decode()


class DummyWriter:
    """src/writer/dummy.rs: records 'name (len)' strings; keeps the bytes too."""

    def __init__(self):
        self.inner, self.files = [], {}

    def write_directory(self, path):
        self.inner.append(path)

    def write_file(self, path, data):
        self.inner.append(f"{path} ({len(data)})")
        self.files[path] = data


def pbf_encode(name: str, range_: str, glyphs) -> bytes:
    """glyphs: [PbfGlyph] (bitmap attribute used when has_bitmap)."""
    n = len(glyphs)
    arr = (PbfGlyph * max(n, 1))()
    ptrs = (C.c_void_p * max(n, 1))()
    keep = []
    for i, g in enumerate(glyphs):
        arr[i] = g
        if g.has_bitmap and g.bitmap is not None:
            b = np.ascontiguousarray(g.bitmap, dtype=np.uint8)
            keep.append(b)
            arr[i].bitmap_len = b.size
            ptrs[i] = b.ctypes.data
    need = _L().vg_pbf_encode(name.encode(), range_.encode(), arr, ptrs, n, None, 0)
    out = np.empty(need, dtype=np.uint8)
    _L().vg_pbf_encode(name.encode(), range_.encode(), arr, ptrs, n, out.ctypes.data, need)
    return out.tobytes()
