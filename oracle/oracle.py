"""ctypes front-end of the CPU ORACLE (oracle/vg_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "build" / "libvgoracle.so"

PRECISE, BRUTE, DUMMY = 0, 1, 2
MOVE, LINE, QUAD, CURVE, CLOSE = 0, 1, 2, 3, 4


class Cmd(C.Structure):
    _fields_ = [("kind", C.c_int32), ("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float),
                ("y2", C.c_float), ("x", C.c_float), ("y", C.c_float)]


class GlyphInfo(C.Structure):
    _fields_ = [("id", C.c_uint32), ("has_bitmap", C.c_int32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("left", C.c_int32), ("top", C.c_int32),
                ("advance", C.c_uint32), ("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_uint32),
                ("h", C.c_uint32), ("n_segments", C.c_int32)]

    def metrics(self):
        return (self.width, self.height, self.left, self.top, self.advance)


def build(force: bool = False) -> Path:
    src = [HERE / "vg_oracle.c", HERE / "vg_oracle.h"]
    if force or not LIB_PATH.exists() or any(
            s.exists() and s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in src):
        subprocess.run(["make", "-C", str(HERE), "-s"], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        # VG_ORACLE_LIB: another build of the oracle (tools/run_asan.sh loads the sanitizer build)
        other = os.environ.get("VG_ORACLE_LIB")
        if not other:
            build()
        L = C.CDLL(other or str(LIB_PATH))
        vp, u8p, u32p, i32p, f64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), \
            C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.vgo_font_open.restype = vp
        L.vgo_font_open.argtypes = [C.c_char_p, C.c_size_t]
        L.vgo_font_close.argtypes = [vp]
        L.vgo_font_units_per_em.argtypes = [vp]
        L.vgo_font_num_glyphs.argtypes = [vp]
        L.vgo_font_codepoints.argtypes = [vp, u32p, C.c_int]
        L.vgo_font_glyph_index.argtypes = [vp, C.c_uint32]
        L.vgo_font_hor_advance.argtypes = [vp, C.c_int]
        L.vgo_font_outline.argtypes = [vp, C.c_int, C.POINTER(Cmd), C.c_int]
        L.vgo_build_rings.argtypes = [C.POINTER(Cmd), C.c_int, f64p, C.c_int, i32p, C.c_int, i32p]
        L.vgo_prepare_glyph.argtypes = [vp, C.c_uint32, C.POINTER(GlyphInfo), f64p, C.c_int]
        L.vgo_sdf_render.restype = None
        L.vgo_sdf_render.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.vgo_render_glyph.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(GlyphInfo), u8p, C.c_size_t]
        L.vgo_pbf_encode.restype = C.c_size_t
        L.vgo_pbf_encode.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(GlyphInfo),
                                     C.POINTER(u8p), C.c_int, u8p, C.c_size_t]
        L.vgo_render_block.restype = C.c_size_t
        L.vgo_render_block.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.c_uint32, C.c_int, u8p,
                                       C.c_size_t, i32p, C.POINTER(C.c_uint64)]
        L.vgo_render_all.restype = C.c_double
        L.vgo_render_all.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64)]
        L.vgo_sdf_render_batch.restype = C.c_double
        L.vgo_sdf_render_batch.argtypes = [C.c_uint32] + [vp] * 10 + [C.c_int, C.c_int, vp]
        _lib = L
    return _lib


def _u8p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Font:
    """ttf_parser::Face as far as the render path uses it."""

    def __init__(self, path_or_bytes):
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else Path(path_or_bytes).read_bytes()
        self._h = lib().vgo_font_open(bytes(data), len(data))
        if not self._h:
            raise ValueError("font parse failed")

    def close(self):
        if self._h:
            lib().vgo_font_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def units_per_em(self):
        return lib().vgo_font_units_per_em(self._h)

    @property
    def num_glyphs(self):
        return lib().vgo_font_num_glyphs(self._h)

    def codepoints(self) -> np.ndarray:
        n = lib().vgo_font_codepoints(self._h, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        lib().vgo_font_codepoints(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out

    def glyph_index(self, cp: int):
        g = lib().vgo_font_glyph_index(self._h, cp)
        return None if g < 0 else g

    def hor_advance(self, gid: int):
        a = lib().vgo_font_hor_advance(self._h, gid)
        return None if a < 0 else a

    def outline(self, gid: int):
        n = lib().vgo_font_outline(self._h, gid, None, 0)
        arr = (Cmd * max(n, 1))()
        lib().vgo_font_outline(self._h, gid, arr, n)
        return [(c.kind, c.x1, c.y1, c.x2, c.y2, c.x, c.y) for c in arr[:n]]

    def prepare_glyph(self, cp: int):
        """-> (info, segs[n,4]) or None (glyph skipped)."""
        info = GlyphInfo()
        some = lib().vgo_prepare_glyph(self._h, cp, C.byref(info), None, 0)
        if not some:
            return None
        segs = np.zeros((max(info.n_segments, 0), 4), dtype=np.float64)
        if info.has_bitmap:
            lib().vgo_prepare_glyph(self._h, cp, C.byref(info), _f64p(segs), info.n_segments)
        return info, segs

    def render_glyph(self, cp: int, mode: int = PRECISE):
        """Renderer::render_glyph -> (info, bitmap[h,w] | None) or None."""
        info = GlyphInfo()
        some = lib().vgo_prepare_glyph(self._h, cp, C.byref(info), None, 0)
        if not some:
            return None
        if not info.has_bitmap:
            return info, None
        bm = np.zeros((info.h, info.w), dtype=np.uint8)
        r = lib().vgo_render_glyph(self._h, cp, mode, C.byref(info), _u8p(bm), bm.size)
        assert r == 1
        return info, bm


def build_rings(cmds):
    """RingBuilder over [(kind,x1,y1,x2,y2,x,y)] -> list of (n,2) point arrays (font units)."""
    arr = (Cmd * max(len(cmds), 1))()
    for i, c in enumerate(cmds):
        arr[i] = Cmd(*c)
    cap = 1 << 16
    pts = np.zeros((cap, 2), dtype=np.float64)
    offs = np.zeros(4096, dtype=np.int32)
    npts = C.c_int32(0)
    n = lib().vgo_build_rings(arr, len(cmds), _f64p(pts), cap, offs.ctypes.data_as(C.POINTER(C.c_int32)),
                              len(offs), C.byref(npts))
    assert n >= 0
    return [pts[offs[i]:offs[i + 1]].copy() for i in range(n)]


def sdf_render(segs: np.ndarray, x0: int, y0: int, w: int, h: int, mode: int = PRECISE) -> np.ndarray:
    """renderer_precise on explicit segments (n,4: sx,sy,ex,ey) -> u8 [h,w] (top row first)."""
    segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((h, w), dtype=np.uint8)
    lib().vgo_sdf_render(_f64p(segs), len(segs), x0, y0, w, h, mode, _u8p(out))
    return out


def pbf_encode(name: str, start: int, glyphs) -> bytes:
    """glyphs: [(GlyphInfo, bitmap|None)] in the order to be written."""
    n = len(glyphs)
    infos = (GlyphInfo * max(n, 1))()
    ptrs = (C.POINTER(C.c_uint8) * max(n, 1))()
    keep = []
    for i, (info, bm) in enumerate(glyphs):
        infos[i] = info
        if bm is not None:
            b = np.ascontiguousarray(bm, dtype=np.uint8)
            keep.append(b)
            ptrs[i] = _u8p(b)
    need = lib().vgo_pbf_encode(name.encode(), start, infos, ptrs, n, None, 0)
    out = np.zeros(need, dtype=np.uint8)
    lib().vgo_pbf_encode(name.encode(), start, infos, ptrs, n, _u8p(out), need)
    return out.tobytes()


def _font_array(fonts):
    arr = (C.c_void_p * len(fonts))()
    for i, f in enumerate(fonts):
        arr[i] = f._h
    return arr


def render_block(fonts, name: str, start: int, mode: int = PRECISE):
    """GlyphBlock::render (canonical id order) -> (pbf bytes, n_glyphs, n_pixels)."""
    arr = _font_array(fonts)
    ng = C.c_int32(0)
    px = C.c_uint64(0)
    need = lib().vgo_render_block(arr, len(fonts), name.encode(), start, mode, None, 0, C.byref(ng), C.byref(px))
    out = np.zeros(need, dtype=np.uint8)
    lib().vgo_render_block(arr, len(fonts), name.encode(), start, mode, _u8p(out), need, C.byref(ng), C.byref(px))
    return out.tobytes(), ng.value, px.value


def render_all(fonts, name: str, mode: int = PRECISE, threads: int = 1, only_block: int = -1):
    """FontManager::render_glyphs analogue -> (seconds, dict counters)."""
    arr = _font_array(fonts)
    ctr = (C.c_uint64 * 6)()
    secs = lib().vgo_render_all(arr, len(fonts), name.encode(), mode, threads, only_block, ctr)
    keys = ["blocks", "glyphs", "pixels", "segments", "pbf_bytes", "hash"]
    return secs, dict(zip(keys, [int(v) for v in ctr]))


def default_threads() -> int:
    """Worker threads for the CPU legs: the CPU affinity of this process (what the box gives this job), no
    cap; VG_CPU_THREADS overrides."""
    if os.environ.get("VG_CPU_THREADS"):
        return max(1, int(os.environ["VG_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, n)


def sdf_render_batch(batch, mode: int = PRECISE, threads: int = 0):
    """Raster-only over a whole SoA batch (object with the vgsdf_batch arrays as numpy
    attributes: seg_off, seg_sx, seg_sy, seg_ex, seg_ey, x0, y0, w, h, out_off).
    -> (u8 output buffer, wall seconds)"""
    n = len(batch.w)
    out = np.zeros(int(batch.out_off[-1]) if n else 0, dtype=np.uint8)
    p = lambda a: a.ctypes.data  # noqa: E731
    secs = lib().vgo_sdf_render_batch(n, p(batch.seg_off), p(batch.seg_sx), p(batch.seg_sy), p(batch.seg_ex),
                                      p(batch.seg_ey), p(batch.x0), p(batch.y0), p(batch.w), p(batch.h),
                                      p(batch.out_off), mode, threads or default_threads(), p(out))
    return out, secs
