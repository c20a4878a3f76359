"""ctypes binding of include/vgsdf.h (the device boundary)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .build import lib_path


class VgsdfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vgsdf error {code}: {msg}")
        self.code = code


class _CBatch(C.Structure):
    _fields_ = [("n_glyphs", C.c_uint32), ("seg_off", C.c_void_p), ("seg_sx", C.c_void_p),
                ("seg_sy", C.c_void_p), ("seg_ex", C.c_void_p), ("seg_ey", C.c_void_p),
                ("x0", C.c_void_p), ("y0", C.c_void_p), ("w", C.c_void_p), ("h", C.c_void_p),
                ("out_off", C.c_void_p)]


class _CStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_glyphs", "n_segments", "n_pixels", "n_pairs", "n_tiles", "alg_bytes")]


OUTLINE_CMD_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("x", "<f4"), ("y", "<f4"),
                              ("kind", "<u4")])
RECT_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("w", "<u4"), ("h", "<u4"), ("n_segments", "<u4"),
                       ("has_raster", "<u4")])


class _COutlines(C.Structure):
    _fields_ = [("n_glyphs", C.c_uint32), ("cmd_off", C.c_void_p), ("cmds", C.c_void_p), ("scale", C.c_void_p),
                ("shift_x", C.c_void_p)]


class _COutlinesPacked(C.Structure):
    _fields_ = [("n_glyphs", C.c_uint32), ("cmd_off", C.c_void_p), ("dat_off", C.c_void_p), ("kinds", C.c_void_p),
                ("coords", C.c_void_p), ("scale", C.c_void_p), ("shift_x", C.c_void_p),
                ("pbf_pre", C.c_void_p), ("pbf_fix", C.c_void_p)]  # in-place PBF assembly: NULL = bitmaps packed back to back


class _COutlinesGlyf(C.Structure):
    _fields_ = [("n_glyphs", C.c_uint32), ("n_parts", C.c_uint32), ("n_bytes", C.c_uint32), ("cmd_off", C.c_void_p),
                ("parts", C.c_void_p), ("bytes", C.c_void_p), ("scale", C.c_void_p), ("shift_x", C.c_void_p),
                ("pbf_pre", C.c_void_p), ("pbf_fix", C.c_void_p)]


# vgsdf_glyf_part: one simple glyph of a (possibly composite) glyph for the device's glyf decoder
GLYF_PART_DTYPE = np.dtype([("byte_off", "<u4"), ("byte_len", "<u4"), ("cmd_at", "<u4"), ("cmd_cap", "<u4"), ("n_contours", "<u4"),
                            ("plain", "<u4"), ("a", "<f4"), ("b", "<f4"), ("c", "<f4"), ("d", "<f4"), ("e", "<f4"), ("f", "<f4")])
VGSDF_E_GLYF = -4

VGSDF_SYMBOLS = [
    "vgsdf_device_count", "vgsdf_create", "vgsdf_destroy", "vgsdf_last_error", "vgsdf_render_batch",
    "vgsdf_batch_upload", "vgsdf_batch_launch", "vgsdf_batch_download", "vgsdf_batch_free", "vgsdf_sync",
    "vgsdf_batch_stats", "vgsdf_batch_time", "vgsdf_set_variant", "vgsdf_batch_device_output",
    "vgsdf_host_alloc", "vgsdf_host_free", "vgsdf_outlines_prepare", "vgsdf_outlines_render", "vgsdf_outlines_render_into", "vgsdf_outlines_submit", "vgsdf_outlines_submit_packed", "vgsdf_outlines_submit_glyf", "vgsdf_outlines_wait", "vgsdf_outlines_segments",
    "vgsdf_add_counters", "vgsdf_reset_counters", "vgsdf_reduce_counters", "vgsdf_reduce_counters_rccl", "vgsdf_reduce_path", "vgsdf_outlines_pbf_positions", "vgsdf_outlines_peek",
]

_lib = None


def load_library():
    """dlopen libvgsdf.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not p.exists():
            raise FileNotFoundError(f"{p} missing: run __graft_entry__.build() (hipcc) first; "
                                    "there is no CPU fallback")
        L = C.CDLL(str(p))
        vp = C.c_void_p
        L.vgsdf_device_count.restype = C.c_int
        L.vgsdf_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.vgsdf_destroy.argtypes = [vp]
        L.vgsdf_destroy.restype = None
        L.vgsdf_last_error.argtypes = [vp]
        L.vgsdf_last_error.restype = C.c_char_p
        L.vgsdf_render_batch.argtypes = [vp, C.POINTER(_CBatch), vp]
        L.vgsdf_batch_upload.argtypes = [vp, C.POINTER(_CBatch), C.POINTER(vp)]
        L.vgsdf_batch_launch.argtypes = [vp, vp]
        L.vgsdf_batch_download.argtypes = [vp, vp, vp]
        L.vgsdf_batch_free.argtypes = [vp, vp]
        L.vgsdf_sync.argtypes = [vp]
        L.vgsdf_batch_stats.argtypes = [vp, C.POINTER(_CStats)]
        L.vgsdf_batch_time.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_float)]
        L.vgsdf_set_variant.argtypes = [vp, C.c_int]
        L.vgsdf_batch_device_output.argtypes = [vp]
        L.vgsdf_batch_device_output.restype = vp
        L.vgsdf_host_alloc.argtypes = [C.c_size_t]
        L.vgsdf_host_alloc.restype = vp
        L.vgsdf_host_free.argtypes = [vp]
        L.vgsdf_host_free.restype = None
        L.vgsdf_outlines_prepare.argtypes = [vp, C.POINTER(_COutlines), vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.vgsdf_outlines_render.argtypes = [vp, vp]
        L.vgsdf_outlines_render_into.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
        L.vgsdf_outlines_submit.argtypes = [vp, vp, vp, C.c_size_t]
        L.vgsdf_outlines_wait.argtypes = [vp, vp, vp, vp, vp]
        L.vgsdf_outlines_submit_packed.argtypes = [vp, vp, vp, C.c_size_t]
        L.vgsdf_outlines_submit_glyf.argtypes = [vp, vp, vp, C.c_size_t]
        L.vgsdf_outlines_segments.argtypes = [vp, vp, vp, vp, vp, vp]
        L.vgsdf_outlines_pbf_positions.argtypes = [vp, vp]
        L.vgsdf_outlines_peek.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.vgsdf_add_counters.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
        L.vgsdf_add_counters.restype = None
        L.vgsdf_reset_counters.argtypes = [vp]
        L.vgsdf_reset_counters.restype = None
        L.vgsdf_reduce_counters.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_uint64)]
        L.vgsdf_reduce_counters_rccl.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_uint64)]
        L.vgsdf_reduce_path.argtypes = [vp]
        L.vgsdf_reduce_path.restype = C.c_char_p
        _lib = L
    return _lib


def device_count() -> int:
    return load_library().vgsdf_device_count()


def reduce_counters(contexts, strict=False):
    """vgsdf_reduce_counters: (blocks, glyphs, pixels) summed over the contexts' run counters — an RCCL all-reduce when
    two or more contexts sit on distinct devices, with the host's own sum as the flagged fallback (reduce_path tells);
    strict=True: vgsdf_reduce_counters_rccl — through RCCL or an error (a single context: a communicator of one rank)."""
    L = load_library()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    out = (C.c_uint64 * 3)()
    rc = (L.vgsdf_reduce_counters_rccl if strict else L.vgsdf_reduce_counters)(arr, len(contexts), out)
    if rc != 0:
        raise VgsdfError(rc, (L.vgsdf_last_error(contexts[0]._h) or b"").decode())
    return tuple(int(v) for v in out)


def reduce_path(context) -> str:
    """vgsdf_reduce_path: "rccl", "host: one context", "host: contexts share a device" or "host: RCCL fallback: <reason>" """
    return (load_library().vgsdf_reduce_path(context._h) or b"").decode()


@dataclass
class Batch:
    """Host SoA batch (vgsdf_batch).  Arrays are kept alive by this object."""
    seg_off: np.ndarray  # u32 [n+1]
    seg_sx: np.ndarray   # f64 [S]
    seg_sy: np.ndarray
    seg_ex: np.ndarray
    seg_ey: np.ndarray
    x0: np.ndarray       # i32 [n]
    y0: np.ndarray
    w: np.ndarray        # u32 [n]
    h: np.ndarray
    out_off: np.ndarray  # u64 [n+1]

    @property
    def n_glyphs(self) -> int:
        return len(self.w)

    @property
    def out_bytes(self) -> int:
        return int(self.out_off[-1]) if len(self.out_off) else 0

    def c_struct(self) -> _CBatch:
        p = lambda a: a.ctypes.data  # noqa: E731
        return _CBatch(self.n_glyphs, p(self.seg_off), p(self.seg_sx), p(self.seg_sy), p(self.seg_ex),
                       p(self.seg_ey), p(self.x0), p(self.y0), p(self.w), p(self.h), p(self.out_off))

    def bitmap(self, out: np.ndarray, g: int) -> np.ndarray:
        a, b = int(self.out_off[g]), int(self.out_off[g + 1])
        return out[a:b].reshape(int(self.h[g]), int(self.w[g]))


def make_batch(glyphs) -> Batch:
    """glyphs: iterable of (segs[n,4] f64 (sx,sy,ex,ey), x0, y0, w, h)."""
    glyphs = list(glyphs)
    n = len(glyphs)
    seg_off = np.zeros(n + 1, dtype=np.uint32)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    x0 = np.zeros(n, dtype=np.int32)
    y0 = np.zeros(n, dtype=np.int32)
    w = np.zeros(n, dtype=np.uint32)
    h = np.zeros(n, dtype=np.uint32)
    parts = []
    for i, (segs, gx0, gy0, gw, gh) in enumerate(glyphs):
        segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 4)
        parts.append(segs)
        seg_off[i + 1] = seg_off[i] + len(segs)
        out_off[i + 1] = out_off[i] + np.uint64(int(gw) * int(gh))
        x0[i], y0[i], w[i], h[i] = gx0, gy0, gw, gh
    allseg = np.concatenate(parts, axis=0) if parts else np.zeros((0, 4))
    col = lambda k: np.ascontiguousarray(allseg[:, k])  # noqa: E731
    return Batch(seg_off, col(0), col(1), col(2), col(3), x0, y0, w, h, out_off)


class DeviceBatch:
    """A batch resident in HBM (vgsdf_dbatch)."""

    def __init__(self, ctx: "SdfContext", handle, out_bytes: int):
        self.ctx, self._h, self.out_bytes = ctx, handle, out_bytes

    def launch(self):
        self.ctx._check(load_library().vgsdf_batch_launch(self.ctx._h, self._h))

    def download(self) -> np.ndarray:
        out = np.empty(self.out_bytes, dtype=np.uint8)
        self.ctx._check(load_library().vgsdf_batch_download(self.ctx._h, self._h, out.ctypes.data))
        return out

    def time(self, iters: int) -> float:
        """total milliseconds for `iters` launches (HIP events on the context stream)."""
        ms = C.c_float(0)
        self.ctx._check(load_library().vgsdf_batch_time(self.ctx._h, self._h, iters, C.byref(ms)))
        return float(ms.value)

    def stats(self) -> dict:
        s = _CStats()
        load_library().vgsdf_batch_stats(self._h, C.byref(s))
        return {k: int(getattr(s, k)) for k, _ in _CStats._fields_}

    def device_output_ptr(self) -> int:
        return int(load_library().vgsdf_batch_device_output(self._h) or 0)

    def free(self):
        if self._h:
            load_library().vgsdf_batch_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass


class SdfContext:
    """vgsdf_ctx: one per (thread, GPU)."""

    def __init__(self, device: int = 0):
        L = load_library()
        h = C.c_void_p()
        rc = L.vgsdf_create(device, C.byref(h))
        if rc != 0:
            raise VgsdfError(rc, (L.vgsdf_last_error(None) or b"").decode())
        self._h = h

    def _check(self, rc: int):
        if rc != 0:
            raise VgsdfError(rc, (load_library().vgsdf_last_error(self._h) or b"").decode())

    def set_variant(self, v: int):
        self._check(load_library().vgsdf_set_variant(self._h, v))

    def add_counters(self, blocks: int, glyphs: int, pixels: int):
        load_library().vgsdf_add_counters(self._h, blocks, glyphs, pixels)

    def reset_counters(self):
        load_library().vgsdf_reset_counters(self._h)

    def render_batch(self, batch: Batch) -> np.ndarray:
        out = np.empty(batch.out_bytes, dtype=np.uint8)
        cb = batch.c_struct()
        self._check(load_library().vgsdf_render_batch(self._h, C.byref(cb), out.ctypes.data))
        return out

    def upload(self, batch: Batch) -> DeviceBatch:
        cb = batch.c_struct()
        h = C.c_void_p()
        self._check(load_library().vgsdf_batch_upload(self._h, C.byref(cb), C.byref(h)))
        self.sync()
        return DeviceBatch(self, h, batch.out_bytes)

    def outlines_prepare(self, cmd_off, cmds, scale, shift_x):
        """device front-end, step 1: outline commands -> (rects, out_bytes, n_segments)"""
        cmd_off = np.ascontiguousarray(cmd_off, dtype=np.uint32)
        cmds = np.ascontiguousarray(cmds, dtype=OUTLINE_CMD_DTYPE)
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        shift_x = np.ascontiguousarray(shift_x, dtype=np.float64)
        n = len(scale)
        rects = np.zeros(n, dtype=RECT_DTYPE)
        ob, ns = C.c_uint64(0), C.c_uint64(0)
        co = _COutlines(n, cmd_off.ctypes.data, cmds.ctypes.data, scale.ctypes.data, shift_x.ctypes.data)
        self._check(load_library().vgsdf_outlines_prepare(self._h, C.byref(co), rects.ctypes.data, C.byref(ob), C.byref(ns)))
        self._fe = (n, int(ob.value), int(ns.value))
        return rects, int(ob.value), int(ns.value)

    def outlines_render_into(self, cmd_off, cmds, scale, shift_x, capacity: int, pinned: bool = True):
        """device front-end as ONE submission -> (rects, bitmaps | None, out_bytes, n_segments); bitmaps is None when
        `capacity` bytes were too few (the batch stays prepared: outlines_render() finishes it)"""
        L = load_library()
        cmd_off = np.ascontiguousarray(cmd_off, dtype=np.uint32)
        cmds = np.ascontiguousarray(cmds, dtype=OUTLINE_CMD_DTYPE)
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        shift_x = np.ascontiguousarray(shift_x, dtype=np.float64)
        n = len(scale)
        rects = np.zeros(n, dtype=RECT_DTYPE)
        ob, ns, done = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        co = _COutlines(n, cmd_off.ctypes.data, cmds.ctypes.data, scale.ctypes.data, shift_x.ctypes.data)
        host = L.vgsdf_host_alloc(max(capacity, 1)) if pinned else None
        if pinned and not host:
            raise MemoryError("vgsdf_host_alloc")
        try:
            buf = (C.c_uint8 * max(capacity, 1)).from_address(host) if pinned else (C.c_uint8 * max(capacity, 1))()
            self._check(L.vgsdf_outlines_render_into(self._h, C.byref(co), rects.ctypes.data, C.addressof(buf), capacity,
                                                     C.byref(ob), C.byref(ns), C.byref(done)))
            self._fe = (n, int(ob.value), int(ns.value))
            out = np.frombuffer(buf, dtype=np.uint8, count=int(ob.value)).copy() if done.value else None
        finally:
            if pinned:
                L.vgsdf_host_free(host)
        return rects, out, int(ob.value), int(ns.value)

    def outlines_submit(self, cmd_off, cmds, scale, shift_x, capacity: int):
        """first half of the one-submission form: everything is enqueued, nothing waited for (one per context)"""
        L = load_library()
        keep = {
            "cmd_off": np.ascontiguousarray(cmd_off, dtype=np.uint32), "cmds": np.ascontiguousarray(cmds, dtype=OUTLINE_CMD_DTYPE),
            "scale": np.ascontiguousarray(scale, dtype=np.float64), "shift": np.ascontiguousarray(shift_x, dtype=np.float64),
        }
        n = len(keep["scale"])
        host = L.vgsdf_host_alloc(max(capacity, 1))
        if not host:
            raise MemoryError("vgsdf_host_alloc")
        co = _COutlines(n, keep["cmd_off"].ctypes.data, keep["cmds"].ctypes.data, keep["scale"].ctypes.data, keep["shift"].ctypes.data)
        rc = L.vgsdf_outlines_submit(self._h, C.byref(co), host, capacity)
        if rc != 0:
            L.vgsdf_host_free(host)
            self._check(rc)
        self._inflight = (keep, host, capacity, n)

    @staticmethod
    def pack_outlines(cmd_off, cmds):
        """28-byte command records -> (dat_off, kinds, coords) of vgsdf_outlines_packed"""
        cmds = np.ascontiguousarray(cmds, dtype=OUTLINE_CMD_DTYPE)
        kinds = cmds["kind"].astype(np.uint8)
        nf = np.select([cmds["kind"] <= 1, cmds["kind"] == 2, cmds["kind"] == 3], [2, 4, 6], 0).astype(np.int64)
        at = np.concatenate([[0], np.cumsum(nf)])
        coords = np.zeros(int(at[-1]), dtype=np.float32)
        for fields, k in ((("x", "y"), (0, 1)), (("x1", "y1", "x", "y"), (2,)), (("x1", "y1", "x2", "y2", "x", "y"), (3,))):
            sel = np.isin(cmds["kind"], k)
            for j, f in enumerate(fields):
                coords[at[:-1][sel] + j] = cmds[f][sel]
        dat_off = at[np.asarray(cmd_off, dtype=np.int64)].astype(np.uint32)
        return dat_off, kinds, coords

    def outlines_submit_packed(self, cmd_off, dat_off, kinds, coords, scale, shift_x, capacity: int, pbf_pre=None, pbf_fix=None):
        """outlines_submit for the compact upload form (vgsdf_outlines_packed); pbf_pre / pbf_fix: in-place PBF assembly"""
        L = load_library()
        keep = {
            "cmd_off": np.ascontiguousarray(cmd_off, dtype=np.uint32), "dat_off": np.ascontiguousarray(dat_off, dtype=np.uint32),
            "kinds": np.ascontiguousarray(kinds, dtype=np.uint8), "coords": np.ascontiguousarray(coords, dtype=np.float32),
            "scale": np.ascontiguousarray(scale, dtype=np.float64), "shift": np.ascontiguousarray(shift_x, dtype=np.float64),
        }
        n = len(keep["scale"])
        host = L.vgsdf_host_alloc(max(capacity, 1))
        if not host:
            raise MemoryError("vgsdf_host_alloc")
        co = _COutlinesPacked(n, keep["cmd_off"].ctypes.data, keep["dat_off"].ctypes.data, keep["kinds"].ctypes.data,
                              keep["coords"].ctypes.data, keep["scale"].ctypes.data, keep["shift"].ctypes.data)
        if pbf_pre is not None:
            keep["pbf_pre"] = np.ascontiguousarray(pbf_pre, dtype=np.uint32)
            co.pbf_pre = keep["pbf_pre"].ctypes.data
        if pbf_fix is not None:
            keep["pbf_fix"] = np.ascontiguousarray(pbf_fix, dtype=np.uint8)
            co.pbf_fix = keep["pbf_fix"].ctypes.data
        rc = L.vgsdf_outlines_submit_packed(self._h, C.byref(co), host, capacity)
        if rc != 0:
            L.vgsdf_host_free(host)
            self._check(rc)
        self._inflight = (keep, host, capacity, n)

    def outlines_submit_glyf(self, cmd_off, parts, glyf_bytes, scale, shift_x, capacity: int, pbf_pre=None, pbf_fix=None):
        """outlines_submit for glyphs that arrive as their `glyf` arrays (vgsdf_outlines_glyf): the device decodes them"""
        L = load_library()
        keep = {
            "cmd_off": np.ascontiguousarray(cmd_off, dtype=np.uint32), "parts": np.ascontiguousarray(parts, dtype=GLYF_PART_DTYPE),
            "bytes": np.ascontiguousarray(glyf_bytes, dtype=np.uint8),
            "scale": np.ascontiguousarray(scale, dtype=np.float64), "shift": np.ascontiguousarray(shift_x, dtype=np.float64),
        }
        n = len(keep["scale"])
        host = L.vgsdf_host_alloc(max(capacity, 1))
        if not host:
            raise MemoryError("vgsdf_host_alloc")
        co = _COutlinesGlyf(n, len(keep["parts"]), len(keep["bytes"]), keep["cmd_off"].ctypes.data, keep["parts"].ctypes.data,
                            keep["bytes"].ctypes.data, keep["scale"].ctypes.data, keep["shift"].ctypes.data)
        if pbf_pre is not None:
            keep["pbf_pre"] = np.ascontiguousarray(pbf_pre, dtype=np.uint32)
            co.pbf_pre = keep["pbf_pre"].ctypes.data
        if pbf_fix is not None:
            keep["pbf_fix"] = np.ascontiguousarray(pbf_fix, dtype=np.uint8)
            co.pbf_fix = keep["pbf_fix"].ctypes.data
        rc = L.vgsdf_outlines_submit_glyf(self._h, C.byref(co), host, capacity)
        if rc != 0:
            L.vgsdf_host_free(host)
            self._check(rc)
        self._inflight = (keep, host, capacity, n)

    def outlines_peek(self):
        """between submit and wait: the front-end's results while the raster is still running -> (rects, out_bytes, in_place)"""
        ob, ip = C.c_uint64(0), C.c_int(0)
        if getattr(self, "_inflight", None) is None:  # (the library says so)
            self._check(load_library().vgsdf_outlines_peek(self._h, None, C.byref(ob), C.byref(ip)))
        keep, host, capacity, n = self._inflight
        rects = np.zeros(n, dtype=RECT_DTYPE)
        self._check(load_library().vgsdf_outlines_peek(self._h, rects.ctypes.data, C.byref(ob), C.byref(ip)))
        return rects, int(ob.value), bool(ip.value)

    def outlines_wait(self):
        """second half -> (rects, bitmaps | None, out_bytes, n_segments)"""
        L = load_library()
        keep, host, capacity, n = self._inflight
        self._inflight = None
        try:
            rects = np.zeros(n, dtype=RECT_DTYPE)
            ob, ns, done = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
            self._check(L.vgsdf_outlines_wait(self._h, rects.ctypes.data, C.byref(ob), C.byref(ns), C.byref(done)))
            self._fe = (n, int(ob.value), int(ns.value))
            buf = (C.c_uint8 * max(capacity, 1)).from_address(host)
            out = np.frombuffer(buf, dtype=np.uint8, count=int(ob.value)).copy() if done.value else None
        finally:
            L.vgsdf_host_free(host)
        return rects, out, int(ob.value), int(ns.value)

    def outlines_pbf_positions(self) -> np.ndarray:
        """after outlines_wait on a batch submitted with pbf_pre / pbf_fix: position of every glyph's bitmap in the arena"""
        at = np.zeros(self._inflight[3] if getattr(self, "_inflight", None) else self._fe[0], dtype=np.uint64)
        self._check(load_library().vgsdf_outlines_pbf_positions(self._h, at.ctypes.data))
        return at

    def outlines_render(self) -> np.ndarray:
        """device front-end, step 2: bitmaps of the glyphs with a raster, packed in glyph order"""
        out = np.empty(self._fe[1], dtype=np.uint8)
        self._check(load_library().vgsdf_outlines_render(self._h, out.ctypes.data))
        return out

    def outlines_segments(self):
        """the segments the device front-end produced -> (seg_off[n+1], segs[S,4])"""
        n, _, ns = self._fe
        seg_off = np.zeros(n + 1, dtype=np.uint32)
        cols = [np.zeros(ns, dtype=np.float64) for _ in range(4)]
        self._check(load_library().vgsdf_outlines_segments(self._h, seg_off.ctypes.data, *[c.ctypes.data for c in cols]))
        return seg_off, np.stack(cols, axis=1) if ns else np.zeros((0, 4))

    def sync(self):
        self._check(load_library().vgsdf_sync(self._h))

    def close(self):
        if self._h:
            load_library().vgsdf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
