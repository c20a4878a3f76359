"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import re

from conftest import ROOT


def declared_symbols(header):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg(?:sdf)?_[a-z_0-9]+)\s*\(", text)))


def test_headers_declare_and_library_exports(vg):
    lib = ctypes.CDLL(str(vg.lib_path()))
    sdf = declared_symbols("vgsdf.h")
    font = [s for s in declared_symbols("vgfont.h") if s != "vg_write_cb"]
    assert "vgsdf_render_batch" in sdf and "vg_manager_render_glyphs" in font
    for sym in sdf + font:
        assert hasattr(lib, sym), f"{sym} declared in include/ but not exported"
    # and the Python binding lists exactly the declared entry points
    from versatiles_glyphs_rs_amd.device import VGSDF_SYMBOLS
    from versatiles_glyphs_rs_amd.host import VGFONT_SYMBOLS
    assert sorted(VGSDF_SYMBOLS) == sdf
    assert sorted(VGFONT_SYMBOLS) == font


def test_device_count_without_gpu_is_zero_or_more(vg):
    assert vg.device_count() >= 0


def test_product_does_not_reference_oracle():
    # the product path must never route through the oracle
    pkg = ROOT / "versatiles-glyphs-rs_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.hip")) + \
            list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.h")) + [pkg / "Makefile"]:
        txt = p.read_text()
        assert "vg_oracle" not in txt and "libvgoracle" not in txt, p
        assert not re.search(r"^\s*(import oracle|from oracle)", txt, flags=re.M), p


def test_product_library_holds_only_product_kernels(vg):
    """The shipped libvgsdf.so carries the default raster, the brute-force raster and the batch
    preparation kernels; the earlier generations and the timing-only ablation instances (wrong
    pixels) exist only in `make dev` builds."""
    blob = vg.lib_path().read_bytes()
    assert b"sdf_tiles_span" in blob and b"sdf_tiles_brute" in blob
    for retired in (b"sdf_tiles_filtered", b"sdf_tiles_pk", b"sdf_tiles_hier"):
        assert retired not in blob, retired
    import ctypes
    lib = ctypes.CDLL(str(vg.lib_path()))
    lib.vgsdf_kernel_known.argtypes = [ctypes.c_int]
    known = [k for k in range(0, 100) if lib.vgsdf_kernel_known(k)]
    assert known == [1, 50]
