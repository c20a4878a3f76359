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


_C_PROGRAM = r"""
/* the `recurse` flow of INTEGRATION.md section 5 from plain C (dummy renderer: runs without a GPU) */
#include <stdio.h>
#include "vgsdf.h"
#include "vgfont.h"
int main(int argc, char **argv)
{
	if (argc < 3)
		return 2;
	vg_manager *m = vg_manager_new(1);
	if (!m || vg_manager_scan(m, argv[1]) < 0) {
		fprintf(stderr, "scan: %s\n", vg_last_error());
		return 1;
	}
	vg_renderer *r = vg_renderer_new(VG_MODE_DUMMY, 0);
	vg_writer *w = vg_writer_new_dir(argv[2]);
	if (!r || !w || vg_manager_render_glyphs_to(m, r, w) < 0 || vg_manager_write_index_json(m, w) < 0 ||
	    vg_manager_write_families_json(m, w) < 0 || vg_writer_finish(w) < 0) {
		fprintf(stderr, "render: %s\n", vg_last_error());
		return 1;
	}
	vg_writer_free(w);
	vg_renderer_free(r);
	/* what the device's glyf decoder would be handed for this font (vgsdf_outlines_submit_glyf): built on the host, no GPU */
	{
		vg_glyf_batch *gb = vg_manager_record_glyf_parts(m, "fira_sans_regular");
		vgsdf_outlines_glyf v;
		const uint32_t *ids = NULL;
		uint32_t i, slots = 0;
		if (!gb || vg_glyf_batch_view(gb, &v, &ids, NULL) != 0) {
			fprintf(stderr, "parts: %s\n", vg_last_error());
			return 1;
		}
		for (i = 0; i < v.n_parts; i++) {
			if (v.parts[i].cmd_at != slots || v.parts[i].byte_off % 4 || v.parts[i].byte_off + v.parts[i].byte_len > v.n_bytes)
				return 3;
			slots += v.parts[i].cmd_cap;
		}
		if (v.n_glyphs < 1000 || v.n_parts < v.n_glyphs || slots != v.cmd_off[v.n_glyphs] || ids[0] != 13)
			return 4;
		printf("parts %u of %u glyphs, %u bytes\n", (unsigned)v.n_parts, (unsigned)v.n_glyphs, (unsigned)v.n_bytes);
		vg_glyf_batch_free(gb);
	}
	vg_manager_free(m);
	printf("devices %d\n", vgsdf_device_count());
	return 0;
}
"""


def test_headers_are_plain_c_and_the_library_links_from_c(vg, tmp_path):
    """include/*.h compile as C99 (-pedantic) and a C program linked against libvgsdf.so runs the scan -> render ->
    index flow (the boundary a cgo / JNI / Rust FFI binding would use)"""
    import shutil
    import subprocess
    from conftest import FIRA
    src = tmp_path / "recurse.c"
    src.write_text(_C_PROGRAM)
    exe = tmp_path / "recurse"
    lib = vg.lib_path()
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                    f"-L{lib.parent}", f"-l:{lib.name}", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    fonts = tmp_path / "fonts"
    fonts.mkdir()
    shutil.copy(FIRA, fonts / "Fira Sans - Regular.ttf")
    out = tmp_path / "out"
    p = subprocess.run([str(exe), str(fonts), str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "parts " in p.stdout
    assert (out / "index.json").exists() and (out / "font_families.json").exists()
    assert len(list((out / "fira_sans_regular").glob("*.pbf"))) == 256
