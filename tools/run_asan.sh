#!/bin/bash
# CPU-only sanitizer pass: the host C++ (TrueType reader, flattening, batching, PBF, C API) built with
# ASan + UBSan, driven by the CPU test files that exercise it, and the oracle's C under the same.
set -e
cd "$(dirname "$0")/.."
make -C versatiles-glyphs-rs_amd asan > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
# the oracle with the same sanitizers and the same (clang) runtime, so both libraries can live in one process
mkdir -p oracle/build
/opt/rocm/lib/llvm/bin/clang -O1 -g -std=c11 -ffp-contract=off -fsanitize=address,undefined -shared-libsan -fPIC -shared \
  -o oracle/build/libvgoracle_asan.so oracle/vg_oracle.c -lm -lpthread
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
  VGSDF_LIB=$PWD/versatiles-glyphs-rs_amd/build/asan/libvgsdf.so VG_ORACLE_LIB=$PWD/oracle/build/libvgoracle_asan.so \
  python -m pytest tests/test_host_facade.py tests/test_host_vs_oracle.py tests/test_capi_exports.py \
    tests/test_cff_outlines.py tests/test_cff2_outlines.py tests/test_glyf_parts_host.py tests/test_font_collections.py tests/test_ingestion_and_sinks.py tests/test_pbf_layout_boundaries.py tests/test_large_font.py tests/test_glyph_sharding.py tests/test_composite_fanout.py tests/test_lane_plan.py \
    "tests/test_malformed_fonts.py::test_damaged_fonts_never_crash" -x -q -m "not gpu" -k "not links_from_c and not gloo"
