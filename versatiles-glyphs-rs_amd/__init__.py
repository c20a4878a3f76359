"""MI355X-native SDF glyph renderer — Python plumbing over the C-ABI library.

The product is the native code in csrc/ (HIP kernels + C++ host, built in-tree as
libvgsdf.so).  This package only binds its C ABI with ctypes for tests, bench.py and
multi-GPU launch plumbing.  It never imports anything from oracle/ and has no CPU
fallback: if libvgsdf.so or a HIP device is missing, calls raise.

The directory name contains hyphens; load it by path (see tests/conftest.py:load_product).
"""
from .build import build, lib_path  # noqa: F401
from .device import (  # noqa: F401
    Batch, DeviceBatch, OUTLINE_CMD_DTYPE, RECT_DTYPE, SdfContext, VgsdfError, device_count, make_batch, load_library, reduce_counters, reduce_path,
)
from .host import (  # noqa: F401
    DummyWriter, FontManager, GlyphBatchHost, NativeWriter, PbfGlyph, Renderer, encode_codeblocks, name_to_id, parse_font_name,
    pbf_encode, pbf_merge,
)
from .dispatch import render_sharded, render_sharded_glyphs, shard_blocks  # noqa: F401
