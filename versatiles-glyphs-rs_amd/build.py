"""In-tree build of libvgsdf.so (hipcc, --offload-arch=gfx950)."""
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent


def lib_path() -> Path:
    # VGSDF_LIB: A/B experiments with alternative in-tree builds (development only)
    return Path(os.environ["VGSDF_LIB"]) if os.environ.get("VGSDF_LIB") else PKG / "libvgsdf.so"


def build(force: bool = False) -> Path:
    """Compile every HIP/C++ source for gfx950.  hipcc cross-compiles without a GPU."""
    cmd = ["make", "-C", str(PKG), "-s"]
    if force:
        subprocess.run(cmd + ["clean"], check=True)
    subprocess.run(cmd, check=True)
    if not lib_path().exists():
        raise RuntimeError("libvgsdf.so was not produced")
    return lib_path()
