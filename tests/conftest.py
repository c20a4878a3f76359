"""pytest wiring: gpu marker, repo paths, and loaders for the oracle (checker) and the
product package (directory name has hyphens, so it is loaded by path)."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
TESTDATA = ROOT / "testdata"
GOLDEN = ROOT / "tests" / "golden"
FIRA = TESTDATA / "Fira Sans - Regular.ttf"
NOTO_DIR = TESTDATA / "Noto Sans"
NOTO = NOTO_DIR / "Noto Sans - Regular.ttf"

if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_product():
    """import versatiles-glyphs-rs_amd/ as module `versatiles_glyphs_rs_amd`."""
    name = "versatiles_glyphs_rs_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg = ROOT / "versatiles-glyphs-rs_amd"
    spec = importlib.util.spec_from_file_location(name, pkg / "__init__.py",
                                                  submodule_search_locations=[str(pkg)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def vg():
    return load_product()


@pytest.fixture(scope="session")
def fira_oracle(oracle):
    return oracle.Font(FIRA)


@pytest.fixture(scope="session")
def noto_oracle(oracle):
    return oracle.Font(NOTO)


def noto_files():
    """canonical merge order = sorted file names (pages/build.sh:6 shell glob)."""
    return sorted(NOTO_DIR.glob("*.ttf"), key=lambda p: p.name)
