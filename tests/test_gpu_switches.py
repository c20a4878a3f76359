"""Every device-stage change of round 4 has a measurement switch that restores the older path (upload by the copy engine, the
outline_context launch, chunk boxes from the segments behind the flattening pass, read-back behind the raster).  Both sides of
every switch must give the golden files: the switches are read once per process, so each setting renders in a child."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CODE = r'''
import hashlib, json, sys
sys.path.insert(0, "tests")
from conftest import load_product, NOTO, FIRA
vg = load_product()
golden = json.load(open("tests/golden/pbf_sha256.json"))
r = vg.Renderer.new_precise(0)
bad = []
for key, name, path in (("noto_regular", "Noto Sans Regular", NOTO), ("fira", "Fira Sans Regular", FIRA)):
    m = vg.FontManager(True)
    fid = m.add_font_with_name(name, [path])
    for run in range(2):                     # (twice: the error words of a context alternate between submissions)
        w = vg.DummyWriter()
        m.render_glyphs(w, r)
        bad += [f"{key}/{s}" for s, h in golden[key].items() if hashlib.sha256(w.files[f"{fid}/{s}-{int(s) + 255}.pbf"]).hexdigest() != h]
    t = m.timings()
    assert t["glyf_groups"] == 1 and t["glyf_fallbacks"] == 0, t
print(json.dumps(bad))
'''


@pytest.mark.parametrize("switch", ["", "VGSDF_COPY_KERNEL", "VGSDF_FUSE_CONTEXT", "VGSDF_CMD_BOXES", "VGSDF_EARLY_COPY"])
def test_both_sides_of_every_switch_give_the_golden_files(switch):
    env = dict(os.environ)
    if switch:
        env[switch] = "0"
    cp = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    assert json.loads(cp.stdout.strip().splitlines()[-1]) == [], switch
