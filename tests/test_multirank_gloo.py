"""N > 1 path on CPU: two gloo ranks shard the blocks of a font (dummy renderer — the
sharding, the per-rank dispatch and the counter all-reduce are what is under test), and the
union of their outputs must equal the single-process output byte for byte."""
import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

from conftest import FIRA, ROOT, load_product


def _worker(rank, world, port, outdir):
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import load_product as lp, FIRA as fira
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    vg = lp()
    m = vg.FontManager(False)
    fid = m.add_font_with_name("Fira Sans Regular", [fira])
    w = vg.DummyWriter()
    res = vg.render_sharded(m, vg.Renderer.new_dummy(), fid, w, rank, world, dist=dist, device="cpu")
    with open(Path(outdir) / f"rank{rank}.pkl", "wb") as f:
        pickle.dump({"files": w.files, "res": res}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(vg):
    import torch.multiprocessing as mp
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        parts = [pickle.load(open(Path(d) / f"rank{r}.pkl", "rb")) for r in range(world)]
    m = vg.FontManager(False)
    fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
    w = vg.DummyWriter()
    m.render_glyphs(w, vg.Renderer.new_dummy())
    merged = {}
    for p in parts:
        assert not (set(p["files"]) & set(merged)), "ranks rendered overlapping blocks"
        merged.update(p["files"])
    assert merged == w.files
    t = m.timings()
    for p in parts:  # every rank holds the reduced, world-wide counters
        assert (p["res"]["blocks"], p["res"]["glyphs"], p["res"]["pixels"]) == (256, t["glyphs"], t["pixels"])
    assert sorted(parts[0]["res"]["my_blocks"] + parts[1]["res"]["my_blocks"]) == [i * 256 for i in range(256)]


def test_shard_blocks_balance(vg):
    m = vg.FontManager(False)
    fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
    costs = m.block_counts(fid)
    for world in (1, 2, 4, 8):
        shards = vg.shard_blocks(costs, world)
        allb = sorted(b for s in shards for b in s)
        assert allb == [i * 256 for i in range(256)]
        loads = [sum(int(costs[b // 256]) for b in s) for s in shards]
        assert max(loads) - min(loads) <= int(costs.max())  # LPT bound
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 20
