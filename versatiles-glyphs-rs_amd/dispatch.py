"""Multi-GPU sharding of the (font, block) task list — one process per GPU.

The reference's unit of parallel work is the 256-code-point GlyphBlock
(/root/reference/src/font/manager.rs:86-97,117-121); blocks are independent, so ranks take
disjoint subsets and no rendered data is ever exchanged.  The only collective is the final
3 x u64 {blocks, glyphs, pixels} counter all-reduce (RCCL under backend "nccl", gloo on CPU).
"""
from __future__ import annotations

import numpy as np


def shard_blocks(costs, world: int):
    """Longest-processing-time-first assignment of blocks to ranks.

    costs: per-block cost (e.g. FontManager.block_counts(font_id), or measured pair counts);
    zero-cost (empty) blocks are spread round-robin so every rank still emits its share of
    the empty PBFs.  -> list (len world) of sorted block start lists."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    shards = [[] for _ in range(world)]
    rr = 0
    for b in order:
        if costs[b] > 0:
            r = int(np.argmin(load))
            load[r] += costs[b]
        else:
            r = rr % world
            rr += 1
        shards[r].append(int(b) * 256)
    return [sorted(s) for s in shards]


def render_sharded(mgr, renderer, font_id: str, writer, rank: int, world: int, dist=None, device=None):
    """Render this rank's shard of `font_id`; returns the world-wide counters dict.

    dist: an initialised torch.distributed module (or None for world == 1).  device: torch
    device for the counter tensor ("cuda" with nccl, "cpu" with gloo)."""
    shards = shard_blocks(mgr.block_counts(font_id), world)
    mine = shards[rank]
    if rank == 0:
        writer.write_directory(font_id + "/")
    mgr.render_glyphs(writer, renderer, font_id=font_id, block_starts=mine)
    t = mgr.timings()
    counters = [t["blocks"], t["glyphs"], t["pixels"]]
    if dist is not None and world > 1:
        import torch
        c = torch.tensor(counters, dtype=torch.int64, device=device or "cpu")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = [int(v) for v in c.tolist()]
    return {"blocks": counters[0], "glyphs": counters[1], "pixels": counters[2], "my_blocks": mine}
