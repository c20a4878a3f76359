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


def _pack_partials(files, font_id):
    """this rank's 256 partial PBFs of a font -> one uint8 buffer: 257 x u64 offsets, then the bytes (block order)"""
    names = [f"{font_id}/{b * 256}-{b * 256 + 255}.pbf" for b in range(256)]
    blobs = [files[n] for n in names]
    off = np.zeros(257, dtype=np.uint64)
    off[1:] = np.cumsum([len(b) for b in blobs])
    return np.concatenate([off.view(np.uint8), np.frombuffer(b"".join(blobs), dtype=np.uint8)])


def _unpack_partial(buf: np.ndarray, block: int) -> bytes:
    off = buf[: 257 * 8].view(np.uint64)
    return buf[257 * 8 + int(off[block]): 257 * 8 + int(off[block + 1])].tobytes()


def render_sharded_glyphs(mgr, renderer, font_id: str, writer, rank: int, world: int, dist=None, device=None):
    """Glyph-level sharding of one font over `world` ranks (the reference's unit is the
    (font, block) task, manager.rs:86-97, but 45 unequal non-empty blocks do not balance over 8 GPUs).

    1. every rank derives the same longest-processing-time-first assignment of glyphs to ranks from the
       font alone (FontManager.shard_glyphs: no communication);
    2. it renders its glyphs of EVERY block -> 256 partial PBFs;
    3. the partials travel to the block owners (block b belongs to rank b % world): one all-gather of the
       packed partials (RCCL under backend "nccl", gloo on the CPU) — the path's one real exchange step;
    4. every rank merges and writes its own blocks (pbf_merge: glyph messages as they are, ascending id), so
       the union of the ranks' files equals the single-process output byte for byte.
    Returns the world-wide counters."""
    from .host import DummyWriter, pbf_merge
    mgr.set_glyph_shard(rank, world)
    try:
        local = DummyWriter()
        mgr.render_glyphs(local, renderer, font_id=font_id, block_starts=range(0, 65536, 256))
        t = mgr.timings()
    finally:
        mgr.set_glyph_shard(0, 1)
    mine = _pack_partials(local.files, font_id)
    if dist is not None and world > 1:
        import torch
        dev = device or "cpu"
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        sizes[rank] = mine.size
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
        cap = int(sizes.max().item())
        send = torch.zeros(cap, dtype=torch.uint8, device=dev)
        send[: mine.size] = torch.from_numpy(mine.copy()).to(dev)
        recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(recv, send)
        recv = recv.cpu().numpy().reshape(world, cap)
        parts = [recv[r, : int(sizes[r].item())] for r in range(world)]
        c = torch.tensor([t["glyphs"], t["pixels"]], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        glyphs, pixels = (int(v) for v in c.tolist())
    else:
        assert world == 1, "world > 1 needs an initialised torch.distributed module"
        parts, glyphs, pixels = [mine], t["glyphs"], t["pixels"]
    if rank == 0:
        writer.write_directory(font_id + "/")
    my_blocks = list(range(rank, 256, world))
    for b in my_blocks:
        writer.write_file(f"{font_id}/{b * 256}-{b * 256 + 255}.pbf", pbf_merge([_unpack_partial(p, b) for p in parts]))
    return {"blocks": 256, "glyphs": glyphs, "pixels": pixels, "my_blocks": [b * 256 for b in my_blocks]}
