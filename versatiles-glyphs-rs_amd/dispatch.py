"""One process per GPU (torchrun): sharding of the (font, block) task list, or of a font's glyphs, over the ranks.

The library itself drives N devices from ONE process (Renderer.new_multi; csrc/host/font_manager.cpp,
render_glyphs_multi) — that is the form a host application links.  This module is the launcher-side plumbing for the
one-process-per-GPU form the benchmark driver uses: the reference's unit of parallel work is the 256-code-point
GlyphBlock (/root/reference/src/font/manager.rs:86-97,117-121); whole blocks need no exchange at all
(render_sharded), glyph-level shards need one all-to-all of partial PBFs to the block owners
(render_sharded_glyphs); the counters {blocks, glyphs, pixels} are all-reduced (RCCL under backend "nccl", gloo on CPU).
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
    writer.write_directory(font_id + "/")  # on every rank: each writes its own blocks, possibly under its own root
    mgr.render_glyphs(writer, renderer, font_id=font_id, block_starts=mine)
    t = mgr.timings()
    counters = [t["blocks"], t["glyphs"], t["pixels"]]
    if dist is not None and world > 1:
        import torch
        c = torch.tensor(counters, dtype=torch.int64, device=device or "cpu")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = [int(v) for v in c.tolist()]
    return {"blocks": counters[0], "glyphs": counters[1], "pixels": counters[2], "my_blocks": mine}


def _pack_for(files, font_id, dest: int, world: int) -> np.ndarray:
    """this rank's partial PBFs of the blocks rank `dest` owns (block b belongs to rank b % world), as one uint8
    buffer: u64 lengths of those blocks, then their bytes, in block order"""
    blobs = [files[f"{font_id}/{b * 256}-{b * 256 + 255}.pbf"] for b in range(dest, 256, world)]
    lens = np.array([len(b) for b in blobs], dtype=np.uint64)
    return np.concatenate([lens.view(np.uint8), np.frombuffer(b"".join(blobs), dtype=np.uint8)])


def _unpack_from(buf: np.ndarray, n_blocks: int):
    lens = buf[: 8 * n_blocks].view(np.uint64)
    off = 8 * n_blocks + np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return [buf[off[i]: off[i + 1]].tobytes() for i in range(n_blocks)]


def render_sharded_glyphs(mgr, renderer, font_id: str, writer, rank: int, world: int, dist=None, device=None):
    """One process per GPU: glyph-level sharding of one font over `world` ranks (the reference's unit is
    the (font, block) task, manager.rs:86-97, but 45 unequal non-empty blocks do not balance over 8 GPUs).  (ONE process
    driving N devices needs none of this: Renderer.new_multi + FontManager.render_glyphs merge in shared memory.)

    1. every rank derives the same longest-processing-time-first assignment of glyphs to ranks from the
       font alone (FontManager.shard_glyphs: no communication);
    2. it renders its glyphs of EVERY block -> 256 partial PBFs;
    3. the partials travel to the block owners (block b belongs to rank b % world): ONE all-to-all in which rank r sends
       rank d only the partials of d's blocks (RCCL under backend "nccl", gloo on the CPU) — total traffic = the
       partials once, instead of every rank receiving every partial;
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
    my_blocks = list(range(rank, 256, world))
    if dist is not None and world > 1:
        import torch
        dev = device or "cpu"
        outgoing = [_pack_for(local.files, font_id, d, world) for d in range(world)]
        in_sizes = torch.tensor([o.size for o in outgoing], dtype=torch.int64, device=dev)
        out_sizes = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(out_sizes, in_sizes)  # how much every peer is about to send me
        out_split = [int(v) for v in out_sizes.tolist()]
        send = torch.from_numpy(np.concatenate(outgoing)).to(dev)
        recv = torch.empty(sum(out_split), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, out_split, [o.size for o in outgoing])
        recv = recv.cpu().numpy()
        offs = np.concatenate([[0], np.cumsum(out_split)])
        per_rank = [_unpack_from(recv[offs[r]: offs[r + 1]], len(my_blocks)) for r in range(world)]
        c = torch.tensor([t["glyphs"], t["pixels"]], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        glyphs, pixels = (int(v) for v in c.tolist())
    else:
        assert world == 1, "world > 1 needs an initialised torch.distributed module"
        per_rank = [_unpack_from(_pack_for(local.files, font_id, 0, 1), 256)]
        glyphs, pixels = t["glyphs"], t["pixels"]
    # every rank makes sure the font's directory exists where IT writes (idempotent for a directory sink; a tar per rank
    # needs its own entry anyway)
    writer.write_directory(font_id + "/")
    for i, b in enumerate(my_blocks):
        writer.write_file(f"{font_id}/{b * 256}-{b * 256 + 255}.pbf", pbf_merge([p[i] for p in per_rank]))
    return {"blocks": 256, "glyphs": glyphs, "pixels": pixels, "my_blocks": [b * 256 for b in my_blocks]}
