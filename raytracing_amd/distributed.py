"""Multi-GPU split of one render: sample ranges per rank and one sum-reduce at the end
(render_distributed), or tiles per rank and one gather (render_tile_split).

Every pixel-sample is independent, so the path shards without any exchange while tracing:
the scene is replicated on every GPU, rank g traces passes [g*S/N, (g+1)*S/N) of the whole
frame into its own float32 accumulator using the GLOBAL pass index in the seed schedule, and
the N accumulators are summed once (`torch.distributed.reduce`, which is RCCL over xGMI with
the "nccl" backend).  The union of the work equals the single-GPU render exactly; only the
order of the final float additions differs.

The reference has no counterpart (it drives one Vulkan device); its only scaling axis is
tiling dispatches in time (src/renderer/megakernel_raytracer.cpp:58-78).
"""
import torch
import torch.distributed as dist


def pass_range(total_passes, world_size, rank):
    """Contiguous, balanced [first, first+count) slice of the passes for `rank`."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_passes, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def reduce_accumulator(accum, dst=0, group=None):
    """Sum the per-rank RGBA accumulators onto rank `dst` (in place).

    The alpha channel of every accumulator is 1 (the shader writes vec4(colour, 1)); after the
    sum it is reset to 1 on the destination so the image keeps the reference's convention.
    `dst` is a GLOBAL rank, as torch.distributed.reduce takes it, also when `group` is a
    sub-group.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum
    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank() == dst:
        accum[..., 3] = 1.0
    return accum


def render_distributed(render_passes, total_passes, accum, group=None, dst=0):
    """Runs this rank's share of `total_passes` and reduces.

    render_passes(first_pass, pass_count, accum) must add those passes of the full frame to
    `accum` (a float32 tensor [H, W, 4] on this rank's device).
    Returns (first_pass, pass_count) traced by this rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = pass_range(total_passes, world, rank)
    if count > 0:
        render_passes(first, count, accum)
    reduce_accumulator(accum, dst=dst, group=group)
    return first, count


def tile_share(tiles_x, tiles_y, world_size, rank):
    """Tiles (tx, ty) of `rank` in the tile split: tiles numbered row-major and dealt round
    robin, so every rank gets tiles from all over the picture (an expensive region is shared)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return [(t % tiles_x, t // tiles_x) for t in range(rank, tiles_x * tiles_y, world_size)]


def render_tile_split(render_rect, total_passes, accum, tile, group=None, dst=0):
    """The other way to share a render: every rank traces ALL passes of its own tiles.

    Each pixel is then computed by exactly one rank, so the combined image equals the
    single-GPU one bit for bit (no float sum is re-associated), and combining is a gather of
    disjoint tiles instead of a sum: every rank sends only its 1/N of the frame.  Suits few
    passes of a large frame (interactive use); the sample-range split (render_distributed)
    suits many passes.

    render_rect(rect, first_pass, pass_count, accum) must add those passes of the pixels
    rect = (x0, y0, x1, y1) to `accum` ([H, W, 4] float32 on this rank's device).
    Returns the tiles this rank traced.
    """
    height, width = accum.shape[0], accum.shape[1]
    tw, th = tile
    if width % tw or height % th:
        raise ValueError("the frame is not a whole number of tiles")
    tiles_x, tiles_y = width // tw, height // th
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = tile_share(tiles_x, tiles_y, world, rank)
    for tx, ty in mine:
        render_rect((tx * tw, ty * th, (tx + 1) * tw, (ty + 1) * th), 0, total_passes, accum)
    if world == 1:
        return mine
    # pack this rank's tiles; ranks with one tile fewer pad, so that all messages are equal
    most = -(-tiles_x * tiles_y // world)
    packed = torch.zeros((most, th, tw, 4), dtype=accum.dtype, device=accum.device)
    for k, (tx, ty) in enumerate(mine):
        packed[k] = accum[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
    on_dst = dist.get_rank() == dst
    parts = [torch.empty_like(packed) for _ in range(world)] if on_dst else None
    dist.gather(packed, parts, dst=dst, group=group)
    if on_dst:
        for r, part in enumerate(parts):
            if r == rank:
                continue
            for k, (tx, ty) in enumerate(tile_share(tiles_x, tiles_y, world, r)):
                accum[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = part[k]
    return mine


def new_accumulator(height, width, device):
    """Zero colour, alpha 1: the state after the reference's clearColorImage."""
    accum = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
    accum[..., 3] = 1.0
    return accum
