"""Multi-GPU split of one render: sample ranges per rank, one sum-reduce at the end.

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


def new_accumulator(height, width, device):
    """Zero colour, alpha 1: the state after the reference's clearColorImage."""
    accum = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
    accum[..., 3] = 1.0
    return accum
