"""Multi-GPU split logic on CPU: pass partition, and a world_size-2 `gloo` run in which each
rank renders its pass range (with the CPU oracle standing in for the GPU kernel) and the
accumulators are reduced exactly as bench.py and the renderer do it on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import distributed as rd

from conftest import ROOT


def test_pass_range_partitions_exactly():
    for total in (0, 1, 7, 64, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for rank in range(world):
                first, count = rd.pass_range(total, world, rank)
                covered += list(range(first, first + count))
                assert count in (total // world, total // world + 1)
            assert covered == list(range(total))
    with pytest.raises(ValueError):
        rd.pass_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_passes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from raytracing_amd import host, scenes
    sc = scenes.load_snapshot("cornell_box")
    w, h, tile, md = 64, 64, (16, 16), 4
    cam = sc.camera(w, h)
    tiles = 16

    def render_passes(first, count, accum):
        seeds = host.seed_schedule(first, count, tiles)   # GLOBAL pass indices
        buf = accum.numpy()
        oracle.render(sc.desc, cam, w, h, md, tile, seeds, accum=buf, first_pass=first,
                      threads=2)

    accum = rd.new_accumulator(h, w, "cpu")
    first, count = rd.render_distributed(render_passes, total_passes, accum)
    np.save(os.path.join(out_dir, f"range_{rank}.npy"), np.array([first, count]))
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_render_equals_single_rank(tmp_path, cornell):
    import oracle
    from raytracing_amd import host
    total = 6
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    ranges = [tuple(np.load(tmp_path / f"range_{r}.npy")) for r in range(2)]
    assert ranges == [(0, 3), (3, 3)]
    reduced = np.load(tmp_path / "reduced.npy")
    w, h, tile, md = 64, 64, (16, 16), 4
    seeds = host.seed_schedule(0, total, 16)
    single, _ = oracle.render(cornell.desc, cornell.camera(w, h), w, h, md, tile, seeds)
    assert (reduced[..., 3] == 1.0).all()
    # same samples, different association of the float sum: (a+b+c) + (d+e+f) vs a+b+...+f
    assert np.allclose(reduced, single, rtol=2e-6, atol=1e-6)
    # and the two halves really are different work: one half alone is not the whole
    half, _ = oracle.render(cornell.desc, cornell.camera(w, h), w, h, md, tile, seeds[:3])
    assert not np.allclose(half, single, rtol=1e-3, atol=1e-3)


def test_four_rank_gloo_render_with_an_uneven_pass_count(tmp_path, cornell):
    """7 passes over 4 ranks: shares of 2, 2, 2 and 1.  The union must be the single-rank
    render, alpha included."""
    import oracle
    from raytracing_amd import host
    total, world = 7, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    ranges = [tuple(int(v) for v in np.load(tmp_path / f"range_{r}.npy")) for r in range(world)]
    assert ranges == [(0, 2), (2, 2), (4, 2), (6, 1)]
    reduced = np.load(tmp_path / "reduced.npy")
    w, h, tile, md = 64, 64, (16, 16), 4
    seeds = host.seed_schedule(0, total, 16)
    single, _ = oracle.render(cornell.desc, cornell.camera(w, h), w, h, md, tile, seeds)
    assert (reduced[..., 3] == 1.0).all()
    assert np.allclose(reduced, single, rtol=2e-6, atol=1e-6)


def _subgroup_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group([1, 2])   # global ranks 1 and 2; group ranks 0 and 1
    accum = rd.new_accumulator(2, 2, "cpu")
    accum[..., :3] = float(rank + 1)
    if rank in (1, 2):
        rd.reduce_accumulator(accum, dst=2, group=group)   # dst is the GLOBAL rank
    np.save(os.path.join(out_dir, f"sub_{rank}.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_onto_a_global_rank_of_a_subgroup(tmp_path):
    """dst names a global rank also when a sub-group reduces: the alpha reset must land on
    global rank 2 (group rank 1), not on the rank whose group rank happens to equal dst."""
    port = _free_port()
    mp.spawn(_subgroup_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    on_dst = np.load(tmp_path / "sub_2.npy")
    assert (on_dst[..., :3] == 2.0 + 3.0).all() and (on_dst[..., 3] == 1.0).all()
    untouched = np.load(tmp_path / "sub_0.npy")
    assert (untouched[..., :3] == 1.0).all()


def test_tile_share_deals_every_tile_once():
    for tiles_x, tiles_y, world in ((10, 10, 8), (5, 3, 4), (2, 2, 8), (7, 1, 3)):
        seen = []
        for rank in range(world):
            seen += rd.tile_share(tiles_x, tiles_y, world, rank)
        assert sorted(seen) == sorted((x, y) for y in range(tiles_y) for x in range(tiles_x))
    with pytest.raises(ValueError):
        rd.tile_share(4, 4, 2, 2)


def _tile_worker(rank, world, port, total_passes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from raytracing_amd import host, scenes
    sc = scenes.load_snapshot("cornell_box")
    w, h, tile, md = 80, 48, (16, 16), 4     # 5 x 3 = 15 tiles over 4 ranks: 4, 4, 4, 3
    cam = sc.camera(w, h)
    seeds = host.seed_schedule(0, total_passes, 15)

    def render_rect(rect, first, count, accum):
        oracle.render(sc.desc, cam, w, h, md, tile, seeds[first:first + count],
                      accum=accum.numpy(), first_pass=first, threads=2, rect=rect)

    accum = rd.new_accumulator(h, w, "cpu")
    mine = rd.render_tile_split(render_rect, total_passes, accum, tile)
    np.save(os.path.join(out_dir, f"tiles_{rank}.npy"), np.array(mine))
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_tile_split_is_bit_exact(tmp_path, cornell):
    """Every pixel is traced by one rank only, so the gathered image equals the single-rank
    render bit for bit - no float sum changes its order."""
    import oracle
    from raytracing_amd import host
    total, world = 3, 4
    port = _free_port()
    mp.spawn(_tile_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    counts = [len(np.load(tmp_path / f"tiles_{r}.npy")) for r in range(world)]
    assert counts == [4, 4, 4, 3]
    gathered = np.load(tmp_path / "gathered.npy")
    w, h, tile, md = 80, 48, (16, 16), 4
    single, _ = oracle.render(cornell.desc, cornell.camera(w, h), w, h, md, tile,
                              host.seed_schedule(0, total, 15))
    assert np.array_equal(gathered, single, equal_nan=True)


def test_reduce_is_identity_without_a_process_group():
    accum = rd.new_accumulator(4, 4, "cpu")
    accum[..., :3] = 2.0
    out = rd.reduce_accumulator(accum)
    assert out is accum and (accum[..., :3] == 2.0).all() and (accum[..., 3] == 1.0).all()


def _bench_module():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _schedule(bench, world, rank, steps, warmup, total_spp, spp=8):
    args = bench.parse_args(["--spp-per-step", str(spp)] +
                            (["--total-spp", str(total_spp)] if total_spp else []))
    return bench.call_schedule(bench.Workload(args), world, rank, steps, warmup)


def test_bench_strong_scaling_schedule_covers_every_pass_once():
    """bench.py --total-spp S: in every step the ranks' render calls together cover the global
    passes [s * S, (s + 1) * S) exactly once, in calls of at most --spp-per-step passes; the
    set-up calls use passes of their own."""
    bench = _bench_module()
    steps, warmup = 3, 1
    for world in (1, 2, 3, 8):
        for total in (5, 64, 100):
            seen = {}
            for rank in range(world):
                calls, starts = _schedule(bench, world, rank, steps, warmup, total)
                assert len(starts) == bench.SETUP_CALLS + warmup + steps
                bounds = starts + [len(calls)]
                for s in range(len(starts)):
                    for first, count in calls[bounds[s]:bounds[s + 1]]:
                        assert 1 <= count <= 8
                        for p in range(first, first + count):
                            assert p not in seen, (world, total, rank, p)
                            seen[p] = (s, rank)
            for s in range(warmup + steps):
                step = [p for p, (k, _) in seen.items() if k == bench.SETUP_CALLS + s]
                assert sorted(step) == list(range(s * total, (s + 1) * total))
            # set-up passes lie beyond every step's
            setup = [p for p, (k, _) in seen.items() if k < bench.SETUP_CALLS]
            assert min(setup) >= (warmup + steps) * total
    # weak scaling: every rank the same number of passes, ranges disjoint
    owned = [set() for _ in range(4)]
    for rank in range(4):
        calls, _ = _schedule(bench, 4, rank, steps, warmup, None)
        assert [c for _, c in calls] == [8] * (bench.SETUP_CALLS + warmup + steps)
        for first, count in calls:
            owned[rank] |= set(range(first, first + count))
    assert all(not (owned[a] & owned[b]) for a in range(4) for b in range(a))


def _bench_rank(rank, world, port, total_spp, out_dir):
    """A bench.py rank with a stub in place of the renderer: a "render" of pass p adds p + 1 to
    the rank's accumulator; the schedule and the reduce are the bench's own."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    steps, warmup = 2, 1
    calls, starts = _schedule(bench, world, rank, steps, warmup, total_spp)
    bounds = starts + [len(calls)]
    accum = rd.new_accumulator(2, 2, "cpu")
    mine = []
    for s in range(bench.SETUP_CALLS + warmup, bench.SETUP_CALLS + warmup + steps):
        for first, count in calls[bounds[s]:bounds[s + 1]]:
            for p in range(first, first + count):
                accum[..., :3] += float(p + 1)
                mine.append(p)
    rd.reduce_accumulator(accum, dst=0)
    np.save(os.path.join(out_dir, f"passes_{rank}.npy"), np.array(mine, dtype=np.int64))
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_bench_ranks_share_the_passes_of_every_step(tmp_path):
    total = 37
    mp.spawn(_bench_rank, args=(2, _free_port(), total, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"passes_{r}.npy") for r in range(2))
    assert not set(a) & set(b)
    # the two timed steps follow one warm-up step
    assert sorted(list(a) + list(b)) == list(range(total, 3 * total))
    reduced = np.load(tmp_path / "sum.npy")
    want = sum(p + 1 for p in range(total, 3 * total))
    assert (reduced[..., :3] == want).all() and (reduced[..., 3] == 1.0).all()
