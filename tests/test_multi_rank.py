"""The N>1 path on CPU: 2 processes, torch.distributed gloo, each rank renders its interleaved pixel subset
with its own paths (oracle contexts stand in for the GPUs), tiles are gathered with the same helper bench.py
uses with RCCL."""
import os
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import common
from fluctus_amd import host, wire, driver, multi

W, H, N, ITERS = 40, 30, 2048, 30


def _render(rank, world):
    from oracle.binding import OracleContext
    d = common.simple_scene()
    p = common.scene_params(d, W, H, maxBounces=4)
    c = OracleContext(N)
    c.upload_scene(d)
    c.set_partition(rank, world)
    c.set_params(p)
    driver.reset_renderer(c)
    lp = multi.local_pixel_count(W * H, rank, world)
    newpaths = 0
    for it in range(ITERS):
        cnt = driver.benchmark_iteration(c, lp)         # the cursor runs over the rank's LOCAL pixels
        if it > 0:
            newpaths += int(cnt[0])
    px = c.read_pixels(0)[:lp]
    st = c.state_export().view(np.uint32)
    assert st[common.COL.PIXEL_INDEX].max() < lp
    return px, newpaths, lp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px, newpaths, lp = _render(rank, world)
    maxlp = (W * H + world - 1) // world
    tile = torch.zeros((maxlp, 4), dtype=torch.float32)
    tile[:lp] = torch.from_numpy(px)
    full = multi.gather_tiles(tile, W * H, rank, world)
    tot = torch.tensor([float(newpaths)], dtype=torch.float64)
    dist.all_reduce(tot)
    if rank == 0:
        q.put((full.numpy(), float(tot.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_tile_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, newpaths = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full.shape == (W * H, 4) and np.isfinite(full).all()
    # every path regenerated after the first iteration splats exactly once into exactly one rank's tile
    assert int(full[:, 3].sum()) == int(newpaths)
    # both ranks contribute: even pixels come from rank 0, odd pixels from rank 1
    assert full[0::2, 3].sum() > 0 and full[1::2, 3].sum() > 0
    # statistical parity with the single-rank render of the same scene
    one, _, _ = _render(0, 1)
    a = full[:, :3].sum() / full[:, 3].sum()
    b = one[:, :3].sum() / one[:, 3].sum()
    assert abs(a - b) <= 0.05 * b


def test_partition_arithmetic():
    for npix in (1, 7, 1200, 2073600):
        for world in (1, 2, 3, 8):
            counts = [multi.local_pixel_count(npix, r, world) for r in range(world)]
            owned = sum(len(range(r, npix, world)) for r in range(world))
            assert owned == npix
            for r in range(world):
                assert counts[r] == max(1, len(range(r, npix, world)))
