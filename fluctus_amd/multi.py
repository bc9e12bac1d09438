"""Multi-GPU plumbing: pixel-interleaved partition + gather of per-rank radiance tiles with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The reference is single-device
(one cl::CommandQueue, src/clcontext.cpp:25-29); the scheme is SURVEY 8(e): rank r of R owns global pixels
p*R + r, runs the whole wavefront loop on its own paths, and only the finished accumulation buffers travel."""
import torch
import torch.distributed as dist


def local_pixel_count(npix, rank, world):
    return (npix - rank + world - 1) // world if npix > rank else 1


def gather_tiles(tile, npix, rank, world, dst=0):
    """tile: (maxlp, 4) float32 tensor holding this rank's local pixels (padded to the largest tile).
    Returns the full (npix, 4) image on rank `dst` (None elsewhere)."""
    out = [torch.empty_like(tile) for _ in range(world)] if rank == dst else None
    dist.gather(tile, out, dst=dst)
    if rank != dst:
        return None
    return torch.stack(out, 1).reshape(-1, 4)[:npix]          # de-interleave: global pixel = p*world + r
