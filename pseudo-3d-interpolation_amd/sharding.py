"""Slice sharding across one-process-per-GPU ranks (replaces the reference's dask LocalCluster slice
farm, cube_POCS_interpolation_3D.py:291-340).

Slices are independent (one ``POCS_algorithm`` call each in the reference), so the slice axis is cut
into ``world`` contiguous blocks, every rank runs its block on its own GPU, and the blocks are put
together again with ONE collective at the end (``all_gather`` over RCCL/xGMI when the tensors live on
GPUs; the same code runs over gloo on CPU tensors, which is how the tests exercise it).
"""
import numpy as np


def slice_block(nslices, world, rank):
    """[lo, hi) of the contiguous block owned by ``rank``; the first ``nslices % world`` ranks own one
    slice more.  Blocks tile range(nslices) exactly and in rank order."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(nslices), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def block_sizes(nslices, world):
    return [slice_block(nslices, world, r)[1] - slice_block(nslices, world, r)[0] for r in range(world)]


def gather_blocks(local, nslices, group=None):
    """All-gather per-rank blocks (torch tensors, leading axis = slices of this rank) into the full
    cube on every rank.  Uneven blocks are padded to the largest one for the collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = block_sizes(nslices, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match the sharding of the slice axis")
    biggest = max(sizes)
    if local.shape[0] < biggest:
        pad = torch.zeros((biggest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def pocs_cube_sharded(cube, mask, group=None, compute=None, **params):
    """Run :func:`functions.POCS.pocs_cube` on this rank's block of ``cube`` (NumPy, the whole cube is
    visible to every rank, e.g. memory-mapped) and return the gathered result as a NumPy array.

    ``compute(block, mask, **params)`` defaults to the HIP ``pocs_cube`` on device ``LOCAL_RANK``.
    """
    import os

    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = slice_block(cube.shape[0], world, rank)
    if compute is None:
        from .functions.POCS import pocs_cube as compute  # noqa: N813
        params.setdefault("device", int(os.environ.get("LOCAL_RANK", rank)))
    block = np.ascontiguousarray(compute(np.asarray(cube[lo:hi]), mask, **params))
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.from_numpy(block.view(np.float32) if np.iscomplexobj(block) else block)
    if on_gpu:
        t = t.to(torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank))))
    full = gather_blocks(t, cube.shape[0], group).cpu().numpy()
    return full.view(block.dtype) if np.iscomplexobj(block) else full
