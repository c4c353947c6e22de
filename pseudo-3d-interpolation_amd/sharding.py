"""Slice sharding across one-process-per-GPU ranks (replaces the reference's dask LocalCluster slice
farm, cube_POCS_interpolation_3D.py:291-340).

Slices are independent (one ``POCS_algorithm`` call each in the reference), so the slice axis is cut
into ``world`` contiguous blocks and every rank runs its block on its own GPU.  How the blocks come
together again is the caller's choice (``pocs_cube_sharded(..., gather=...)``):

``'none'``  every rank moves ITS OWN block host -> device -> host, straight from the caller's cube into
            the caller's result array (``out=``: shared memory, a memory-mapped file -- what the step-13
            driver merges its batch files into).  No collective at all, and the PCIe links of all GPUs
            work side by side: a 4-GiB cube in and out is ~0.14 s through one link, ~0.02 s through
            eight.  This is the end-to-end path.
``'root'``  ONE collective on device tensors at the end, ``gather`` to rank 0 (the "trivial gather" of
            north_star: only the rank that consumes the cube on its GPU needs all of it), or
``'all'``   ``all_gather`` (every rank ends up with the cube).  On GPUs the collective moves DEVICE
            tensors over RCCL / xGMI -- the block never visits the host between the last kernel and the
            collective; the same code runs over gloo on CPU tensors, which is how the tests exercise it.

``import torch`` comes BEFORE the first call into the HIP library in a process that uses both (torch's
copy of the HIP runtime has to initialise first, _ffi._preload_torch_hip): import this module -- or
torch -- before ``functions.POCS`` does any work.
"""
import os

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


def _padded(local, nslices, group):
    """(block padded to the largest block of the sharding, sizes of all blocks)."""
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
    return local.contiguous(), sizes


def gather_blocks(local, nslices, group=None):
    """All-gather per-rank blocks (torch tensors, leading axis = slices of this rank) into the full
    cube on every rank.  Uneven blocks are padded to the largest one for the collective."""
    import torch
    import torch.distributed as dist

    if local.is_complex():   # not every backend moves complex elements (gloo's gather refuses them): as (re, im) pairs
        return torch.view_as_complex(gather_blocks(torch.view_as_real(local), nslices, group))
    local, sizes = _padded(local, nslices, group)
    parts = [torch.empty_like(local) for _ in sizes]
    dist.all_gather(parts, local, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def gather_blocks_to_root(local, nslices, group=None, root=0):
    """Gather the per-rank blocks on rank ``root`` only (``dist.gather``: one receive per peer on the root, one send on everybody
    else -- 1/world of all_gather's traffic and no second copy of the cube on the other ranks).  Returns the full cube on the
    root and an empty tensor (0 slices) elsewhere."""
    import torch
    import torch.distributed as dist

    if local.is_complex():
        return torch.view_as_complex(gather_blocks_to_root(torch.view_as_real(local), nslices, group, root))
    local, sizes = _padded(local, nslices, group)
    me = dist.get_rank(group)
    dst = dist.get_global_rank(group, root) if group is not None else root
    parts = [torch.empty_like(local) for _ in sizes] if me == root else None
    dist.gather(local, parts, dst=dst, group=group)
    if me != root:
        return local[:0]
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def _upload(arr, dev):
    """A C-contiguous NumPy array -> a tensor on ``dev``, page-locked IN PLACE for the copy (``_ffi.host_register``: a DMA transfer at the link's
    rate instead of the runtime's staged copy of pageable memory, 26 GB/s on the MI355X boxes -- profiles/r04_pcie_probe.txt).  A refused
    registration (read-only mapping, memory the caller locked itself) just means the pageable copy."""
    import torch

    from . import _ffi

    t = torch.from_numpy(arr)
    if dev.type != "cuda" or arr.nbytes < (8 << 20):
        return t.to(dev)
    pinned = _ffi.host_register(arr)
    try:
        return t.to(dev)
    finally:
        if pinned:
            _ffi.host_unregister(arr)


def _download(t):
    """A device tensor -> a new NumPy array whose fresh pages are touched by a few threads and page-locked for the copy (a download into
    untouched pageable pages runs at a third of the link's rate)."""
    import torch

    from . import _ffi
    from .functions import POCS as P

    if not t.is_cuda or t.numel() * t.element_size() < (8 << 20):
        return t.cpu().numpy()
    res = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    for f in P._touch_pages(res):
        f.result()
    pinned = _ffi.host_register(res)
    try:
        torch.from_numpy(res).copy_(t)
    finally:
        if pinned:
            _ffi.host_unregister(res)
    return res


def pocs_block_on_device(block, mask, device=0, **params):
    """``functions.POCS.pocs_cube`` for one rank's block with the RESULT LEFT ON THE GPU: the block is uploaded once into a torch
    tensor, statistics / schedule / iterations run on raw device pointers and the result comes back as a device tensor (complex64 or
    float32) -- what the gather collective wants.  Covers the FFT transform with a statistics-driven schedule (``Plan.prime_dev`` +
    ``run_dev``) and the WAVELET / SHEARLET transforms (``WaveletPlan`` / ``ShearletPlan.stats_dev`` + ``run_dev``; BASELINE
    configs[4] is the 8-GPU SHEARLET configuration).  The arguments are checked exactly as ``pocs_cube`` checks them, before
    anything is uploaded.  ``results`` / ``batch_slices`` (per-slice records, caller-chosen chunks), the data-driven schedule and
    cubes that are neither complex64 nor float32 go through ``pocs_cube`` itself (host arrays) and are uploaded afterwards."""
    import torch

    from . import _ffi
    from .functions import POCS as P

    thresh_op = params.get("thresh_op", "hard")
    version = params.get("version", "regular")
    block, mask, kind, niter, eps, p_max, alpha, p_min = P._check_cube_args(
        block, mask, params.get("transform_kind", "FFT"), thresh_op, version, params.get("niter", 50), params.get("eps", 1e-9),
        params.get("p_max", 0.99), params.get("alpha", 1.0), params.get("p_min", 1e-5))
    dev = torch.device("cuda", int(device))
    model = params.get("thresh_model", "exponential")
    decay_kind = params.get("decay_kind", "values")
    n, nil, nxl = block.shape
    psi = params.get("auxiliary_data")
    fast = (n > 0 and block.dtype in (np.complex64, np.float32) and params.get("results") is None and not params.get("batch_slices"))
    # a job that asks for (or is routed to) one of the double-precision loops goes through pocs_cube: the resident path below is the float32 kernels'
    fast = fast and not P._double_loop_wanted(block.dtype, nil, nxl, kind, thresh_op, params.get("precision"), params.get("wavelet"), params.get("transform"), psi)[1]
    if kind == "FFT":
        fast = fast and model != "data-driven" and thresh_op in _ffi.P3D_OP
    else:
        fast = fast and thresh_op in P._WAVELET_OPS and not (kind == "SHEARLET" and psi is None)
        if kind == "WAVELET" and decay_kind == "factors" and not all(s in model for s in ["inverse", "proportional"]):
            fast = False   # (pocs_cube raises the reference's IndexError there)
    if not fast:
        res = np.ascontiguousarray(P.pocs_cube(block, mask, device=int(device), **params))
        return _upload(res, dev)
    x = _upload(np.ascontiguousarray(block), dev)
    out = torch.empty_like(x)
    m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32)).to(dev)
    dt = _ffi.P3D_C64 if np.iscomplexobj(block) else _ffi.P3D_F32
    esz = 8 if dt == _ffi.P3D_C64 else 4
    torch.cuda.synchronize(dev)      # the plans work on their own streams
    common = dict(thresh_op=thresh_op, version=version, eps=eps, alpha=alpha)
    if kind == "FFT":
        plan = P._get_plan(nil, nxl, n, int(device), slot=14)
        stats = plan.prime_dev(x.data_ptr(), dt, m.data_ptr(), n)
        active = ~(stats[:, 2] == 0)
        stats[~active] = 1.0
        tau = P._schedule_from_stats(stats, nil * nxl, model, niter, p_max, p_min, decay_kind)
        if params.get("sqrt_decay", False):
            tau = np.sqrt(tau)
        plan.run_dev(x.data_ptr(), dt, m.data_ptr(), tau, niter, out.data_ptr(), n, active=active, primed=True, want_sums=False, **common)
        return out
    # WAVELET / SHEARLET: the same statistics -> schedule -> loop as pocs_cube, batch by batch on device pointers
    if kind == "SHEARLET":
        psi = np.asarray(psi)
        if psi.ndim != 3 or psi.shape[:2] != (nil, nxl):
            raise ValueError(f"Psi must be ({nil}, {nxl}, nshearlets), got shape {psi.shape}")
        step = max(1, min(int((8 << 30) // (psi.shape[2] * nil * nxl * 8)), 65535 // psi.shape[2], n))
        plan = P._get_shearlet_plan(psi, step, int(device))
    else:
        step = n
        plan = P._get_wavelet_plan(nil, nxl, n, P._wavelet_name(params.get("transform"), params.get("wavelet")), int(device))
    per = nil * nxl * esz
    for lo in range(0, n, step):
        nb = min(step, n - lo)
        xp, op = x.data_ptr() + lo * per, out.data_ptr() + lo * per
        stats = plan.stats_dev(xp, dt, nb)
        active = (x[lo:lo + nb].reshape(nb, -1) != 0).any(dim=1).cpu().numpy()   # np.count_nonzero(x) == 0 -> untouched (POCS.py:515-521)
        stats[~active] = 1.0
        stats[~active, ..., 1] = 0.0
        if kind == "WAVELET":
            tau = P._wavelet_schedule_from_stats(stats, model, niter, p_max, p_min, decay_kind)
        else:
            tau = P._shearlet_schedule_from_stats(stats, (nil, nxl), model, niter, p_max, p_min, decay_kind)
        if params.get("sqrt_decay", False):
            tau = np.sqrt(tau)
        plan.run_dev(xp, dt, m.data_ptr(), tau, niter, op, nb, active=active, **common)
    return out


def pocs_cube_sharded(cube, mask, group=None, compute=None, gather="root", out=None, **params):
    """Run the POCS interpolation of ``cube`` (NumPy, the whole cube is visible to every rank, e.g. memory-mapped) sharded over the
    ranks of ``group``.

    ``gather='none'``: every rank interpolates its own block straight into ``out[lo:hi]`` -- ``out`` an array of the cube's shape and
    dtype that all ranks see (shared memory, ``np.memmap`` / ``np.lib.format.open_memmap``) -- through the host-buffer entry point
    (``functions.POCS.pocs_cube``: the rank's block page-locked in place, chunks up and down on its own PCIe link); no collective but the
    barrier that tells every rank the result is complete.  Returns ``out`` (without ``out=``: the rank's own block, a new array).
    ``gather='root'`` / ``'all'``: the gathered cube as a NumPy array on rank 0 only (``None`` elsewhere) / on every rank, through ONE
    collective on device tensors.

    ``compute(block, mask, **params)`` defaults to the HIP path on device ``LOCAL_RANK``: with a collective the result is kept on the
    device (:func:`pocs_block_on_device`), so that over RCCL the collective moves device tensors and only the gathered cube is
    downloaded; a ``compute`` that returns NumPy (the gloo tests inject one) is gathered on CPU tensors / written to ``out``.
    """
    import torch
    import torch.distributed as dist

    if gather not in ("root", "all", "none"):
        raise ValueError("gather must be 'root', 'all' or 'none'")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = slice_block(cube.shape[0], world, rank)
    on_gpu = dist.get_backend(group) == "nccl"
    local_dev = int(params.pop("device", os.environ.get("LOCAL_RANK", rank)))   # (a caller's device=... wins over LOCAL_RANK)
    if out is not None and (gather != "none" or tuple(out.shape) != tuple(cube.shape) or out.dtype != cube.dtype):
        raise ValueError("out= goes with gather='none' and has the shape and dtype of the cube")
    if gather == "none":
        block = np.asarray(cube[lo:hi])
        if compute is None:
            from .functions import POCS as P
            res = P.pocs_cube(block, mask, device=local_dev, out=None if out is None else out[lo:hi], **params)
        else:
            res = np.asarray(compute(block, mask, **params))
            if out is not None:
                out[lo:hi] = res
        if out is not None and hasattr(out, "flush") and hi > lo:
            out.flush()   # a memory-mapped result: the block is in the page cache of the file every rank maps
        dist.barrier(group)
        return res if out is None else out
    if compute is None:
        block = pocs_block_on_device(np.asarray(cube[lo:hi]), mask, device=local_dev, **params)
        if not on_gpu:
            block = block.cpu()
    else:
        block = compute(np.asarray(cube[lo:hi]), mask, **params)
    complex_np = None
    if not torch.is_tensor(block):
        block = np.ascontiguousarray(block)
        complex_np = block.dtype if np.iscomplexobj(block) else None
        block = torch.from_numpy(block.view(np.float32 if block.dtype == np.complex64 else np.float64) if complex_np is not None else block)
        if on_gpu:
            block = block.to(torch.device("cuda", local_dev))
    elif block.is_complex():
        complex_np = np.complex64 if block.dtype == torch.complex64 else np.complex128
        block = torch.view_as_real(block).flatten(-2)      # (n, nil, 2 nxl) reals
    full = gather_blocks(block, cube.shape[0], group) if gather == "all" else gather_blocks_to_root(block, cube.shape[0], group)
    if gather == "root" and rank != 0:
        return None
    full = _download(full)
    return full.view(complex_np) if complex_np is not None else full
