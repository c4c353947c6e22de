"""The N > 1 path on CPU: slice sharding + one all_gather over gloo, world_size 2 (and 3, uneven)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pseudo_3d_interpolation_amd.sharding import block_sizes, slice_block


def test_blocks_tile_the_slice_axis():
    for n in (1, 2, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = slice_block(n, world, r)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))
            assert sum(block_sizes(n, world)) == n
            assert max(block_sizes(n, world)) - min(block_sizes(n, world)) <= 1
    assert slice_block(512, 8, 3) == (192, 256)
    with pytest.raises(ValueError):
        slice_block(8, 2, 2)


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["P3D_ROOT"])
import numpy as np
import torch.distributed as dist
from pseudo_3d_interpolation_amd.sharding import pocs_cube_sharded, slice_block
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(os.environ["P3D_NSLICES"])
rng = np.random.default_rng(0)
cube = (rng.standard_normal((n, 8, 4)) + 1j * rng.standard_normal((n, 8, 4))).astype(np.complex64)
mask = (rng.random((8, 4)) > 0.5).astype(np.uint8)
seen = []
def fake_pocs(block, m, **kw):          # stands in for the GPU call: the test is about the plumbing
    lo, hi = slice_block(n, world, rank)
    assert block.shape[0] == hi - lo and kw["niter"] == 3
    seen.append((lo, hi))
    return block * (1 - m) * 2 + block * m
want = cube * (1 - mask) * 2 + cube * mask
full = pocs_cube_sharded(cube, mask, compute=fake_pocs, gather="all", niter=3)          # the cube on every rank
assert full.dtype == cube.dtype and full.shape == cube.shape
assert np.array_equal(full, want), rank
root = pocs_cube_sharded(cube, mask, compute=fake_pocs, niter=3)                        # default: the trivial gather to rank 0
assert (root is None) == (rank != 0)
if rank == 0:
    assert root.dtype == cube.dtype and np.array_equal(root, want)
real = pocs_cube_sharded(cube.real.copy(), mask, compute=lambda b, m, **kw: b + 1, gather="all", niter=3)
assert np.array_equal(real, cube.real + 1)
import torch
from pseudo_3d_interpolation_amd.sharding import gather_blocks, gather_blocks_to_root
lo, hi = slice_block(n, world, rank)
mine = torch.from_numpy(cube.real[lo:hi].copy())
assert torch.equal(gather_blocks(mine, n), torch.from_numpy(cube.real.copy()))
got = gather_blocks_to_root(mine, n)
assert got.shape[0] == (n if rank == 0 else 0) and (rank != 0 or torch.equal(got, torch.from_numpy(cube.real.copy())))
cplx = torch.from_numpy(cube[lo:hi].copy())                                             # complex blocks go as (re, im) pairs
assert torch.equal(gather_blocks(cplx, n), torch.from_numpy(cube))
got = gather_blocks_to_root(cplx, n)
assert got.dtype == torch.complex64 and (rank != 0 or torch.equal(got, torch.from_numpy(cube)))
as_tensor = pocs_cube_sharded(cube, mask, compute=lambda b, m, **kw: torch.from_numpy(np.ascontiguousarray(b * 3)), gather="all", niter=3)
assert as_tensor.dtype == cube.dtype and np.array_equal(as_tensor, cube * 3)            # a compute that hands back a (complex) tensor
# gather='none': every rank writes its own block into a result array all ranks map (here a .npy file; the step-13 driver's merged cube) -- no
# collective, the result is complete on return (barrier inside)
path = os.path.join(os.environ["P3D_TMP"], "out.npy")
if rank == 0:
    np.lib.format.open_memmap(path, mode="w+", dtype=cube.dtype, shape=cube.shape).flush()
dist.barrier()
shared = np.load(path, mmap_mode="r+")
ret = pocs_cube_sharded(cube, mask, compute=fake_pocs, gather="none", out=shared, niter=3)
assert ret is shared and np.array_equal(np.load(path, mmap_mode="r"), want), rank
own = pocs_cube_sharded(cube, mask, compute=fake_pocs, gather="none", niter=3)          # without out=: the rank's own block
assert own.shape[0] == hi - lo and np.array_equal(own, want[lo:hi])
for bad in (dict(gather="root", out=shared), dict(gather="none", out=shared[:-1]), dict(gather="sideways")):
    try:
        pocs_cube_sharded(cube, mask, compute=fake_pocs, niter=3, **bad)
    except ValueError:
        pass
    else:
        raise AssertionError(f"no ValueError for {list(bad)}")
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", seen)
'''


@pytest.mark.parametrize("world,nslices", [(2, 6), (2, 5), (3, 7)])
def test_sharded_run_and_gather_gloo(tmp_path, world, nslices):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, P3D_ROOT=ROOT, P3D_NSLICES=str(nslices), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P3D_TMP=str(tmp_path))
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.count("ok") == world
