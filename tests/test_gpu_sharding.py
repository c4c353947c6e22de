"""sharding.pocs_cube_sharded with its real compute (the HIP path, result left on the device) -- two and three ranks on the one GPU
of the box, collectives over gloo (a second GPU is not available to the tests; over RCCL the same code gathers the device tensors
directly).  The gathered cube must equal the single-process pocs_cube bit for bit: slices are independent."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["P3D_ROOT"])
import numpy as np
import torch.distributed as dist
from oracle import pocs_oracle as orc
from pseudo_3d_interpolation_amd.functions import POCS as P
from pseudo_3d_interpolation_amd.functions import shearlets
from pseudo_3d_interpolation_amd.sharding import pocs_block_on_device, pocs_cube_sharded
os.environ["LOCAL_RANK"] = "0"          # every rank of this test shares the one GPU
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(os.environ["P3D_NSLICES"])
for real, shape, prm in ((False, (64, 128), dict(niter=9, thresh_op="hard", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2)),
                         (True, (128, 128), dict(niter=7, thresh_op="hard", thresh_model="linear", eps=1e-7, p_max=0.9, p_min=1e-2)),
                         (False, (60, 50), dict(niter=5, thresh_op="soft", thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-2, version="adaptive", alpha=0.8)),
                         (True, (64, 64), dict(niter=4, thresh_op="soft", thresh_model="linear", eps=0.0, p_max=0.9, p_min=0.1, transform_kind="WAVELET", wavelet="db2")),
                         (False, (64, 64), dict(niter=3, thresh_op="hard", thresh_model="exponential", eps=0.0, p_max=0.9, p_min=0.1, transform_kind="WAVELET", wavelet="db4")),
                         (True, (64, 64), dict(niter=4, thresh_op="soft", thresh_model="linear", eps=0.0, p_max=0.9, p_min=0.1, transform_kind="SHEARLET",
                                               auxiliary_data=shearlets.scalesShearsAndSpectra((64, 64))))):
    _, mask, obs = orc.synthetic_cube(shape[0], shape[1], n, 0.6, real=real)
    obs[n // 2] = 0                                          # an all-zero slice passes through untouched
    want = P.pocs_cube(obs, mask, **prm)
    everywhere = pocs_cube_sharded(obs, mask, gather="all", **prm)
    assert everywhere.dtype == obs.dtype and np.array_equal(everywhere, want), (rank, shape)
    root = pocs_cube_sharded(obs, mask, **prm)
    assert (root is None) == (rank != 0) and (rank != 0 or np.array_equal(root, want))
    # no collective at all: every rank moves its own block through its own PCIe link, straight into a result file all ranks map
    path = os.path.join(os.environ["P3D_TMP"], f"out_{shape[0]}_{shape[1]}_{int(real)}_{prm.get('transform_kind', 'FFT')}.npy")
    if rank == 0:
        np.lib.format.open_memmap(path, mode="w+", dtype=obs.dtype, shape=obs.shape).flush()
    dist.barrier()
    shared = np.load(path, mmap_mode="r+")
    assert pocs_cube_sharded(obs, mask, gather="none", out=shared, **prm) is shared
    assert np.array_equal(np.load(path, mmap_mode="r"), want), (rank, shape, "gather='none'")
dev = pocs_block_on_device(obs[:2], mask, device=0, **prm)      # (the SHEARLET case: the device-resident branch of configs[4])
assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), want[:2])
# the argument checks of pocs_cube come before any upload (a short mask would be read out of bounds by the kernels)
for bad in (dict(mask=mask[:-1]), dict(niter=0), dict(version="nope"), dict(thresh_op="nope")):
    kw = dict(prm, **{k: v for k, v in bad.items() if k != "mask"})
    try:
        pocs_block_on_device(obs[:2], bad.get("mask", mask), device=0, **kw)
    except ValueError:
        pass
    else:
        raise AssertionError(f"no ValueError for {list(bad)}")
# per-slice records and caller-chosen chunks are honoured (through pocs_cube), and device= may be passed explicitly
rows = []
dev = pocs_block_on_device(obs[:3], mask, device=0, results=rows, batch_slices=2, **prm)
assert len(rows) == 3 and np.array_equal(dev.cpu().numpy(), want[:3])
again = pocs_cube_sharded(obs, mask, gather="all", device=0, **prm)
assert np.array_equal(again, want)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world,nslices", [(2, 6), (3, 7)])
def test_sharded_hip_path_equals_the_single_process_result(tmp_path, world, nslices):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, P3D_ROOT=ROOT, P3D_NSLICES=str(nslices), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P3D_TMP=str(tmp_path))
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.count("ok") == world
