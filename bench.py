#!/usr/bin/env python3
"""
bench.py -- POCS iterations/s (BASELINE.json metric) on N MI355X GPUs.

    python bench.py [--gpus 1] [--steps K] [--warmup W] [--config {0,1,2,3,4}] [--repeats R] [--density M] [--only-main]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one POCS iteration of the whole cube: every (iline, xline) slice goes once through forward transform ->
threshold -> inverse transform -> re-insertion of the observed traces -> cost sum.  The timed region is one complete job of K
iterations on a cube already resident in HBM: statistics of transform(x_obs), threshold schedule (host, a few scalars per
slice), the K iterations, final store of the result.  The job is repeated R (>= 3) times; `value` is K / the MEDIAN job time
(max over ranks per repeat), min and max are printed beside it.  For N > 1 the slice axis is cut into N contiguous blocks,
one rank per GPU; there is no collective inside the timed region -- the blocks are gathered afterwards with one RCCL collective
on device tensors (all_gather, and gather-to-root beside it; both timed, "gather_ms").  Rank 0 prints ONE JSON line.

--config selects the BASELINE.json configuration of the line's `value` (default 2 = the metric's cube):
    0  64 x 64 x 128 complex64, 50 % missing, FFT, hard, 20 iterations
    1  512 x 512 x 256 complex64, 70 % missing, FFT, hard, exponential decay, 50 iterations
    2  1024 x 1024 x 512 complex64, 80 % missing, FFT, hard, exponential decay, 100 iterations
    3  512 x 512 x 256 float32, 70 % missing, WAVELET db4 ('smooth'), soft, 50 iterations
    4  2048 x 1024 x 1024 float32, 80 % missing, SHEARLET (125 shearlets), hard, 100 iterations.  One GPU holds 2 GiB of
       coefficients per slice, so the cube is worked through in batches of 8 slices; the line times a SAMPLE of the cube
       (--nslices, default 8 slices PER RANK, each rank's sample taken from its own block of the 1024 slices) over the FULL
       100-iteration schedule, and scales the rate by sample / cube.  `--nslices 1024` runs the whole cube.
--steps overrides the iteration count of the line's configuration (the driver runs --steps 20).

The default line (config 2) also carries `other_configs`: configs[1], [3] and [4] (N > 1: configs[4] only -- it is the other
configuration BASELINE.json defines on 8 GPUs) run in the same process at their OWN iteration counts, each with value,
steady-state rate and a `roofline` object whose launch_ms and algorithmic_bytes_per_launch let the fraction be recomputed; and
`end_to_end`: the same job through the host-buffer entry point (NumPy cube in, NumPy cube out: PCIe both ways, chunk pipeline).
--only-main leaves both out (profiling runs: tools/profile.sh).

`roofline.traffic` is read from profiles/traffic_config<i>.json (PMC passes over this file, tools/profile.sh) and is reported
only when that file was measured on the kernel sources of this tree (`kernel_source_hash`); otherwise it is null and the stale
measurement is named under `traffic_last_measured`.

At N = 1 the process is torch-free: device memory comes from the library's own p3d_dev_* entry points, inputs are generated on the
host with NumPy (SURVEY 8d's generator for every slice: plane waves as a rank-6 outer product + its own noise) and uploaded before the clock starts, and the library
runs on the HIP runtime it was built against (`hip_runtime` in the line).  At N > 1 torch is imported FIRST (it must initialise its own
copy of the HIP runtime before ours does, _ffi._preload_torch_hip) for torch.distributed over RCCL; the result blocks are torch tensors
there so that the gather collective can take them.  No torch kernel runs in either case, so `rocprofv3 --pmc ... -- python3 bench.py`
sees the library's dispatches only.
"""
import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
SHEARLET_BATCH = 8        # slices per p3d_shearlet_run call (8 x 125 work slices of 16 MiB = 16 GiB)

CONFIGS = {
    0: dict(kind="FFT", nil=64, nxl=64, nslices=128, missing=0.5, steps=20, op="hard", real=False),
    1: dict(kind="FFT", nil=512, nxl=512, nslices=256, missing=0.7, steps=50, op="hard", real=False),
    2: dict(kind="FFT", nil=1024, nxl=1024, nslices=512, missing=0.8, steps=100, op="hard", real=False),
    3: dict(kind="WAVELET", nil=512, nxl=512, nslices=256, missing=0.7, steps=50, op="soft", real=True, wavelet="db4"),
    4: dict(kind="SHEARLET", nil=2048, nxl=1024, nslices=None, sample_per_rank=SHEARLET_BATCH, cube_slices=1024, missing=0.8, steps=100,
            op="hard", real=True),
}
# ALGORITHMIC bytes per point and iteration:
#   FFT, complex64: iterate read 8 + written 8 + observed data 8 + float32 weight 4                                        = 28
#   WAVELET, float32: (4 read + 4 written) per transform x 2 transforms x 4/3 (coarser levels) + observed 4 + weight 4    = 29.33
#   SHEARLET: per shearlet 12 (spectrum x Psi -> coefficients) + 16 (threshold in place) + 12 (coefficients -> sum)        = 40 nsh
#             -- of the rows a shearlet's spectrum does not vanish on (a Parseval frame covers every frequency about twice: ~41 % of
#             the (shearlet, row) pairs at 2048 x 1024 x 125; the share is read from the plan), and for float32 cubes on symmetric
#             spectra of the Hermitian half of them (rows 0 ... nil/2): the zeros and the conjugates need not move.  The dense figure
#             (40 nsh, what round 2 priced) is printed beside it as `dense_accounting`.
ALG_BYTES = {"FFT": lambda nsh: 28.0, "WAVELET": lambda nsh: 8.0 * 2 * 4 / 3 + 8.0, "SHEARLET": lambda nsh: 40.0 * nsh}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[i] (default: the metric's cube)")
    ap.add_argument("--steps", type=int, default=None, help="POCS iterations in the timed job (default: the configuration's)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of the job (0: as many as fill ~1.5 s, between 3 and 15)")
    ap.add_argument("--nil", type=int, default=None)
    ap.add_argument("--nxl", type=int, default=None)
    ap.add_argument("--nslices", type=int, default=None, help="slices worked on by all GPUs together (config 4: the sample of the cube)")
    ap.add_argument("--missing", type=float, default=None)
    ap.add_argument("--thresh-op", default=None)
    ap.add_argument("--alpha", type=float, default=1.0, help="re-insertion weight (the metric's setting: 1)")
    ap.add_argument("--p-min", default="1e-3", help="final threshold factor or 'adaptive' (the metric's setting: 1e-3)")
    ap.add_argument("--eps", type=float, default=0.0, help="cost threshold of the early exit (0 = run all iterations, the metric's setting)")
    ap.add_argument("--density", type=int, default=0,
                    help="0: the survey's recipe (6 plane waves + 1 %% noise: a very sparse spectrum); M > 0: M random spectral "
                         "coefficients per slice + 1 %% noise -- denser spectra keep more column blocks (FFT configurations)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the side runs (sparse shortcut off, real cube, density curve)")
    ap.add_argument("--only-main", action="store_true", help="no other_configs, no end_to_end (profiling runs)")
    ap.add_argument("--no-end-to-end", action="store_true")
    return ap.parse_args()


def kernel_source_hash():
    """sha256[:16] over the kernel sources of this tree (csrc/*.hip, *.hpp, Makefile): stamps PMC traffic measurements."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pseudo-3d-interpolation_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + [os.path.join(csrc, "Makefile")]):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_valu(config, kernel_substring):
    """The VALU-issue figures of one kernel from profiles/traffic_config<i>.json (tools/traffic.py: SQ_INSTS_VALU per dispatch over the kernel's
    average duration, as a fraction of the chip's vector issue rate) -- only while the file was measured on the kernel sources of this tree."""
    path = os.path.join(ROOT, "profiles", f"traffic_config{config}.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    for name, k in rec.get("kernels", {}).items():
        if kernel_substring in name and "valu_issue_frac" in k:
            return {"kernel": name, "valu_wave_instructions_per_dispatch": k["valu_wave_instructions_per_dispatch"], "average_ns": k["average_ns"],
                    "valu_frac": k["valu_issue_frac"], "peak_wave_instructions_per_s": rec.get("valu_issue_peak_wave_instructions_per_s"),
                    "from": f"profiles/traffic_config{config}.json: rocprofv3 --pmc SQ_INSTS_VALU per dispatch / the kernel's average duration in the kernel trace of "
                            f"the same command / (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wavefront instruction)"}
    return None


def measured_traffic(config, workload_key, scale=1.0):
    """(traffic, traffic_from, last_measured) from profiles/traffic_config<i>.json.  A measurement counts only for the kernel sources it
    was taken on; a stale one is named, not reported."""
    path = os.path.join(ROOT, "profiles", f"traffic_config{config}.json")
    if not os.path.isfile(path):
        return None, None, None
    try:
        with open(path) as fh:
            rec = json.load(fh)
        per = rec["hbm_bytes_per_iteration"]
    except (OSError, ValueError, KeyError):
        return None, None, None
    info = {"file": f"profiles/traffic_config{config}.json", "hbm_bytes_per_iteration": per, "workload": rec.get("workload"),
            "round": rec.get("round"), "kernel_source_hash": rec.get("kernel_source_hash"), "iterations": rec.get("iterations")}
    if rec.get("workload_key") != workload_key:
        return None, None, dict(info, why="measured on another workload")
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, None, dict(info, why="kernel sources changed since the measurement")
    return per * scale, (f"profiles/traffic_config{config}.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --config {config} "
                         f"--only-main, round {rec.get('round')}, K = {rec.get('iterations')}; (FETCH_SIZE x 2 [gfx950] + WRITE_SIZE) x 1024 B per "
                         f"iteration of the steady-state kernels"), None


# ---------------------------------------------------------------------------------------------------------------------------------
# synthetic inputs, generated on the host (NumPy) and uploaded; device memory without torch at N = 1
# ---------------------------------------------------------------------------------------------------------------------------------
NOISE_POOL = 16
_GEN_THREADS = max(1, min(16, os.cpu_count() or 1))
_gen_pool = None


def _threads():
    global _gen_pool
    if _gen_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _gen_pool = ThreadPoolExecutor(_GEN_THREADS)
    return _gen_pool


def noise_pool(nil, nxl):
    """NOISE_POOL complex Gaussian slices for the random-spectrum cubes of the density curve (not SURVEY 8d's recipe anyway): slice s uses
    pool[s % P] rolled by a slice-dependent shift."""
    rng = np.random.default_rng(20240917)
    return (rng.standard_normal((NOISE_POOL, nil, nxl), dtype=np.float32)
            + 1j * rng.standard_normal((NOISE_POOL, nil, nxl), dtype=np.float32)).astype(np.complex64)


def _finish_slice(x, s, pool, mask):
    """x += 0.01 * pool[s % P] rolled by a shift that depends on s // P; x *= mask (trace decimation)."""
    r, j = s // NOISE_POOL, s % NOISE_POOL
    x += np.float32(0.01) * np.roll(pool[j], shift=(3 * r + 1, 5 * r + 2), axis=(0, 1))
    x *= mask
    return x


def plane_wave_slices(nil, nxl, first, count, pool, mask):
    """SURVEY 8d's generator, slice by slice (the recipe and the random stream of oracle.synthetic_slice: rng = default_rng(1234 + s); six plane
    waves with integer wavenumbers and complex normal amplitudes; 1 % complex Gaussian noise drawn from the same stream), times the trace mask.
    Re-stated here (the oracle is not imported for data generation): a plane wave is an outer product exp(2 pi i k1 il / nil) x
    exp(2 pi i k2 xl / nxl), so a slice is one (nil x 6) @ (6 x nxl) product in double precision + its noise, cast to complex64 -- the oracle's
    values up to the order of a six-term sum.  A few threads share the slices (NumPy's generators and BLAS release the GIL)."""
    out = np.empty((count, nil, nxl), np.complex64)
    il = np.arange(nil, dtype=np.float64) / nil
    xl = np.arange(nxl, dtype=np.float64) / nxl
    maskc = mask.astype(np.float32)

    def one(i):
        rng = np.random.default_rng(1234 + first + i)
        k1 = np.empty(6); k2 = np.empty(6); amp = np.empty(6, np.complex128)
        for e in range(6):   # the draw order of oracle.synthetic_slice
            k1[e] = int(rng.integers(-(nil // 8), max(nil // 8, 1)))
            k2[e] = int(rng.integers(-(nxl // 8), max(nxl // 8, 1)))
            amp[e] = complex(rng.standard_normal(), rng.standard_normal())
        acc = (np.exp(2j * np.pi * np.outer(il, k1)) * amp) @ np.exp(2j * np.pi * np.outer(k2, xl))
        nre = rng.standard_normal((nil, nxl))
        acc += 0.01 * (nre + 1j * rng.standard_normal((nil, nxl)))
        out[i] = acc
        out[i] *= maskc
    list(_threads().map(one, range(count)))
    return out


def random_spectrum_slices(dev, plan, nil, nxl, first, pool, mask, m, chunk=32):
    """M random spectral coefficients per slice (positions uniform over the spectrum, complex normal amplitudes, seeded per slice), brought to the
    space domain by the library's own inverse FFT, + 1 % noise, times the mask -- chunk by chunk through the device array `dev` (complex64)."""
    n = dev.shape[0]
    maskc = mask.astype(np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        spec = np.zeros((hi - lo, nil * nxl), np.complex64)
        for i in range(lo, hi):
            rng = np.random.default_rng(4321 + first + i)
            pos = rng.choice(nil * nxl, size=m, replace=False)
            spec[i - lo, pos] = ((rng.standard_normal(m) + 1j * rng.standard_normal(m)) * (nil * nxl / math.sqrt(m))).astype(np.complex64)
        dev.upload(spec.reshape(hi - lo, nil, nxl), first=lo)
        per = nil * nxl * 8
        plan.fft2_dev(dev.ptr + lo * per, dev.ptr + lo * per, hi - lo, inverse=True)
        x = dev.download(lo, hi - lo)
        list(_threads().map(lambda i: _finish_slice(x[i - lo], first + i, pool, maskc), range(lo, hi)))
        dev.upload(x, first=lo)


class TorchArray:
    """The interface of _ffi.DeviceArray over a torch tensor (N > 1: the gather collective takes the tensor)."""

    def __init__(self, torch, shape, dtype, device):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.tensor = torch.empty(self.shape, dtype=getattr(torch, self.dtype.name), device=device)
        self.ptr, self.nbytes, self._torch = self.tensor.data_ptr(), self.tensor.numel() * self.tensor.element_size(), torch

    def upload(self, host, first=0):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        self.tensor[first:first + host.shape[0]].copy_(self._torch.from_numpy(host))
        return self

    def download(self, first=0, count=None, out=None):
        count = self.shape[0] - first if count is None else count
        res = self.tensor[first:first + count].cpu().numpy()
        if out is not None:
            out[...] = res
            return out
        return res

    def free(self):
        self.tensor = None
        self.ptr = None
        self._torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on a bounded sample of the same cube, one single-threaded NumPy process per core (the shape of the
# reference's LocalCluster(processes=True, threads_per_worker=1)); the shearlet slice is too big for that and uses all cores on ONE
# slice through scipy.fft's threads instead
# ---------------------------------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    kind, x, mask, niter, op, extra = job[:6]
    keep = len(job) > 6 and job[6]
    t0 = time.perf_counter()
    res = None
    if kind == "FFT":
        from oracle import pocs_oracle as orc
        res = orc.pocs_slice(x, mask, niter=niter, thresh_op=op, thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
    elif kind == "WAVELET":
        from oracle import wavelet_oracle as wo
        res = wo.pocs_slice_wavelet(x, mask, wavelet=extra, niter=niter, thresh_op=op, thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-3)
    elif kind == "WAVELET32":
        # the WAVELET loop CARRIED in float32 -- what PyWavelets executes for a float32 cube (its C kernels are typed), i.e. the
        # reference's own arithmetic for configs[3]; the schedule from the double-precision decomposition, as the oracle's
        # (tests/test_gpu_wavelet.py::test_wavelet_long_run_tracks_float32_reference has the same loop)
        from oracle import pocs_oracle as orc
        from oracle import wavelet_oracle as wo
        n0, n1 = x.shape
        bank32 = tuple(b.astype(np.float32) for b in wo.filter_bank(extra))
        tau = wo.wavelet_schedule("exponential", niter, 0.99, 1e-3, wo.wavedec2(x.astype(np.float64), wo.filter_bank(extra))[1:])
        x32 = x.astype(np.float32)
        res = x32
        for k in range(niter):
            c = wo.wavedec2(res, bank32)
            shr = [tuple(orc.apply_threshold(c[l + 1][d], np.float32(tau[k, l, d]), kind=op) for d in range(3)) for l in range(len(c) - 1)]
            res = (wo.waverec2([c[0]] + shr, bank32)[:n0, :n1] * (1 - mask) + x32).astype(np.float32)
    wall = time.perf_counter() - t0
    return (wall, res) if keep else wall


def cpu_parity(pool, kind, obs_slices, mask, op, niter, gpu_result, extra=None, also=None):
    """rel-L2 per slice between the GPU result of the TIMED job (the first slices of its output cube) and the oracle run on the
    same observed slices for the same number of iterations -- fed in double precision (BASELINE's yardstick), and once more in the
    cube's own precision: NumPy's float32-vs-float64 spread on these very slices, which is what the reference itself gives up."""
    wide = np.complex128 if np.iscomplexobj(obs_slices[0]) else np.float64
    narrow = np.complex64 if np.iscomplexobj(obs_slices[0]) else np.float32
    # (the GPU was fed the float32 / complex64 casts of these slices: the oracle gets exactly those values, widened or not)
    ref64 = [r for _, r in pool.map(_cpu_worker, [(kind, s.astype(narrow).astype(wide), mask, niter, op, extra, True) for s in obs_slices])]
    kind32 = "WAVELET32" if (kind == "WAVELET" and narrow is np.float32) else kind   # (the wavelet oracle promotes to double whatever it is fed)
    ref32 = [r for _, r in pool.map(_cpu_worker, [(kind32, s.astype(narrow), mask, niter, op, extra, True) for s in obs_slices])]
    rel = [float(np.linalg.norm(g - r) / np.linalg.norm(r)) for g, r in zip(gpu_result, ref64)]
    spread = [float(np.linalg.norm(a.astype(wide) - r) / np.linalg.norm(r)) for a, r in zip(ref32, ref64)]
    other = {}
    for name, res in (also or {}).items():   # the same slices through another device path (the loop in double precision)
        r2 = [float(np.linalg.norm(g - r) / np.linalg.norm(r)) for g, r in zip(res, ref64)]
        other[name] = {"rel_l2_max": max(r2), "rel_l2_median": float(np.median(r2)), "slices": len(r2), "niter": int(niter)}
    return {
        "other_paths": other or None,
        "rel_l2_max": max(rel), "rel_l2_median": float(np.median(rel)), "slices": len(rel), "niter": int(niter),
        "against": "oracle (NumPy restatement of the reference, pinned on its golden vectors) fed the same observed slices in double precision",
        "what": "out[:slices] of the TIMED job (same cube, same K, same schedule) vs the oracle, ||gpu - ref|| / ||ref|| per slice",
        "reference_own_float32_spread_max": max(spread), "reference_own_float32_spread_median": float(np.median(spread)),
        "spread_note": "the oracle fed the cube's own precision (what NumPy executes for such a cube) vs the oracle fed double precision, same slices",
        "tolerance": 1e-5,
        "note": None if kind != "WAVELET" else
        "the WAVELET iteration with 'smooth' boundaries is expansive on decimated data (the iterate grows by orders of magnitude over the schedule, "
        "DESIGN.md section 4): rounding noise grows with it, in the reference's own float32 run as in the device's -- compare rel_l2 with "
        "reference_own_float32_spread, not with the tolerance; short / well-conditioned runs are held to 1e-5 by tests/test_gpu_wavelet.py",
    }


def cpu_baseline(kind, obs_slices, mask, op, budget_s, nslices_cube, extra=None, parity_of=None, parity_niter=None, also=None):
    """(cpu_baseline record, parity record or None).  parity_of: the GPU result on exactly these slices after parity_niter iterations."""
    import multiprocessing as mp

    workers = len(obs_slices)
    pts = obs_slices[0].size
    per_it = {"FFT": 93.3e-3, "WAVELET": 0.35}[kind] * pts / (1024 * 1024)   # s per slice-iteration on one core (measured)
    niter = int(max(4, min(100, budget_s / max(per_it, 1e-6))))
    ctx = mp.get_context("spawn")
    parity = None
    with ctx.Pool(workers) as pool:
        tiny = (kind, obs_slices[0][:16, :16].copy(), mask[:16, :16].copy(), 2, op, extra)
        pool.map(_cpu_worker, [tiny] * workers)  # spin up
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(kind, s, mask, niter, op, extra) for s in obs_slices])
        wall = time.perf_counter() - t0
        if parity_of is not None:
            parity = cpu_parity(pool, kind, obs_slices, mask, op, parity_niter, parity_of, extra, also)
    slice_iters_per_s = workers * niter / wall
    return parity, {
        "value": slice_iters_per_s / nslices_cube,
        "unit": "iterations/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} slices x {niter} iterations of the same cube, one NumPy process per core, scaled by 1/{nslices_cube} slices",
        "slice_iterations_per_s": slice_iters_per_s,
    }


def cpu_baseline_shearlet(x, mask, psi, op, nslices_cube, niter=2):
    """(cpu_baseline record, the oracle's result on this slice after `niter` iterations)."""
    from oracle import shearlet_oracle as so
    cores = min(os.cpu_count() or 1, 16)   # the GPU box's CPU share for one GPU
    so.pocs_slice_shearlet_real(x[:64, :32].astype(np.float64), mask[:64, :32], np.ones((64, 32, 1)), niter=2, thresh_op=op, p_min=1e-3)  # imports, threads
    t0 = time.perf_counter()
    res = so.pocs_slice_shearlet_real(x.astype(np.float64), mask, psi, niter=niter, thresh_op=op, thresh_model="exponential", p_max=0.99, p_min=1e-3,
                                      workers=cores)
    wall = time.perf_counter() - t0
    sips = (niter + 1) / wall   # the schedule's transform + `niter` iterations: niter + 1 forward transforms, niter inverse ones ~ niter + 1/2 iterations
    return {
        "value": sips / nslices_cube, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": f"1 slice x {niter} iterations (+ the schedule's transform) of the same cube, scipy.fft real transforms on {cores} threads, "
                  f"scaled by 1/{nslices_cube} slices",
        "slice_iterations_per_s": sips,
    }, res


class Ctx:
    """What every leg shares: this rank's device, device memory, fences -- and, for N > 1 only, torch and the process group."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
            raise SystemExit(f"WORLD_SIZE={self.world} does not match --gpus {args.gpus}")
        self.torch = self.dist = None
        self.rehearsal = False
        self.dev_index = local_rank
        if self.world > 1:
            import torch   # BEFORE the library is loaded: torch's copy of the HIP runtime has to initialise first (_ffi._preload_torch_hip)
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            # P3D_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo -- exercises the multi-rank code path where no second GPU exists
            # (the numbers mean nothing then).  The real thing: one rank per GPU over RCCL.
            self.rehearsal = os.environ.get("P3D_BENCH_REHEARSAL") == "1"
            self.dev_index = local_rank % max(torch.cuda.device_count(), 1) if self.rehearsal else local_rank
            torch.cuda.set_device(self.dev_index)
            self.device = torch.device("cuda", self.dev_index)
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.device)
        from pseudo_3d_interpolation_amd import _ffi
        self.ffi = _ffi

    def empty(self, shape, dtype):
        """Uninitialised device memory: the library's own allocation at N = 1, a torch tensor at N > 1."""
        if self.torch is None:
            return self.ffi.DeviceArray(shape, dtype, device=self.dev_index)
        return TorchArray(self.torch, shape, dtype, self.device)

    def sync(self):
        self.ffi.device_synchronize(self.dev_index)

    def fence(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
            self.sync()

    def max_over_ranks(self, sec):
        if self.world > 1:
            t = self.torch.tensor([sec], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            sec = float(t.item())
        return sec


def run_leg(ctx, config, K_override, main, override=None):
    """One BASELINE configuration: generate its cube in HBM (this rank's block), time R jobs of K iterations, profile, side runs.
    Returns the record of the leg (rank 0: complete; other ranks: the fields they computed)."""
    args, rank, world = ctx.args, ctx.rank, ctx.world
    _ffi = ctx.ffi
    from pseudo_3d_interpolation_amd.functions import POCS as P
    from pseudo_3d_interpolation_amd.sharding import slice_block

    cfg = dict(CONFIGS[config])
    if main:
        for key, val in (("nil", args.nil), ("nxl", args.nxl), ("nslices", args.nslices), ("missing", args.missing), ("op", args.thresh_op)):
            if val is not None:
                cfg[key] = val
    if override:
        cfg.update(override)   # a side leg on another grid (the 7-smooth extents of a survey grid): no CPU sample, no side runs
    kind, nil, nxl, missing, op = cfg["kind"], cfg["nil"], cfg["nxl"], cfg["missing"], cfg["op"]
    K = K_override if K_override is not None else cfg["steps"]
    W = args.warmup
    dev_index = ctx.dev_index
    p_min = args.p_min if args.p_min == "adaptive" else float(args.p_min)
    density = args.density if (main and kind == "FFT") else 0

    # ---- which slices this rank works on ---------------------------------------------------------
    if kind == "SHEARLET" and cfg.get("cube_slices"):
        # config 4: the cube's slice axis is cut into `world` blocks; all ranks together time a sample of `nslices` slices, every
        # rank its share of the sample taken from the head of ITS block of the cube (--nslices 1024: the whole cube)
        cube_slices = cfg["cube_slices"]
        nslices = min(cube_slices, cfg["nslices"] if cfg["nslices"] is not None else cfg["sample_per_rank"] * world)
        s_lo, s_hi = slice_block(nslices, world, rank)
        n_local = s_hi - s_lo
        lo = slice_block(cube_slices, world, rank)[0]
    else:
        nslices = cube_slices = cfg["nslices"]
        lo, hi = slice_block(nslices, world, rank)
        n_local = hi - lo
    if n_local < 1:
        raise SystemExit(f"rank {rank} of {world} has no slice to work on ({nslices} slices)")
    pts_local = n_local * nil * nxl

    # ---- inputs, resident in HBM before the clock starts -----------------------------------------
    mask = (np.random.default_rng(42).random((nil, nxl)) >= missing).astype(np.uint8)   # shared trace mask
    mask_t = ctx.empty((nil, nxl), np.float32).upload(mask.astype(np.float32))
    pool = noise_pool(nil, nxl)
    fft_plan = _ffi.Plan(nil, nxl, n_local, device=dev_index) if kind == "FFT" else None
    np_dtype = np.float32 if cfg["real"] else np.complex64
    x_obs = ctx.empty((n_local, nil, nxl), np_dtype)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and density == 0 and not override
    cpu_slices = None
    n_cpu = 0
    if want_cpu:   # the CPU baseline is a single-GPU-run extra (rank 0, N = 1)
        from oracle import pocs_oracle as orc  # the cpu_baseline leg (and the slices it is fed) -- nothing else touches the oracle
        n_cpu = max(1, min(os.cpu_count() or 1, 16, n_local)) if kind != "SHEARLET" else 1
        cpu_slices = np.stack([orc.synthetic_slice(nil, nxl, lo + s, real=cfg["real"]) for s in range(n_cpu)]) * mask

    def generate(dens):
        """This rank's observed block into x_obs, chunk by chunk from the host; returns nothing (the host copy is not kept)."""
        if dens > 0:
            random_spectrum_slices(x_obs, fft_plan, nil, nxl, lo, pool, mask, dens)
        else:
            chunk = max(1, (256 << 20) // (nil * nxl * 8))
            for c0 in range(0, n_local, chunk):
                xs = plane_wave_slices(nil, nxl, lo + c0, min(chunk, n_local - c0), pool, mask)
                x_obs.upload(xs.real if cfg["real"] else xs, first=c0)
        if n_cpu:   # the CPU sample sees exactly the slices the GPU processes
            x_obs.upload(cpu_slices.astype(np_dtype), first=0)

    generate(density)
    out = ctx.empty((n_local, nil, nxl), np_dtype)
    ctx.sync()
    DT = _ffi.P3D_F32 if cfg["real"] else _ffi.P3D_C64
    esz = 4 if cfg["real"] else 8

    # ---- the job of each transform kind ------------------------------------------------------------
    nsh = 0
    psi = None
    if kind == "FFT":
        plan = fft_plan

        def job(niter, profile=False):
            # statistics of fft2(x_obs) with the mask at hand: the pass doubles as the first pass of the job (p3d_pocs_prime_dev)
            stats = plan.prime_dev(x_obs.ptr, DT, mask_t.ptr, n_local)
            active = stats[:, 2] > 0
            stats[~active] = 1.0
            tau = P._schedule_from_stats(stats, nil * nxl, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(x_obs.ptr, DT, mask_t.ptr, tau, niter, out.ptr, n_local,
                                thresh_op=op, eps=args.eps, alpha=args.alpha, active=active, profile=profile, want_sums=False, primed=True)
    elif kind == "WAVELET":
        plan = _ffi.WaveletPlan(nil, nxl, n_local, wavelet=cfg["wavelet"], device=dev_index)

        def job(niter, profile=False):
            stats = plan.stats_dev(x_obs.ptr, DT, n_local)
            tau = P._wavelet_schedule_from_stats(stats, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(x_obs.ptr, DT, mask_t.ptr, tau, niter, out.ptr, n_local, thresh_op=op, eps=args.eps,
                                alpha=args.alpha)
    else:
        from pseudo_3d_interpolation_amd.functions import shearlets
        psi = shearlets.scalesShearsAndSpectra((nil, nxl), dtype=np.float32)
        nsh = psi.shape[-1]
        batch = min(n_local, SHEARLET_BATCH)
        plan = _ffi.ShearletPlan(psi, max_slices=batch, device=dev_index)
        per = nil * nxl * esz

        def job(niter, profile=False):
            # the rank's slices in batches of <= 8: statistics, schedule and the niter iterations of one batch, then the next
            done_all, ms_all = [], 0.0
            for b0 in range(0, n_local, batch):
                nb = min(batch, n_local - b0)
                xp, op_ = x_obs.ptr + b0 * per, out.ptr + b0 * per
                stats = plan.stats_dev(xp, DT, nb)
                tau = P._shearlet_schedule_from_stats(stats, (nil, nxl), "exponential", niter, 0.99, p_min, "values")
                done, _, ms = plan.run_dev(xp, DT, mask_t.ptr, tau, niter, op_, nb, thresh_op=op, eps=args.eps, alpha=args.alpha)
                done_all.append(done)
                ms_all += ms
            return np.concatenate(done_all), None, ms_all

    def timed(niter):
        """One job of `niter` iterations between two fences; seconds = max over ranks."""
        ctx.fence()
        t0 = time.perf_counter()
        res = job(niter)
        ctx.fence()
        return ctx.max_over_ranks(time.perf_counter() - t0), res

    if W > 0:
        job(min(W, K))
    first_s, (done, _, dev_ms) = timed(K)
    assert args.eps > 0 or (int(done.min()) == K and int(done.max()) == K)
    floor = 5 if main else 3
    R = args.repeats if (args.repeats > 0 and main) else int(max(floor, min(15, math.ceil(1.5 / max(first_s, 1e-4)))))
    if kind == "SHEARLET" and not (args.repeats > 0 and main):
        R = 3 if first_s < 4.0 else 2
    times, dev_times = [first_s], [dev_ms]
    for _ in range(R - 1):
        sec, (_, _, dms) = timed(K)
        times.append(sec)
        dev_times.append(dms)
    seconds = float(np.median(times))
    # the timed job's result on the slices the CPU leg also computes (parity figure of the line; nothing below may overwrite it)
    gpu_first = out.download(0, n_cpu) if (cpu_slices is not None and kind != "SHEARLET" and args.eps == 0) else None

    nz_fraction = plan.last_sparsity() if kind == "FFT" else -1.0

    # ---- strong scaling without the hardware: the SAME job on 1/8 of this rank's slices (the block a GPU gets at N = 8) -------
    block8 = None
    if rank == 0 and world == 1 and main and kind == "FFT" and n_local % 8 == 0 and n_local >= 8 and not args.only_main:   # (not under a profiler: launches of ONE size there)
        nl_full = n_local
        n_local = nl_full // 8          # (job() reads n_local; the plan was created for nl_full slices)
        try:
            job(min(W, 3) or 1)
            bt = []
            for _ in range(5):
                ctx.sync()
                b0 = time.perf_counter()
                job(K)
                ctx.sync()
                bt.append(time.perf_counter() - b0)
        finally:
            n_local = nl_full
        b_s = float(np.median(bt))
        block8 = {"slices": nl_full // 8, "job_s_median": b_s, "predicted_speedup_8": seconds / b_s,
                  "note": "the whole job (statistics, schedule round trip, K iterations, last pass) on the first 1/8 of the slices, on this one GPU: "
                          "what each of 8 ranks would run; seconds(all slices) / seconds(1/8) is the strong-scaling speed-up the job's own fixed costs "
                          "allow BEFORE launch skew between ranks (no collective inside the timed region).  A prediction, not a measurement: unmeasured on 8 GPUs"}
        job(min(W, 3) or 1)             # leave `out` holding the full result again for the legs below
        job(K)

    # ---- per-kernel durations of the same job, HIP events on the plan's stream --------------------
    alg_bytes = ALG_BYTES[kind](nsh) * pts_local
    shear_share = None
    if kind == "SHEARLET":
        half = (nil // 2 + 1) / nil if plan.paired else 1.0
        shear_share = {"row_group_fraction": plan.row_group_fraction, "hermitian_half": plan.paired, "rows_moved_fraction": plan.row_group_fraction * half}
        dense_bytes = alg_bytes
        alg_bytes = dense_bytes * shear_share["rows_moved_fraction"]

    def profile_fft():
        job(K, profile=True)   # the same job once more with HIP events around every pass (same schedule, same sparsity as the timed one)
        prof = plan.last_profile()
        kept = plan.last_sparsity()
        return prof, (1.0 if kept < 0 else kept)

    roof = None
    steady = None
    if rank == 0 and not args.no_profile:
        default_shape = (nil, nxl) == (CONFIGS[config]["nil"], CONFIGS[config]["nxl"]) and density == 0 and op == CONFIGS[config]["op"]
        if kind == "FFT":
            prof, kept = profile_fft()
            it_ms = prof["colpass_ms"] + prof["rowpass_ms"]
            resident = prof["colpass_launches"] == 0   # small slices: the whole job is ONE kernel (p3d_resident.hip), no per-pass events
            if resident:
                it_ms = float(np.median(dev_times)) / K   # HIP events of the library around the kernel, on the plan's stream
            achieved = alg_bytes / (it_ms * 1e-3) / 1e9 if it_ms > 0 else 0.0
            traffic, traffic_from, last = measured_traffic(config, f"{nil}x{nxl}x{n_local}") if default_shape else (None, None, None)
            steady = 1e3 / it_ms if it_ms > 0 else None
            roof = {
                "bound": "hbm",
                "kernel": (f"resident_kernel<{nil},{nxl}>: one workgroup per slice, all {K} iterations in registers / LDS -- the iterations "
                           f"move NO HBM bytes (16 B/point per job), so `achieved` (algorithmic bytes over time) may exceed the HBM peak"
                           if resident else
                           f"mix_col_kernel<{nil}> + mix_row_kernel<{nxl}> (mixed-radix register engine, csrc/p3d_mix.hpp) = one POCS iteration of {n_local} slices"
                           if (nil & (nil - 1)) and (nxl & (nxl - 1)) else
                           f"col_kernel<{nil},COL_ITER> + the persistent row pass (row_pipe32_kernel for rows of 1024 samples, row_pipe64_kernel<{nxl}> "
                           f"for other rows of whole wavefronts, row_pipe_kernel otherwise) = one POCS iteration of {n_local} slices"),
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_from": traffic_from, "traffic_last_measured": last,
                "algorithmic_bytes_per_launch": alg_bytes,
                "launch_ms": it_ms,
                "colpass_ms": prof["colpass_ms"], "rowpass_ms": prof["rowpass_ms"],
                # bytes the two passes move per point with a fraction f of the spectrum's column blocks kept: column pass 8 read +
                # 8 f written; row pass 8 f read + 8 written + 8 (1 - missing) observed samples (compact)
                "colpass_moved_GBps": (8 + 8 * kept) * pts_local / (prof["colpass_ms"] * 1e-3) / 1e9 if prof["colpass_ms"] else 0.0,
                "rowpass_moved_GBps": (8 * kept + 8 + 8 * (1 - missing)) * pts_local / (prof["rowpass_ms"] * 1e-3) / 1e9
                if prof["rowpass_ms"] else 0.0,
            }
        else:
            it_ms = float(np.median(dev_times)) / K   # HIP events of the library around the K-iteration loop(s), on the plan's stream
            achieved = alg_bytes / (it_ms * 1e-3) / 1e9
            steady = 1e3 / it_ms
            # config 4: the traffic file holds bytes per slice-iteration; scale to this rank's slices
            key = f"{nil}x{nxl}" + (f"x{n_local}" if kind == "WAVELET" else f"x{nsh}sh")
            traffic, traffic_from, last = measured_traffic(config, key, scale=(n_local if kind == "SHEARLET" else 1.0)) if default_shape else (None, None, None)
            roof = {
                "bound": "hbm",
                "kernel": ("dwt2_tile_kernel / idwt2_tile_kernel chain (level 1 and 2 one launch per level and direction, the coarser levels "
                           "in one slice-resident kernel)" if kind == "WAVELET" else
                           "row_kernel<ROW_SPREAD_INV> + col_shear_pair_kernel (float32 cubes; col_pipe_kernel<COL_SHRINK> otherwise) + "
                           "row_kernel<ROW_GATHER_FWD> + the two fft2 passes")
                          + f" = one POCS iteration of {n_local} slices",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_from": traffic_from, "traffic_last_measured": last,
                "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": it_ms,
            }
            if shear_share is not None:
                # the column pass of this leg is bound by vector-instruction issue, not by memory (profiles/r04_shearlet_column_ablation.txt): its roofline
                vi = measured_valu(config, "col_shear_pair_kernel") if default_shape else None
                roof["valu"] = vi
                roof["valu_frac"] = None if vi is None else vi["valu_frac"]
                roof["shearlet_rows"] = dict(shear_share, note=(
                    "algorithmic bytes = 40 B x points x shearlets x rows_moved_fraction: rows on which a shearlet's spectrum vanishes carry only zeros "
                    "through the iteration (row_group_fraction = share of the (shearlet, 8-row group) pairs that do not), and a float32 cube on symmetric "
                    "spectra has Hermitian work slices (rows 0 ... nil/2 suffice)"))
                roof["dense_accounting"] = {"algorithmic_bytes_per_launch": dense_bytes, "achieved": dense_bytes / (it_ms * 1e-3) / 1e9,
                                            "note": "40 B x points x shearlets, every row of every shearlet counted (the round-2 figure): not a bound any more -- `achieved` here "
                                                    "may exceed the HBM peak"}
        if kind == "FFT" and default_shape:
            # the second resource the two passes spend: vector-instruction issue (the transforms), from the SQ_INSTS_VALU pass of tools/traffic.sh
            vc, vr = measured_valu(config, "col_kernel<"), measured_valu(config, "row_pipe")
            if vc and vr:
                roof["valu"] = {"colpass": vc, "rowpass": vr,
                                "note": "valu_frac = wave instructions per second over the guide's FP32 vector rate (one wavefront instruction per 2 cycles and SIMD); "
                                        "packed-math and single-wave issue cost 4 cycles, and SQ_ACTIVE_INST_VALU has the vector ALU busy about half of either pass "
                                        "(profiles/r05_sq_counters_1024x1024.txt): the passes are bound by neither resource alone but by how little two wavefronts per "
                                        "SIMD overlap them"}
        if roof["traffic"]:
            roof["moved_GBps"] = roof["traffic"] / (roof["launch_ms"] * 1e-3) / 1e9
            roof["moved_frac"] = roof["moved_GBps"] / HBM_PEAK_GBPS

    # ---- the same job with the sparse-spectrum shortcut switched off (reported beside `value`, rank 0 only) ----
    dense_its = None
    side = rank == 0 and not args.no_dense and kind == "FFT" and main
    if side and nz_fraction >= 0:
        os.environ["P3D_NO_SPARSE"] = "1"
        job(min(W, 3) or 1)
        ctx.sync()
        d0 = time.perf_counter()
        job(K)
        ctx.sync()
        dense_its = K / (time.perf_counter() - d0)   # the ranks run in parallel: rank 0's time for its block is the job's
        del os.environ["P3D_NO_SPARSE"]

    # ---- the real-valued (float32, time-domain) cube of the same shape, reported beside `value` (rank 0, N = 1 only) ----
    real_its = None
    if side and world == 1 and op == "hard" and config == 2:
        xr = ctx.empty((n_local, nil, nxl), np.float32)
        for c0 in range(0, n_local, 64):
            xr.upload(x_obs.download(c0, min(64, n_local - c0)).real, first=c0)
        outr = ctx.empty((n_local, nil, nxl), np.float32)
        ctx.sync()

        def job_real(niter):
            st = plan.prime_dev(xr.ptr, _ffi.P3D_F32, mask_t.ptr, n_local)   # (row pairs: the statistics pass is the job's first pass)
            act = st[:, 2] > 0
            st[~act] = 1.0
            tau_r = P._schedule_from_stats(st, nil * nxl, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(xr.ptr, _ffi.P3D_F32, mask_t.ptr, tau_r, niter, outr.ptr, n_local,
                                thresh_op=op, eps=args.eps, alpha=args.alpha, active=act, want_sums=False, primed=True)
        job_real(min(W, 3) or 1)
        ctx.sync()
        rt = []
        for _ in range(3):
            r0 = time.perf_counter()
            job_real(K)
            ctx.sync()
            rt.append(time.perf_counter() - r0)
        real_its = K / float(np.median(rt))
        xr.free()
        outr.free()

    # ---- the same loop in the reference's double precision (p3d_f64.hip), a sample of the cube (rank 0, N = 1, config 2 only) ----
    ref_prec = None
    if side and world == 1 and config == 2 and not cfg["real"]:
        n64 = min(32, n_local)
        plan64 = _ffi.Plan64(nil, nxl, n64, device=dev_index)
        host64 = x_obs.download(0, n64)
        st64 = plan64.stats(host64)
        st64[st64[:, 2] == 0] = 1.0
        tau64 = P._schedule_from_stats(st64, nil * nxl, "exponential", K, 0.99, p_min, "values")
        ms64 = [plan64.run(host64, mask, tau64, K, thresh_op=op, eps=args.eps, alpha=args.alpha)[3] for _ in range(3)]
        plan64.close()
        it64 = float(np.median(ms64)) / K            # ms per iteration of the n64-slice sample (device time of the loop)
        ref_prec = {
            "what": f"the same job with precision='reference': the loop in double precision (the reference's arithmetic for soft / garrote / FPOCS / APOCS and "
                    f"for every run under NumPy < 2), two fused kernels per iteration on the mixed-radix register engine with complex128 elements (p3d_mix64.hip: "
                    f"8 points per thread, twiddles from memory, sparse shortcut; round 4: LDS-resident tiles, col64_kernel / row64_kernel) -- a sample of {n64} slices "
                    f"of the cube, complex64 in and out, device time of the {K}-iteration loop",
            "slice_iterations_per_s": n64 / (it64 * 1e-3), "iterations_per_s_of_the_cube": n64 / (it64 * 1e-3) / cube_slices,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_point": 56.0, "achieved": 56.0 * nil * nxl * n64 / (it64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": 56.0 * nil * nxl * n64 / (it64 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "note": "28 B/point of the float32 accounting, doubled; the two fused passes move 16 B/point per read or write of the work buffer (emptied "
                                 "column tiles excepted), the observed sample and an 8-byte weight in the row pass"},
        }
        del host64

    # ---- the WAVELET loop in the reference's double precision (p3d_wavelet64.hip): the slices the CPU leg computes, the full schedule ----
    also64 = None
    if kind == "WAVELET" and rank == 0 and world == 1 and n_cpu > 0 and args.eps == 0:
        n64 = n_local   # (a batch of 16 slices is bound by its ~40 launches per iteration: 186 against 303 cube-iterations/s for the whole block)
        host64 = x_obs.download(0, n64)
        with _ffi.WaveletPlan64(nil, nxl, n64, wavelet=cfg["wavelet"], device=dev_index) as plan64:
            tau64 = P._wavelet_schedule_from_stats(plan64.stats(host64), "exponential", K, 0.99, p_min, "values")
            runs = [plan64.run(host64, mask, tau64, K, thresh_op=op, eps=args.eps, alpha=args.alpha) for _ in range(2)]
        it64 = min(r[3] for r in runs) / K            # ms per iteration of this rank's block (device time of the loop)
        also64 = {"reference_precision": runs[-1][0][:n_cpu]}
        b64 = 2.0 * ALG_BYTES["WAVELET"](0) * nil * nxl * n64
        ref_prec = {
            "what": f"the same job with precision='reference': the WAVELET loop in double precision (pywt keeps float64 for float64 input, POCS.py:585-609 never "
                    f"narrows) on per-axis kernels without LDS tiles -- the {n64} slices of the block, float32 in and out, device time of the {K}-iteration loop; "
                    f"parity (first {n_cpu} slices) under parity.other_paths.reference_precision",
            "slice_iterations_per_s": n64 / (it64 * 1e-3), "iterations_per_s_of_the_cube": n64 / (it64 * 1e-3) / cube_slices,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_point": 2.0 * ALG_BYTES["WAVELET"](0), "achieved": b64 / (it64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": b64 / (it64 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "note": "the float32 accounting (29.33 B/point), doubled; three launches per level and direction move several times that"},
        }
        del host64

    # ---- end to end: host NumPy cube in -> host NumPy cube out through the host-buffer entry point (every rank its block) ----
    e2e = None
    if not args.only_main and not args.no_end_to_end and kind != "SHEARLET" and density == 0:
        host = x_obs.download()
        kw = dict(transform_kind=kind, thresh_op=op, thresh_model="exponential", eps=args.eps, alpha=args.alpha, p_max=0.99, p_min=p_min,
                  device=dev_index, wavelet=cfg.get("wavelet"))
        # plans, device buffers and streams of ALL chunk workers (four chunks of the size the timed call will use), not the cube's page-locking
        warm = min(n_local, 64)
        P.pocs_cube(host[:warm], mask, niter=2, batch_slices=max(1, min(warm // 4, (128 << 20) // (nil * nxl * esz))) if kind == "FFT" else None, **kw)
        ctx.fence()
        # three calls, each with a brand-new result array (what a pipeline that calls once per group of slices does); `seconds` is their
        # median, `seconds_each` all three -- the first one also pays for the first page-locking of the process
        # N > 1: the sharded entry point without a collective (sharding.pocs_cube_sharded(gather='none', out=...)): the cube and the result are
        # files in /dev/shm that every rank maps; each rank moves its own block through its own PCIe link, straight from one into the other
        shared = None
        if world > 1:
            from pseudo_3d_interpolation_amd.sharding import pocs_cube_sharded
            tag = f"{os.environ.get('MASTER_PORT', '0')}_{config}"
            paths = [f"/dev/shm/p3d_bench_{name}_{tag}.npy" for name in ("cube", "out")]
            if rank == 0:
                for pth in paths:
                    np.lib.format.open_memmap(pth, mode="w+", dtype=host.dtype, shape=(nslices, nil, nxl)).flush()
            ctx.dist.barrier()
            shared = [np.load(pth, mmap_mode="r+") for pth in paths]
            shared[0][lo:lo + n_local] = host
            ctx.dist.barrier()
        e_each = []
        for _ in range(3):
            res_host = None
            P._timeline = [] if kind == "FFT" else None      # phases of the chunk workers (rank 0's block), kept of the last call
            ctx.fence()
            t0 = time.perf_counter()
            if shared is None:
                res_host = P.pocs_cube(host, mask, niter=K, **kw)
            else:
                res_host = pocs_cube_sharded(shared[0], mask, gather="none", out=shared[1], niter=K, **kw)[lo:lo + n_local]
            ctx.fence()
            t1 = time.perf_counter()
            e_each.append(ctx.max_over_ranks(t1 - t0))
        e_s = sorted(e_each)[1]
        tl, P._timeline = P._timeline, None
        phases = None
        if tl:
            setup = [m for wid, m in tl if wid == 'setup']
            tl = [(wid, m) for wid, m in tl if wid != 'setup']
            tot = {}
            for _, marks in tl:
                for (_, a_), (name, b_) in zip(marks[:-1], marks[1:]):
                    tot[name] = tot.get(name, 0.0) + (b_ - a_)
            nw = len({wid for wid, _ in tl})
            phases = {"workers": nw, "chunks": len(tl), "first_lane_starts_after_ms": 1e3 * (min(m[0][1] for m in setup) - t0) if setup else None,
                      "first_chunk_starts_after_ms": 1e3 * (min(m[0][1] for _, m in tl) - t0),
                      "last_chunk_ends_before_return_ms": 1e3 * (t1 - max(m[-1][1] for _, m in tl)),
                      "summed_over_chunks_ms": {k: 1e3 * v for k, v in tot.items()},
                      "note": "h2d / d2h: copies on the worker's own stream between the caller's page-locked-in-place arrays and the device; prime: statistics pass = first pass; "
                              "loop: the K iterations; the workers run side by side, so the wall share of a phase is its sum / workers"}
        same = bool(np.array_equal(res_host, out.download())) if args.eps == 0 else None
        e2e = {"iterations_per_s": K / e_s, "interpolated_traces_per_s": float(np.count_nonzero(mask == 0)) / e_s, "seconds": e_s,
               "seconds_each": e_each, "host_bytes_in_plus_out": 2 * host.nbytes * world if world == 1 else 2 * host.nbytes,
               "equals_resident_result": same, "phases": phases,
               "what": f"functions.POCS.pocs_cube(host cube, mask, niter={K}) per rank on its block: pageable NumPy array in, NumPy array "
                       f"out, statistics + schedule + iterations + PCIe both ways (FFT: chunks of ~128 MiB, four in flight); max over ranks; median of three calls (seconds_each), phases of the last"}
        if shared is not None:
            e2e["what"] = (f"sharding.pocs_cube_sharded(cube, mask, gather='none', out=result, niter={K}): cube and result are files in /dev/shm mapped by every "
                           f"rank, each rank moves its own block of {n_local} slices host -> device -> host on its own PCIe link (page-locked in place, chunk "
                           f"pipeline), no collective; wall time of the slowest rank, median of three calls")
            del shared, res_host
            ctx.dist.barrier()
            if rank == 0:
                for pth in paths:
                    os.remove(pth)
            res_host = None
        del host, res_host
        P.release_plans()

    # ---- how the rate depends on the data: the same job on denser spectra (rank 0, N = 1) ----
    by_density = None
    if side and world == 1 and roof is not None and density == 0:
        by_density = [{"coefficients_per_slice": "survey recipe (6 plane waves)", "nonzero_block_fraction": nz_fraction,
                       "iterations_per_s": K / seconds, "steady_state_iterations_per_s": steady, "roofline_frac": roof["frac"]}]
        for m in (96, 1024):
            random_spectrum_slices(x_obs, fft_plan, nil, nxl, lo, pool, mask, m)
            job(min(W, 3) or 1)
            ctx.sync()
            d0 = time.perf_counter()
            _, _, d_ms = job(K)
            ctx.sync()
            d_s = time.perf_counter() - d0
            prof, kept = profile_fft()
            it_ms = prof["colpass_ms"] + prof["rowpass_ms"] if prof["colpass_launches"] else d_ms / K
            by_density.append({"coefficients_per_slice": m, "nonzero_block_fraction": kept, "iterations_per_s": K / d_s,
                               "steady_state_iterations_per_s": 1e3 / it_ms, "roofline_frac": alg_bytes / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})

    # ---- the trivial gather of the blocks (outside the timed steps): device tensors, one collective ----
    gather = None
    if world > 1 and main:
        from pseudo_3d_interpolation_amd.sharding import gather_blocks, gather_blocks_to_root
        src = out.tensor.cpu() if ctx.rehearsal else out.tensor
        gather = {}
        for name, fn in (("all_gather", gather_blocks), ("gather_to_root", gather_blocks_to_root)):
            ctx.fence()
            g0 = time.perf_counter()
            full = fn(src, nslices)
            ctx.fence()
            gather[name + "_ms"] = ctx.max_over_ranks(time.perf_counter() - g0) * 1e3
            if rank == 0:
                assert full.shape[0] == nslices
            del full
        gather["bytes_per_rank"] = int(src.numel() * src.element_size())
        gather["note"] = ("all_gather lands the whole cube on every rank; gather_to_root (dist.gather of equal blocks, padded when uneven) only on "
                          "rank 0 -- the 'trivial gather' of north_star; device tensors over RCCL/xGMI, no host round trip")

    cpu = None
    parity = None
    if rank == 0 and cpu_slices is not None:
        budget = args.cpu_seconds if main else min(args.cpu_seconds, 5.0)
        if kind == "SHEARLET":
            # one slice, two iterations on all cores (a slice-iteration is 2 x 125 transforms of 2 Mi points): the timed job's 100 iterations are out of
            # reach of the CPU leg, so the parity figure is that of a SECOND device job of the same two iterations on the same slice
            n_par = 2
            stats = plan.stats_dev(x_obs.ptr, DT, 1)
            tau = P._shearlet_schedule_from_stats(stats, (nil, nxl), "exponential", n_par, 0.99, p_min, "values")
            plan.run_dev(x_obs.ptr, DT, mask_t.ptr, tau, n_par, out.ptr, 1, thresh_op=op, eps=0.0, alpha=args.alpha)
            gpu_two = out.download(0, 1)[0]
            plan.close()
            x_obs.free(); out.free()
            cpu, ref_two = cpu_baseline_shearlet(cpu_slices[0], mask, psi, op, cube_slices, niter=n_par)
            rel = float(np.linalg.norm(gpu_two - ref_two) / np.linalg.norm(ref_two))
            parity = {"rel_l2_max": rel, "rel_l2_median": rel, "slices": 1, "niter": n_par, "tolerance": 1e-5,
                      "against": "oracle/shearlet_oracle.py (NumPy / SciPy restatement; its FFST spectra are restated from the paper: transform parity UNPINNED, "
                                 "the reference's SHEARLET branches of schedule and loop are pinned) fed the same observed slice in double precision",
                      "what": f"slice 0 of the timed cube through a separate device job of {n_par} iterations (the schedule of a {n_par}-iteration run) vs the oracle's "
                              f"{n_par} iterations -- the CPU leg cannot follow the timed job's {K} iterations of 125 shearlets on 2 Mi points"}
            if world == 1 and not args.only_main:
                # the same two iterations in the reference's double precision (p3d_shearlet64.hip), and the device time of a short loop of it
                with _ffi.ShearletPlan64(psi, max_slices=1, device=dev_index) as plan64:
                    x1 = np.ascontiguousarray(cpu_slices[0], dtype=np.float32 if cfg["real"] else np.complex64)
                    tau64 = P._shearlet_schedule_from_stats(plan64.stats(x1), (nil, nxl), "exponential", n_par, 0.99, p_min, "values")
                    got64 = plan64.run(x1, mask, tau64, n_par, thresh_op=op, eps=0.0, alpha=args.alpha)[0][0]
                    k64 = 6
                    tau64 = P._shearlet_schedule_from_stats(plan64.stats(x1), (nil, nxl), "exponential", k64, 0.99, p_min, "values")
                    ms64 = min(plan64.run(x1, mask, tau64, k64, thresh_op=op, eps=0.0, alpha=args.alpha)[3] for _ in range(2))
                rel64 = float(np.linalg.norm(got64 - ref_two) / np.linalg.norm(ref_two))
                parity["other_paths"] = {"reference_precision": {"rel_l2_max": rel64, "rel_l2_median": rel64, "slices": 1, "niter": n_par}}
                it64 = ms64 / k64
                b64 = 2.0 * ALG_BYTES["SHEARLET"](nsh) * nil * nxl
                ref_prec = {
                    "what": f"the same job with precision='reference': the SHEARLET loop in double precision (np.fft inside FFST computes in double, POCS.py:589-619 "
                            f"never narrows), three fused passes over the coefficients on the double-precision register engine (p3d_mix64.hip) -- slice 0 of the cube, float32 in and out, device time of a {k64}-iteration "
                            f"loop; parity under parity.other_paths.reference_precision",
                    "slice_iterations_per_s": 1.0 / (it64 * 1e-3), "iterations_per_s_of_the_cube": 1.0 / (it64 * 1e-3) / cube_slices,
                    "roofline": {"bound": "hbm", "algorithmic_bytes_per_point": b64 / (nil * nxl), "achieved": b64 / (it64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": b64 / (it64 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 "note": "the float32 accounting (40 B per point and shearlet), doubled: spectrum x Psi -> coefficients (16 + 8), threshold in "
                                         "place between two column transforms (32), coefficients x Psi -> sum (16 + 8), every row of every shearlet counted -- the passes skip the "
                                         "rows on which a shearlet's spectrum vanishes (p3d_shearlet64_info: row_group_fraction), and a real cube works on Hermitian coefficient slices (rows 0 ... nil/2, "
                                         "two columns per transform), so `achieved` is above what moves and may exceed the peak"},
                }
        else:
            plan.close()
            x_obs.free(); out.free()
            cs = cpu_slices if not cfg["real"] else cpu_slices.real.astype(np.float64)
            parity, cpu = cpu_baseline(kind, cs, mask, op, budget, cube_slices, extra=cfg.get("wavelet"), parity_of=gpu_first, parity_niter=K, also=also64)
    else:
        plan.close()
        x_obs.free(); out.free()
    mask_t.free()
    del pool

    if rank != 0:
        return None
    scale = nslices / cube_slices   # config 4: the sample's rate, expressed per whole cube
    its = K / seconds * scale
    dtype_s = "float32" if cfg["real"] else "complex64"
    tag = f" (BASELINE configs[{config}])" if (nil, nxl, missing, op) == tuple(CONFIGS[config][k] for k in ("nil", "nxl", "missing", "op")) and \
        (not main or args.nslices is None or kind == "SHEARLET") else ""
    what = {"FFT": "FFT transform", "WAVELET": f"wavelet transform ({cfg.get('wavelet')}, mode 'smooth')", "SHEARLET": f"shearlet transform ({nsh} shearlets)"}[kind]
    workload = (f"{nil}x{nxl}x{cube_slices} {dtype_s} cube, {int(missing * 100)}% missing traces, {what}, {op} threshold, exponential decay, "
                f"{K} iterations" + tag)
    if cube_slices != nslices:
        workload += (f"; timed on a sample of {nslices} of its {cube_slices} slices ({n_local} per GPU, each GPU's sample taken from its own block of "
                     f"the cube, batches of {min(n_local, SHEARLET_BATCH)}), all {K} iterations of the schedule, rate scaled by {nslices}/{cube_slices}")
    return {
        "metric": f"POCS iterations/s on the {nil}x{nxl}x{cube_slices} cube",
        "value": its,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": seconds * 1e3 / K / scale,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": f"{dtype_s} (f32 arithmetic)",
        "data": ("synthetic: SURVEY 8d's generator for every slice (6 complex plane waves + 1% complex Gaussian noise from default_rng(1234 + s), random trace "
                 "mask from default_rng(42)), generated on the host and uploaded before the clock starts; the first slices (those the cpu_baseline / parity leg "
                 "computes) are the oracle's own arrays, the others its recipe and random stream re-stated in bench.py"
                 if density == 0 else
                 f"synthetic: {density} random spectral coefficients + 1% Gaussian noise per slice (seeded), random trace mask"),
        "config": {
            "workload": workload,
            "slices_per_gpu": n_local,
            "parallelism": f"slice axis in {world} contiguous block(s), one rank per GPU, no collective in the loop",
        },
        "repeats": {"n": R, "job_s_median": seconds, "job_s_min": float(np.min(times)), "job_s_max": float(np.max(times)),
                    "iterations_per_s_best": K / float(np.min(times)) * scale,
                    "note": "`value` = steps / median job time; a job = statistics + schedule + the K iterations + final store"},
        # rank 0's loop time for its n_local slices; a sample leg's ranks run their shares side by side, so the cube rate is x sample/cube
        "steady_state_iterations_per_s": None if steady is None else steady * scale,
        # (the steady rate comes from the per-launch HIP events of a profiled repeat; on cubes whose iteration takes a fraction
        # of a millisecond the event records themselves lengthen it, and the difference below would come out negative: null)
        "fixed_ms_per_job": None if steady is None or seconds * 1e3 < K * 1e3 / steady else seconds * 1e3 - K * 1e3 / steady,
        "slice_iterations_per_s": K / seconds * nslices,
        "interpolated_traces_per_s": float(np.count_nonzero(mask == 0)) / (seconds / scale),
        "device_ms_rank0": float(np.median(dev_times)),
        "gather_ms": None if gather is None else gather["all_gather_ms"],
        "gather": gather,
        "sparse_spectrum": None if kind != "FFT" else {
            "nonzero_block_fraction": nz_fraction,
            "note": "8-column blocks of the thresholded spectrum that kept a coefficient (rank 0, mean over slices and "
                    "iterations); emptied blocks are not transformed back, stored or re-read -- exact. Data dependent: "
                    "dense_path_iterations_per_s is the rate with the shortcut off (P3D_NO_SPARSE=1), from rank 0's block; "
                    "by_density repeats the job on denser synthetic spectra",
            "dense_path_iterations_per_s": dense_its,
            "by_density": by_density,
        },
        "real_cube": None if real_its is None else {
            "iterations_per_s": real_its,
            "note": "the same job on the real part of the cube as float32 (a time-domain cube): rows share one complex transform "
                    "in pairs and the work buffer holds half the spectrum; not the metric's configuration (complex64 slices)",
        },
        "end_to_end": e2e,
        "roofline": roof,
        "cpu_baseline": cpu,
        "parity": parity,
        "strong_scaling_model": block8,
        "reference_precision": ref_prec,
    }


def compact(rec):
    """What an `other_configs` entry keeps of a leg's record: enough to recompute its rate and roofline fraction."""
    keep = ("value", "unit", "steps", "ms_per_step", "dtype", "steady_state_iterations_per_s", "fixed_ms_per_job", "slice_iterations_per_s",
            "interpolated_traces_per_s", "end_to_end", "roofline", "cpu_baseline", "parity", "reference_precision")
    out = {"workload": rec["config"]["workload"], "slices_per_gpu": rec["config"]["slices_per_gpu"]}
    out.update({k: rec[k] for k in keep})
    out["repeats"] = {k: rec["repeats"][k] for k in ("n", "job_s_median", "job_s_min", "job_s_max")}
    if rec.get("sparse_spectrum"):
        out["nonzero_block_fraction"] = rec["sparse_spectrum"]["nonzero_block_fraction"]
    return out


def release(ctx):
    """Between legs: the device arrays of the finished leg were freed explicitly; drop what Python still holds."""
    import gc
    gc.collect()
    ctx.sync()
    if ctx.torch is not None:
        ctx.torch.cuda.empty_cache()


def main():
    args = parse_args()
    ctx = Ctx(args)
    t_all = time.perf_counter()
    line = run_leg(ctx, args.config, args.steps, main=True)
    release(ctx)
    plain = all(v is None for v in (args.nil, args.nxl, args.nslices, args.missing, args.thresh_op)) and args.density == 0
    if args.config == 2 and plain and not args.only_main:
        others = {}
        for c in ((1, 3, 4) if ctx.world == 1 else (4,)):
            t0 = time.perf_counter()
            try:
                rec = run_leg(ctx, c, None, main=False)
                release(ctx)
                if ctx.rank == 0:
                    others[str(c)] = compact(rec)
                    others[str(c)]["leg_wall_s"] = time.perf_counter() - t0
            except Exception as exc:  # noqa: BLE001 -- a failing side leg must not take the metric's line with it
                if ctx.rank == 0:
                    others[str(c)] = {"error": f"{type(exc).__name__}: {exc}"}
        if ctx.rank == 0:
            line["other_configs"] = others
        if ctx.world == 1:
            # survey grids are rarely powers of two: the same job (configs[2]'s parameters) on a 1000 x 1000 x 128 cube -- 7-smooth extents run the
            # mixed-radix register engine (csrc/p3d_mix.hpp), not the tuned power-of-two kernels
            try:
                rec = run_leg(ctx, 2, args.steps, main=False, override=dict(nil=1000, nxl=1000, nslices=128))
                release(ctx)
                sg = compact(rec)
                sg["points_per_s"] = rec["slice_iterations_per_s"] * 1000 * 1000
                sg["note"] = ("BASELINE configs[2]'s job on a 1000 x 1000 x 128 cube (80 % missing, hard threshold, exponential decay): 1000 = 5 x 10 x 20 on the "
                              "mixed-radix register engine; roofline on the same 28 B/point accounting; no CPU sample for this leg")
                line["smooth_grid"] = sg
            except Exception as exc:  # noqa: BLE001
                line["smooth_grid"] = {"error": f"{type(exc).__name__}: {exc}"}
    if ctx.rank == 0:
        line["kernel_source_hash"] = kernel_source_hash()
        from pseudo_3d_interpolation_amd import _ffi as _f
        line["hip_runtime"] = _f.runtime_info()
        line["bench_wall_s"] = time.perf_counter() - t_all
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
