#!/usr/bin/env python3
"""
bench.py -- POCS iterations/s (BASELINE.json metric) on N MI355X GPUs.

    python bench.py [--gpus 1] [--steps K] [--warmup W] [--config {0,1,2,3,4}] [--repeats R] [--density M]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one POCS iteration of the whole cube: every (iline, xline) slice goes once through forward transform ->
threshold -> inverse transform -> re-insertion of the observed traces -> cost sum.  The timed region is one complete job of K
iterations on a cube already resident in HBM: statistics of transform(x_obs), threshold schedule (host, a few scalars per
slice), the K iterations, final store of the result.  The job is repeated R (>= 5) times; `value` is K / the MEDIAN job time
(max over ranks per repeat), min and max are printed beside it.  For N > 1 the slice axis is cut into N contiguous blocks,
one rank per GPU; there is no collective inside the timed region -- the blocks are gathered with one RCCL all_gather
afterwards (timed separately, "gather_ms").  Rank 0 prints ONE JSON line.

--config selects the BASELINE.json configuration (default 2 = the metric's cube, 1024 x 1024 x 512 complex64):
    0  64 x 64 x 128 complex64, 50 % missing, FFT, hard, 20 iterations
    1  512 x 512 x 256 complex64, 70 % missing, FFT, hard, exponential decay, 50 iterations
    2  1024 x 1024 x 512 complex64, 80 % missing, FFT, hard, exponential decay, 100 iterations
    3  512 x 512 x 256 float32, 70 % missing, WAVELET db4 ('smooth'), soft, 50 iterations
    4  2048 x 1024 x 1024 float32, 80 % missing, SHEARLET (125 shearlets), hard, 100 iterations -- timed on a SAMPLE of the
       cube's slices (--nslices, default 8) and of its iterations (default 5): one GPU holds 2 GiB of coefficients per slice
--steps overrides the iteration count (the driver runs --steps 20).

Inputs are generated without any torch random-number kernel (NumPy noise pool + plane waves / index writes + the library's
own inverse FFT), so that `rocprofv3 --pmc ... -- python3 bench.py` completes (round 1: counter collection aborted inside
torch's normal_ kernel).
"""
import argparse
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

CONFIGS = {
    0: dict(kind="FFT", nil=64, nxl=64, nslices=128, missing=0.5, steps=20, op="hard", real=False),
    1: dict(kind="FFT", nil=512, nxl=512, nslices=256, missing=0.7, steps=50, op="hard", real=False),
    2: dict(kind="FFT", nil=1024, nxl=1024, nslices=512, missing=0.8, steps=100, op="hard", real=False),
    3: dict(kind="WAVELET", nil=512, nxl=512, nslices=256, missing=0.7, steps=50, op="soft", real=True, wavelet="db4"),
    4: dict(kind="SHEARLET", nil=2048, nxl=1024, nslices=8, cube_slices=1024, missing=0.8, steps=5, cube_steps=100, op="hard", real=True),
}
# ALGORITHMIC bytes per point and iteration:
#   FFT, complex64: iterate read 8 + written 8 + observed data 8 + float32 weight 4                                        = 28
#   WAVELET, float32: (4 read + 4 written) per transform x 2 transforms x 4/3 (coarser levels) + observed 4 + weight 4    = 29.33
#   SHEARLET: per shearlet 12 (spectrum x Psi -> coefficients) + 16 (threshold in place) + 12 (coefficients -> sum)        = 40 nsh
ALG_BYTES = {"FFT": lambda nsh: 28.0, "WAVELET": lambda nsh: 8.0 * 2 * 4 / 3 + 8.0, "SHEARLET": lambda nsh: 40.0 * nsh}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[i] (default: the metric's cube)")
    ap.add_argument("--steps", type=int, default=None, help="POCS iterations in the timed job (default: the configuration's)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of the job (0: as many as fill ~1.5 s, between 5 and 15)")
    ap.add_argument("--nil", type=int, default=None)
    ap.add_argument("--nxl", type=int, default=None)
    ap.add_argument("--nslices", type=int, default=None, help="slices of the whole cube (sharded over the GPUs)")
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
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------------
# synthetic inputs (no torch RNG)
# ---------------------------------------------------------------------------------------------------------------------------------
NOISE_POOL = 16


def noise_pool(torch, nil, nxl, device):
    """NOISE_POOL complex Gaussian slices from NumPy, uploaded once; slice s uses pool[s % P] rolled by a slice-dependent shift."""
    rng = np.random.default_rng(20240917)
    host = (rng.standard_normal((NOISE_POOL, nil, nxl), dtype=np.float32)
            + 1j * rng.standard_normal((NOISE_POOL, nil, nxl), dtype=np.float32)).astype(np.complex64)
    return torch.from_numpy(host).to(device)


def _add_noise(torch, out, first, pool):
    """out[i] += 0.01 * pool[s % P] rolled by a shift that depends on s // P (s = first + i), one roll per group of P slices."""
    n = out.shape[0]
    i = 0
    while i < n:
        s = first + i
        r, j0 = s // NOISE_POOL, s % NOISE_POOL
        cnt = min(NOISE_POOL - j0, n - i)
        out[i:i + cnt] += 0.01 * torch.roll(pool[j0:j0 + cnt], shifts=(3 * r + 1, 5 * r + 2), dims=(1, 2))
        i += cnt


def fill_plane_waves(torch, out, nil, nxl, first, pool, batch=16):
    """The recipe of oracle.synthetic_slice: 6 complex plane waves + 1 % Gaussian noise per slice, generated on the GPU
    `batch` slices per kernel launch (a few hundred launches for the whole cube: rocprofv3's counter collection aborts inside
    whichever torch kernel happens to be the N-thousandth dispatch of a process -- profiles/r02_pmc_on_bench.txt)."""
    device = out.device
    il = (torch.arange(nil, device=device, dtype=torch.float32) / nil)[None, :, None]
    xl = (torch.arange(nxl, device=device, dtype=torch.float32) / nxl)[None, None, :]
    n = out.shape[0]
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        k1 = np.empty((b1 - b0, 6), np.float32); k2 = np.empty_like(k1); amp = np.empty((b1 - b0, 6), np.complex64)
        for i in range(b0, b1):
            rng = np.random.default_rng(1234 + first + i)
            for e in range(6):   # the draw order of oracle.synthetic_slice
                k1[i - b0, e] = int(rng.integers(-(nil // 8), max(nil // 8, 1)))
                k2[i - b0, e] = int(rng.integers(-(nxl // 8), max(nxl // 8, 1)))
                amp[i - b0, e] = complex(rng.standard_normal(), rng.standard_normal())
        k1t, k2t, ampt = (torch.from_numpy(a).to(device) for a in (k1, k2, amp))
        acc = out[b0:b1]
        acc.zero_()
        for e in range(6):
            ph = (2.0 * np.pi) * (k1t[:, e, None, None] * il + k2t[:, e, None, None] * xl)
            acc += ampt[:, e, None, None] * torch.polar(torch.ones_like(ph), ph)
            del ph
    _add_noise(torch, out, first, pool)


def fill_random_spectrum(torch, out, nil, nxl, first, pool, m, plan):
    """M random spectral coefficients per slice (positions uniform over the spectrum, complex normal amplitudes, seeded per
    slice), brought to the space domain by the library's own inverse FFT, + 1 % noise."""
    device = out.device
    n = out.shape[0]
    out.zero_()
    flat = out.view(n, -1)
    chunk = 64
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        rows, cols, vals = [], [], []
        for i in range(lo, hi):
            rng = np.random.default_rng(4321 + first + i)
            pos = rng.choice(nil * nxl, size=m, replace=False)
            amp = (rng.standard_normal(m) + 1j * rng.standard_normal(m)) * (nil * nxl / math.sqrt(m))
            rows.append(np.full(m, i, np.int64)); cols.append(pos.astype(np.int64)); vals.append(amp.astype(np.complex64))
        flat[torch.from_numpy(np.concatenate(rows)).to(device), torch.from_numpy(np.concatenate(cols)).to(device)] = \
            torch.from_numpy(np.concatenate(vals)).to(device)
    torch.cuda.synchronize()
    plan.fft2_dev(out.data_ptr(), out.data_ptr(), n, inverse=True)
    _add_noise(torch, out, first, pool)


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on a bounded sample of the same cube, one single-threaded NumPy process per core (the shape of the
# reference's LocalCluster(processes=True, threads_per_worker=1)); the shearlet slice is too big for that and uses all cores on ONE
# slice through scipy.fft's threads instead
# ---------------------------------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    kind, x, mask, niter, op, extra = job
    t0 = time.perf_counter()
    if kind == "FFT":
        from oracle import pocs_oracle as orc
        orc.pocs_slice(x, mask, niter=niter, thresh_op=op, thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
    elif kind == "WAVELET":
        from oracle import wavelet_oracle as wo
        wo.pocs_slice_wavelet(x, mask, wavelet=extra, niter=niter, thresh_op=op, thresh_model="exponential", eps=0.0, p_max=0.99, p_min=1e-3)
    return time.perf_counter() - t0


def cpu_baseline(kind, obs_slices, mask, op, budget_s, nslices_cube, extra=None):
    import multiprocessing as mp

    workers = len(obs_slices)
    pts = obs_slices[0].size
    per_it = {"FFT": 93.3e-3, "WAVELET": 0.35}[kind] * pts / (1024 * 1024)   # s per slice-iteration on one core (measured)
    niter = int(max(4, min(100, budget_s / max(per_it, 1e-6))))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        tiny = (kind, obs_slices[0][:16, :16].copy(), mask[:16, :16].copy(), 2, op, extra)
        pool.map(_cpu_worker, [tiny] * workers)  # spin up
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(kind, s, mask, niter, op, extra) for s in obs_slices])
        wall = time.perf_counter() - t0
    slice_iters_per_s = workers * niter / wall
    return {
        "value": slice_iters_per_s / nslices_cube,
        "unit": "iterations/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} slices x {niter} iterations of the same cube, one NumPy process per core, scaled by 1/{nslices_cube} slices",
        "slice_iterations_per_s": slice_iters_per_s,
    }


def cpu_baseline_shearlet(x, mask, psi, op, nslices_cube):
    from oracle import shearlet_oracle as so
    cores = min(os.cpu_count() or 1, 16)   # the GPU box's CPU share for one GPU
    niter = 2
    so.pocs_slice_shearlet_real(x[:64, :32].astype(np.float64), mask[:64, :32], np.ones((64, 32, 1)), niter=2, thresh_op=op, p_min=1e-3)  # imports, threads
    t0 = time.perf_counter()
    so.pocs_slice_shearlet_real(x.astype(np.float64), mask, psi, niter=niter, thresh_op=op, thresh_model="exponential", p_max=0.99, p_min=1e-3,
                                workers=cores)
    wall = time.perf_counter() - t0
    sips = (niter + 1) / wall   # the schedule's transform + `niter` iterations: niter + 1 forward transforms, niter inverse ones ~ niter + 1/2 iterations
    return {
        "value": sips / nslices_cube, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": f"1 slice x {niter} iterations (+ the schedule's transform) of the same cube, scipy.fft real transforms on {cores} threads, "
                  f"scaled by 1/{nslices_cube} slices",
        "slice_iterations_per_s": sips,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    cfg = dict(CONFIGS[args.config])
    for key, val in (("nil", args.nil), ("nxl", args.nxl), ("nslices", args.nslices), ("missing", args.missing), ("op", args.thresh_op)):
        if val is not None:
            cfg[key] = val
    kind, nil, nxl, nslices, missing, op = cfg["kind"], cfg["nil"], cfg["nxl"], cfg["nslices"], cfg["missing"], cfg["op"]
    K = args.steps if args.steps is not None else cfg["steps"]
    W = args.warmup
    cube_slices = cfg.get("cube_slices", nslices) if args.nslices is None else nslices   # config 4: a sample stands for the cube

    import torch
    import torch.distributed as dist

    from pseudo_3d_interpolation_amd import _ffi
    from pseudo_3d_interpolation_amd.functions import POCS as P
    from pseudo_3d_interpolation_amd.sharding import slice_block

    # P3D_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo -- exercises the multi-rank code path where no second GPU exists
    # (the numbers mean nothing then).  The real thing: one rank per GPU over RCCL.
    rehearsal = os.environ.get("P3D_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    lo, hi = slice_block(nslices, world, rank)
    n_local = hi - lo
    pts_local = n_local * nil * nxl
    p_min = args.p_min if args.p_min == "adaptive" else float(args.p_min)

    # ---- inputs, resident in HBM before the clock starts -----------------------------------------
    mask = (np.random.default_rng(42).random((nil, nxl)) >= missing).astype(np.uint8)   # shared trace mask
    mask_t = torch.from_numpy(mask.astype(np.float32)).to(device)
    pool = noise_pool(torch, nil, nxl, device)
    fft_plan = _ffi.Plan(nil, nxl, n_local, device=dev_index) if kind == "FFT" else None
    xc = torch.empty((n_local, nil, nxl), dtype=torch.complex64, device=device)

    def generate(density):
        if density > 0:
            fill_random_spectrum(torch, xc, nil, nxl, lo, pool, density, fft_plan)
        else:
            fill_plane_waves(torch, xc, nil, nxl, lo, pool)
        xc.mul_(mask_t)

    generate(args.density if kind == "FFT" else 0)
    n_cpu = 0
    cpu_slices = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.density == 0:   # the CPU baseline is a single-GPU-run extra (rank 0, N = 1)
        from oracle import pocs_oracle as orc  # the cpu_baseline leg (and the slices it is fed) -- nothing else touches the oracle
        n_cpu = max(1, min(os.cpu_count() or 1, 16, n_local)) if kind != "SHEARLET" else 1
        cpu_slices = np.stack([orc.synthetic_slice(nil, nxl, lo + s, real=cfg["real"]) for s in range(n_cpu)]) * mask
        xc[:n_cpu] = torch.from_numpy(cpu_slices.astype(np.complex64)).to(device)   # the CPU sample sees exactly the slices the GPU processes
    if cfg["real"]:
        x_obs = xc.real.contiguous()
        del xc
    else:
        x_obs = xc
    out = torch.empty_like(x_obs)
    torch.cuda.synchronize()
    DT = _ffi.P3D_F32 if cfg["real"] else _ffi.P3D_C64

    # ---- the job of each transform kind ------------------------------------------------------------
    nsh = 0
    psi = None
    if kind == "FFT":
        plan = fft_plan

        def job(niter, profile=False):
            # statistics of fft2(x_obs) with the mask at hand: the pass doubles as the first pass of the job (p3d_pocs_prime_dev)
            stats = plan.prime_dev(x_obs.data_ptr(), DT, mask_t.data_ptr(), n_local)
            active = stats[:, 2] > 0
            stats[~active] = 1.0
            tau = P._schedule_from_stats(stats, nil * nxl, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(x_obs.data_ptr(), DT, mask_t.data_ptr(), tau, niter, out.data_ptr(), n_local,
                                thresh_op=op, eps=args.eps, alpha=args.alpha, active=active, profile=profile, want_sums=False, primed=True)
    elif kind == "WAVELET":
        plan = _ffi.WaveletPlan(nil, nxl, n_local, wavelet=cfg["wavelet"], device=dev_index)

        def job(niter, profile=False):
            stats = plan.stats_dev(x_obs.data_ptr(), DT, n_local)
            tau = P._wavelet_schedule_from_stats(stats, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(x_obs.data_ptr(), DT, mask_t.data_ptr(), tau, niter, out.data_ptr(), n_local, thresh_op=op, eps=args.eps,
                                alpha=args.alpha)
    else:
        from pseudo_3d_interpolation_amd.functions import shearlets
        psi = shearlets.scalesShearsAndSpectra((nil, nxl), dtype=np.float32)
        nsh = psi.shape[-1]
        plan = _ffi.ShearletPlan(psi, max_slices=n_local, device=dev_index)

        def job(niter, profile=False):
            stats = plan.stats_dev(x_obs.data_ptr(), DT, n_local)
            tau = P._shearlet_schedule_from_stats(stats, (nil, nxl), "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(x_obs.data_ptr(), DT, mask_t.data_ptr(), tau, niter, out.data_ptr(), n_local, thresh_op=op, eps=args.eps,
                                alpha=args.alpha)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(niter):
        """One job of `niter` iterations between two fences; seconds = max over ranks."""
        fence()
        t0 = time.perf_counter()
        res = job(niter)
        fence()
        sec = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([sec], dtype=torch.float64, device="cpu" if rehearsal else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        return sec, res

    if W > 0:
        job(W)
    first_s, (done, _, dev_ms) = timed(K)
    assert args.eps > 0 or (int(done.min()) == K and int(done.max()) == K)
    R = args.repeats if args.repeats > 0 else int(max(5, min(15, math.ceil(1.5 / max(first_s, 1e-4)))))
    times, dev_times = [first_s], [dev_ms]
    for _ in range(R - 1):
        sec, (_, _, dms) = timed(K)
        times.append(sec)
        dev_times.append(dms)
    seconds = float(np.median(times))

    nz_fraction = plan.last_sparsity() if kind == "FFT" else -1.0

    # ---- per-kernel durations of the same job, HIP events on the plan's stream --------------------
    alg_bytes = ALG_BYTES[kind](nsh) * pts_local

    def profile_fft():
        job(K, profile=True)   # the same job once more with HIP events around every pass (same schedule, same sparsity as the timed one)
        prof = plan.last_profile()
        kept = plan.last_sparsity()
        return prof, (1.0 if kept < 0 else kept)

    roof = None
    steady = None
    if rank == 0 and not args.no_profile:
        if kind == "FFT":
            prof, kept = profile_fft()
            it_ms = prof["colpass_ms"] + prof["rowpass_ms"]
            resident = prof["colpass_launches"] == 0   # small slices: the whole job is ONE kernel (p3d_resident.hip), no per-pass events
            if resident:
                it_ms = float(np.median(dev_times)) / K   # HIP events of the library around the kernel, on the plan's stream
            achieved = alg_bytes / (it_ms * 1e-3) / 1e9 if it_ms > 0 else 0.0
            traffic, traffic_from = None, None
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.isfile(tfile):
                try:
                    rec = json.load(open(tfile))
                    if rec.get("workload") == f"{nil}x{nxl}x{n_local}" and args.density == 0:
                        traffic = rec.get("hbm_bytes_per_iteration")
                        traffic_from = (f"profiles/pmc_traffic.json: rocprofv3 --pmc passes, round {rec.get('round')}, of a K = "
                                        f"{rec.get('iterations', 100)} job (this line: K = {K}; the kept-block fraction, hence the traffic, "
                                        f"depends on K: {rec.get('nonzero_block_fraction', 'n/a')} there, {kept:.4f} here)")
                except Exception:  # noqa
                    traffic = None
            steady = 1e3 / it_ms if it_ms > 0 else None
            roof = {
                "bound": "hbm",
                "kernel": (f"resident_kernel<{nil},{nxl}>: one workgroup per slice, all {K} iterations in registers / LDS -- the iterations "
                           f"move NO HBM bytes (16 B/point per job), so `achieved` (algorithmic bytes over time) may exceed the HBM peak"
                           if resident else
                           f"col_kernel<{nil},COL_ITER> + the persistent row pass (row_pipe64_kernel<{nxl}> for rows of whole "
                           f"wavefronts, row_pipe_kernel otherwise) = one POCS iteration of {n_local} slices"),
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_from": traffic_from,
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
            it_ms = float(np.median(dev_times)) / K   # HIP events of the library around the K-iteration loop, on the plan's stream
            achieved = alg_bytes / (it_ms * 1e-3) / 1e9
            steady = 1e3 / it_ms
            w_traffic, w_from = None, None
            wfile = os.path.join(ROOT, "profiles", "r02_wavelet_traffic.json")
            if kind == "WAVELET" and (nil, nxl, nslices) == (512, 512, 256) and world == 1 and os.path.exists(wfile):
                try:   # counters of the same loop, collected by tools/pmc_kernels.sh (one TCC counter per pass)
                    with open(wfile) as fh:
                        w_traffic = json.load(fh).get("hbm_bytes_per_iteration")
                    w_from = "profiles/r02_wavelet_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same loop (tools/wavelet_bench.py), round 2"
                except (OSError, ValueError):
                    w_traffic, w_from = None, None
            sfile = os.path.join(ROOT, "profiles", "r02_shearlet_traffic.json")
            if kind == "SHEARLET" and (nil, nxl) == (2048, 1024) and os.path.exists(sfile):
                try:   # per slice-iteration, scaled to this rank's slices
                    with open(sfile) as fh:
                        w_traffic = json.load(fh).get("hbm_bytes_per_slice_iteration") * n_local
                    w_from = ("profiles/r02_shearlet_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/shearlet_bench.py "
                              "(4 slices of this shape), per slice-iteration x the slices of this rank, round 2")
                except (OSError, ValueError, TypeError):
                    w_traffic, w_from = None, None
            roof = {
                "bound": "hbm",
                "kernel": ("dwt2_tile_kernel / idwt2_tile_kernel chain (one launch per level and direction)" if kind == "WAVELET" else
                           "row_kernel<ROW_SPREAD_INV> + col_kernel<COL_SHRINK> + row_kernel<ROW_GATHER_FWD> + the two fft2 passes")
                          + f" = one POCS iteration of {n_local} slices",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": w_traffic, "traffic_from": w_from,
                "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": it_ms,
            }

    # ---- the same job with the sparse-spectrum shortcut switched off (reported beside `value`, rank 0 only) ----
    dense_its = None
    side = rank == 0 and not args.no_dense and kind == "FFT"
    if side and nz_fraction >= 0:
        os.environ["P3D_NO_SPARSE"] = "1"
        job(min(W, 3) or 1)
        torch.cuda.synchronize()
        d0 = time.perf_counter()
        job(K)
        torch.cuda.synchronize()
        dense_its = K / (time.perf_counter() - d0)   # the ranks run in parallel: rank 0's time for its block is the job's
        del os.environ["P3D_NO_SPARSE"]

    # ---- the real-valued (float32, time-domain) cube of the same shape, reported beside `value` (rank 0, N = 1 only) ----
    real_its = None
    if side and world == 1 and op == "hard" and args.config == 2:
        xr = x_obs.real.contiguous()
        outr = torch.empty_like(xr)
        torch.cuda.synchronize()

        def job_real(niter):
            st = plan.stats_dev(xr.data_ptr(), _ffi.P3D_F32, n_local)
            act = st[:, 2] > 0
            st[~act] = 1.0
            tau_r = P._schedule_from_stats(st, nil * nxl, "exponential", niter, 0.99, p_min, "values")
            return plan.run_dev(xr.data_ptr(), _ffi.P3D_F32, mask_t.data_ptr(), tau_r, niter, outr.data_ptr(), n_local,
                                thresh_op=op, eps=args.eps, alpha=args.alpha, active=act, want_sums=False)
        job_real(min(W, 3) or 1)
        torch.cuda.synchronize()
        rt = []
        for _ in range(3):
            r0 = time.perf_counter()
            job_real(K)
            torch.cuda.synchronize()
            rt.append(time.perf_counter() - r0)
        real_its = K / float(np.median(rt))
        del xr, outr

    # ---- how the rate depends on the data: the same job on denser spectra (rank 0, N = 1) ----
    by_density = None
    if side and world == 1 and roof is not None and args.density == 0:
        by_density = [{"coefficients_per_slice": "survey recipe (6 plane waves)", "nonzero_block_fraction": nz_fraction,
                       "iterations_per_s": K / seconds, "steady_state_iterations_per_s": steady, "roofline_frac": roof["frac"]}]
        for m in (96, 1024):
            generate(m)
            job(min(W, 3) or 1)
            torch.cuda.synchronize()
            d0 = time.perf_counter()
            _, _, d_ms = job(K)
            torch.cuda.synchronize()
            d_s = time.perf_counter() - d0
            prof, kept = profile_fft()
            it_ms = prof["colpass_ms"] + prof["rowpass_ms"] if prof["colpass_launches"] else d_ms / K
            by_density.append({"coefficients_per_slice": m, "nonzero_block_fraction": kept, "iterations_per_s": K / d_s,
                               "steady_state_iterations_per_s": 1e3 / it_ms, "roofline_frac": alg_bytes / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})

    # ---- the trivial gather of the blocks (outside the timed steps) -------------------------------
    gather_ms = 0.0
    if world > 1:
        src = out.cpu() if rehearsal else out
        blocks = [torch.empty_like(src) for _ in range(world)] if n_local * world == nslices else None
        if blocks is not None:
            fence()
            g0 = time.perf_counter()
            dist.all_gather(blocks, src)
            fence()
            gather_ms = (time.perf_counter() - g0) * 1e3
            del blocks

    cpu = None
    if rank == 0 and cpu_slices is not None:
        plan.close()
        del x_obs, out
        torch.cuda.empty_cache()
        if kind == "SHEARLET":
            cpu = cpu_baseline_shearlet(cpu_slices[0], mask, psi, op, cube_slices)
        else:
            cs = cpu_slices if not cfg["real"] else cpu_slices.real.astype(np.float64)
            cpu = cpu_baseline(kind, list(cs), mask, op, args.cpu_seconds, cube_slices, extra=cfg.get("wavelet"))

    if rank == 0:
        scale = nslices / cube_slices   # config 4: the sample's rate, expressed per whole cube
        its = K / seconds * scale
        dtype_s = "float32" if cfg["real"] else "complex64"
        tag = f" (BASELINE configs[{args.config}])" if all(v is None for v in (args.nil, args.nxl, args.missing, args.thresh_op)) and \
            (args.nslices is None or kind == "SHEARLET") else ""
        what = {"FFT": "FFT transform", "WAVELET": f"wavelet transform ({cfg.get('wavelet')}, mode 'smooth')", "SHEARLET": f"shearlet transform ({nsh} shearlets)"}[kind]
        workload = (f"{nil}x{nxl}x{cube_slices} {dtype_s} cube, {int(missing * 100)}% missing traces, {what}, {op} threshold, exponential decay, "
                    f"{K} iterations" + tag)
        if cube_slices != nslices:
            workload += f"; timed on a sample of {nslices} of its {cube_slices} slices, rate scaled by {nslices}/{cube_slices}"
        if kind == "SHEARLET" and K != cfg.get("cube_steps", K):
            workload += f" ({K} of the configuration's {cfg['cube_steps']} iterations: the schedule of a {K}-iteration job)"
        line = {
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
            "data": ("synthetic: 6 plane waves + 1% Gaussian noise per slice (seeded; NumPy noise pool, no torch RNG), random trace mask"
                     if args.density == 0 or kind != "FFT" else
                     f"synthetic: {args.density} random spectral coefficients + 1% Gaussian noise per slice (seeded), random trace mask"),
            "config": {
                "workload": workload,
                "slices_per_gpu": n_local,
                "parallelism": f"slice axis in {world} contiguous block(s), one rank per GPU, no collective in the loop",
            },
            "repeats": {"n": R, "job_s_median": seconds, "job_s_min": float(np.min(times)), "job_s_max": float(np.max(times)),
                        "iterations_per_s_best": K / float(np.min(times)) * scale,
                        "note": "`value` = steps / median job time; a job = statistics + schedule + the K iterations + final store"},
            "steady_state_iterations_per_s": None if steady is None else steady * scale,
            # (the steady rate comes from the per-launch HIP events of a profiled repeat; on cubes whose iteration takes a fraction
            # of a millisecond the event records themselves lengthen it, and the difference below would come out negative: null)
            "fixed_ms_per_job": None if steady is None or seconds * 1e3 < K * 1e3 / steady else seconds * 1e3 - K * 1e3 / steady,
            "slice_iterations_per_s": K / seconds * nslices,
            "interpolated_traces_per_s": float(np.count_nonzero(mask == 0)) / (seconds / scale),
            "device_ms_rank0": float(np.median(dev_times)),
            "gather_ms": gather_ms,
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
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
