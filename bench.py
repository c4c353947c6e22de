#!/usr/bin/env python3
"""
bench.py -- POCS iterations/s on the 1024 x 1024 x 512 cube (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 100 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one POCS iteration of the whole cube: every (iline, xline) slice goes once through
forward FFT2 -> threshold -> inverse FFT2 -> re-insertion of the observed traces -> cost sum.  The
timed region is one complete job of K iterations on a cube already resident in HBM: statistics of
fft2(x_obs), threshold schedule (host, a few scalars per slice), the K iterations, final store of
the result.  For N > 1 the slice axis is cut into N contiguous blocks, one rank per GPU; there is no
collective inside the timed region -- the blocks are gathered with one RCCL all_gather afterwards
(timed separately, "gather_ms").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_POINT = 28  # iterate read 8 + write 8 + observed data 8 + float32 weight 4
HBM_PEAK_GBPS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="POCS iterations in the timed job (configs[2]: 100)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nil", type=int, default=1024)
    ap.add_argument("--nxl", type=int, default=1024)
    ap.add_argument("--nslices", type=int, default=512, help="slices of the whole cube (sharded over the GPUs)")
    ap.add_argument("--missing", type=float, default=0.8)
    ap.add_argument("--thresh-op", default="hard")
    ap.add_argument("--alpha", type=float, default=1.0, help="re-insertion weight (the metric's setting: 1)")
    ap.add_argument("--p-min", default="1e-3", help="final threshold factor or 'adaptive' (the metric's setting: 1e-3)")
    ap.add_argument("--eps", type=float, default=0.0, help="cost threshold of the early exit (0 = run all iterations, the metric's setting)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the extra run with the sparse-spectrum shortcut switched off")
    return ap.parse_args()


def torch_slices(torch, nil, nxl, first, count, device):
    """Same recipe as oracle.synthetic_slice (6 plane waves + 1 % noise), generated on the GPU."""
    il = (torch.arange(nil, device=device, dtype=torch.float32) / nil)[:, None]
    xl = (torch.arange(nxl, device=device, dtype=torch.float32) / nxl)[None, :]
    out = torch.empty((count, nil, nxl), dtype=torch.complex64, device=device)
    for i in range(count):
        rng = np.random.default_rng(1234 + first + i)
        acc = torch.zeros((nil, nxl), dtype=torch.complex64, device=device)
        for _ in range(6):
            k1 = int(rng.integers(-(nil // 8), max(nil // 8, 1)))
            k2 = int(rng.integers(-(nxl // 8), max(nxl // 8, 1)))
            amp = complex(rng.standard_normal(), rng.standard_normal())
            ph = 2.0 * np.pi * (k1 * il + k2 * xl)
            acc += amp * torch.polar(torch.ones_like(ph), ph)
        g = torch.Generator(device=device)
        g.manual_seed(1234 + first + i)
        acc += 0.01 * torch.complex(torch.randn((nil, nxl), generator=g, device=device),
                                    torch.randn((nil, nxl), generator=g, device=device))
        out[i] = acc
    return out


def _cpu_worker(job):
    from oracle import pocs_oracle as orc
    x, mask, niter, op = job
    t0 = time.perf_counter()
    orc.pocs_slice(x, mask, niter=niter, thresh_op=op, thresh_model="exponential", eps=0, p_max=0.99, p_min=1e-3)
    return time.perf_counter() - t0


def cpu_baseline(obs_slices, mask, op, budget_s, nslices_cube):
    """The oracle (NumPy restatement of the reference loop) on a bounded sample of the same cube:
    one single-threaded process per core, one slice per process -- the shape of the reference's
    LocalCluster(processes=True, threads_per_worker=1)."""
    import multiprocessing as mp

    workers = len(obs_slices)
    pts = obs_slices[0].size
    est = 93.3e-3 * pts / (1024 * 1024)  # s per slice-iteration on one core at 1024^2
    niter = int(max(4, min(100, budget_s / max(est, 1e-6))))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        pool.map(_cpu_worker, [(obs_slices[0][:8, :8].copy(), mask[:8, :8].copy(), 2, op)] * workers)  # spin up
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(s, mask, niter, op) for s in obs_slices])
        wall = time.perf_counter() - t0
    slice_iters_per_s = workers * niter / wall
    return {
        "value": slice_iters_per_s / nslices_cube,
        "unit": "iterations/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} slices x {niter} iterations of the same cube, one NumPy process per core, "
                  f"scaled by 1/{nslices_cube} slices",
        "slice_iterations_per_s": slice_iters_per_s,
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

    import torch
    import torch.distributed as dist

    from pseudo_3d_interpolation_amd import _ffi
    from pseudo_3d_interpolation_amd.functions.POCS import _schedule_from_stats
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

    nil, nxl, K, W = args.nil, args.nxl, args.steps, args.warmup
    lo, hi = slice_block(args.nslices, world, rank)
    n_local = hi - lo
    pts_local = n_local * nil * nxl

    # ---- inputs, resident in HBM before the clock starts -----------------------------------------
    mask = (np.random.default_rng(42).random((nil, nxl)) >= args.missing).astype(np.uint8)   # shared trace mask
    mask_t = torch.from_numpy(mask.astype(np.float32)).to(device)
    n_cpu = 0
    cpu_slices = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is a single-GPU-run extra (rank 0, N = 1)
        from oracle import pocs_oracle as orc  # the cpu_baseline leg (and the slices it is fed) -- nothing else touches the oracle
        n_cpu = max(1, min(os.cpu_count() or 1, 16, n_local))
        cpu_slices = np.stack([orc.synthetic_slice(nil, nxl, lo + s) for s in range(n_cpu)]) * mask
    x_obs = torch_slices(torch, nil, nxl, lo, n_local, device)
    x_obs *= mask_t
    if cpu_slices is not None:  # the CPU sample sees exactly the slices the GPU processes
        x_obs[:n_cpu] = torch.from_numpy(cpu_slices).to(device)
    out = torch.empty_like(x_obs)
    torch.cuda.synchronize()

    plan = _ffi.Plan(nil, nxl, n_local, device=dev_index)

    def job(niter, profile=False):
        stats = plan.stats_dev(x_obs.data_ptr(), _ffi.P3D_C64, n_local)
        active = stats[:, 2] > 0
        stats[~active] = 1.0
        p_min = args.p_min if args.p_min == "adaptive" else float(args.p_min)
        tau = _schedule_from_stats(stats, nil * nxl, "exponential", niter, 0.99, p_min, "values")
        return plan.run_dev(x_obs.data_ptr(), _ffi.P3D_C64, mask_t.data_ptr(), tau, niter, out.data_ptr(), n_local,
                            thresh_op=args.thresh_op, eps=args.eps, alpha=args.alpha, active=active, profile=profile,
                            want_sums=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if W > 0:
        job(W)
    fence()
    t0 = time.perf_counter()
    done, _, dev_ms = job(K)
    fence()
    seconds = time.perf_counter() - t0
    assert args.eps > 0 or (int(done.min()) == K and int(done.max()) == K)
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())

    # ---- the same job with the sparse-spectrum shortcut switched off (reported beside `value`, rank 0 only) ----
    nz_fraction = plan.last_sparsity()
    dense_its = None
    if rank == 0 and nz_fraction >= 0 and not args.no_dense:
        os.environ["P3D_NO_SPARSE"] = "1"
        torch.cuda.synchronize()
        d0 = time.perf_counter()
        job(K)
        torch.cuda.synchronize()
        dense_its = K / (time.perf_counter() - d0)   # the ranks run in parallel: rank 0's time for its block is the job's
        del os.environ["P3D_NO_SPARSE"]

    # ---- the real-valued (float32, time-domain) cube of the same shape, reported beside `value` (rank 0, N = 1 only) ----
    real_its = None
    if rank == 0 and world == 1 and not args.no_dense and args.thresh_op == "hard":
        xr = x_obs.real.contiguous()
        outr = torch.empty_like(xr)
        torch.cuda.synchronize()

        def job_real(niter):
            st = plan.stats_dev(xr.data_ptr(), _ffi.P3D_F32, n_local)
            act = st[:, 2] > 0
            st[~act] = 1.0
            tau_r = _schedule_from_stats(st, nil * nxl, "exponential", niter, 0.99, float(args.p_min) if args.p_min != "adaptive" else "adaptive", "values")
            return plan.run_dev(xr.data_ptr(), _ffi.P3D_F32, mask_t.data_ptr(), tau_r, niter, outr.data_ptr(), n_local,
                                thresh_op=args.thresh_op, eps=args.eps, alpha=args.alpha, active=act, want_sums=False)
        job_real(min(W, 3) or 1)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        job_real(K)
        torch.cuda.synchronize()
        real_its = K / (time.perf_counter() - r0)
        del xr, outr

    # ---- the trivial gather of the blocks (outside the timed steps) -------------------------------
    gather_ms = 0.0
    if world > 1:
        src = out.cpu() if rehearsal else out
        blocks = [torch.empty_like(src) for _ in range(world)] if n_local * world == args.nslices else None
        if blocks is not None:
            fence()
            g0 = time.perf_counter()
            dist.all_gather(blocks, src)
            fence()
            gather_ms = (time.perf_counter() - g0) * 1e3
            del blocks

    # ---- per-kernel durations of the same job, HIP events on the plan's stream --------------------
    roof = None
    if rank == 0 and not args.no_profile:
        job(K, profile=True)   # the same job once more with HIP events around every pass (same schedule, same sparsity as the timed one)
        prof = plan.last_profile()
        it_ms = prof["colpass_ms"] + prof["rowpass_ms"]
        kept = plan.last_sparsity()
        kept = 1.0 if kept < 0 else kept
        alg_bytes = ALG_BYTES_PER_POINT * pts_local
        achieved = alg_bytes / (it_ms * 1e-3) / 1e9 if it_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(tfile):
            try:
                rec = json.load(open(tfile))
                if rec.get("workload") == f"{nil}x{nxl}x{n_local}":
                    traffic = rec.get("hbm_bytes_per_iteration")
            except Exception:  # noqa
                traffic = None
        roof = {
            "bound": "hbm",
            "kernel": f"col_kernel<{nil},COL_ITER> + the persistent row pass (row_pipe64_kernel<{nxl}> for rows of whole "
                      f"wavefronts, row_pipe_kernel otherwise) = one POCS iteration of {n_local} slices",
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes,
            "launch_ms": it_ms,
            "colpass_ms": prof["colpass_ms"], "rowpass_ms": prof["rowpass_ms"],
            # bytes the two passes move per point with a fraction f of the spectrum's column blocks kept: column pass 8 read +
            # 8 f written; row pass 8 f read + 8 written + 8 (1 - missing) observed samples (compact)
            "colpass_moved_GBps": (8 + 8 * kept) * pts_local / (prof["colpass_ms"] * 1e-3) / 1e9 if prof["colpass_ms"] else 0.0,
            "rowpass_moved_GBps": (8 * kept + 8 + 8 * (1 - args.missing)) * pts_local / (prof["rowpass_ms"] * 1e-3) / 1e9
            if prof["rowpass_ms"] else 0.0,
        }

    cpu = None
    if rank == 0 and cpu_slices is not None:
        plan.close()
        del x_obs, out
        torch.cuda.empty_cache()
        cpu = cpu_baseline(list(cpu_slices), mask, args.thresh_op, args.cpu_seconds, args.nslices)

    if rank == 0:
        its = K / seconds
        shape = (nil, nxl, args.nslices, round(args.missing, 2))
        baseline_tag = {(1024, 1024, 512, 0.8): " (BASELINE configs[2])", (512, 512, 256, 0.7): " (BASELINE configs[1])",
                        (64, 64, 128, 0.5): " (BASELINE configs[0])"}.get(shape, "")
        line = {
            "metric": "POCS iterations/s on the 1024x1024x512 cube" if (nil, nxl, args.nslices) == (1024, 1024, 512)
            else f"POCS iterations/s on the {nil}x{nxl}x{args.nslices} cube",
            "value": its,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": seconds * 1e3 / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "complex64 (f32 arithmetic)",
            "data": "synthetic: 6 complex plane waves + 1% Gaussian noise per slice (seeded), random trace mask",
            "config": {
                "workload": f"{nil}x{nxl}x{args.nslices} complex64 cube, {int(args.missing * 100)}% missing traces, FFT "
                            f"transform, {args.thresh_op} threshold, exponential decay, {K} iterations" + baseline_tag,
                "slices_per_gpu": n_local,
                "parallelism": f"slice axis in {world} contiguous block(s), one rank per GPU, no collective in the loop",
            },
            "slice_iterations_per_s": its * args.nslices,
            "interpolated_traces_per_s": float(np.count_nonzero(mask == 0)) / seconds,
            "device_ms_rank0": dev_ms,
            "gather_ms": gather_ms,
            "sparse_spectrum": {
                "nonzero_block_fraction": nz_fraction,
                "note": "8-column blocks of the thresholded spectrum that kept a coefficient (rank 0, mean over slices and "
                        "iterations); emptied blocks are not transformed back, stored or re-read -- exact. Data dependent: "
                        "dense_path_iterations_per_s is the rate with the shortcut off (P3D_NO_SPARSE=1), from rank 0's block",
                "dense_path_iterations_per_s": dense_its,
            },
            "real_cube": {
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
