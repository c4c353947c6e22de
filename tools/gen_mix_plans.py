#!/usr/bin/env python3
"""Compile-time plans of the mixed-radix register engine (csrc/p3d_mix.hpp) for 7-smooth line lengths that are not powers of two:
N = R0 * R1 * ..., pass p with radix R_p and B_p butterflies per thread (PPT_p = R_p B_p points per thread, N / PPT_p threads per line in that
pass).  Writes csrc/p3d_mix_plans.inc:  X(N, COLT, ROWLB, NPASS, R0, B0, R1, B1, R2, B2, R3, B3).

    python tools/gen_mix_plans.py [--min 96] [--max 4096] [--list] > pseudo-3d-interpolation_amd/csrc/p3d_mix_plans.inc
"""
import argparse
import itertools
import sys


PRIMES13 = (2, 3, 5, 7, 11, 13)   # radices may contain 11 and 13 (in-register butterflies of the paired form); lengths with such factors from --min13 on


def smooth(n, ps=(2, 3, 5, 7)):
    for p in ps:
        while n % p == 0:
            n //= p
    return n == 1


def dft_cost(r):
    """rough real-operation count of the in-register DFT of size r (recursive Cooley-Tukey on 2/3/4/5/7 bases)."""
    base = {1: 0, 2: 4, 3: 16, 4: 16, 5: 40, 7: 84, 11: 230, 13: 330}
    if r in base:
        return base[r]
    for a in (4, 2, 3, 5, 7, 11, 13):
        if r % a == 0:
            b = r // a
            return b * dft_cost(a) + a * dft_cost(b) + 6 * (a - 1) * (b - 1)
    raise ValueError(r)


MAX_RADIX = 20


def factorizations(n, allowed, maxlen):
    """multisets of radices (from `allowed`, descending) whose product is n, at most maxlen long."""
    out = []

    def rec(rem, start, cur):
        if rem == 1:
            out.append(tuple(cur))
            return
        if len(cur) == maxlen:
            return
        for i in range(start, len(allowed)):
            r = allowed[i]
            if rem % r == 0:
                rec(rem // r, i, cur + [r])
    rec(n, 0, [])
    return out


PPT_LO, PPT_HI, PPT_TARGET, ELEM = 11, 32, 20, 8   # float32 engine; --f64: 6, 16, 8 and 16-byte elements


def splits(n, r):
    """(B, PPT) choices of a pass of radix r: PPT = r B divides n, PPT_LO <= PPT <= PPT_HI where possible (else the nearest)."""
    cands = [(b, r * b) for b in range(1, 33) if n % (r * b) == 0 and r * b <= 64]
    good = [c for c in cands if PPT_LO <= c[1] <= PPT_HI and (r <= 20 or c[0] == 1)]
    return good or sorted(cands, key=lambda c: abs(c[1] - PPT_TARGET))[:2]


def best_plan(n, max_radix=None):
    """(score, colt, ((R0, B0), (R1, B1), ...)) -- forward order."""
    best = None
    allowed = sorted([d for d in range(2, (max_radix or MAX_RADIX) + 1) if smooth(d, PRIMES13) and n % d == 0], reverse=True)
    for fac in factorizations(n, allowed, 4):
        if len(fac) == 1 and n > 32:
            continue
        per_pass = [splits(n, r) for r in fac]
        for combo in itertools.product(*per_pass):
            ppts = [c[1] for c in combo]
            tpls = [n // p for p in ppts]
            tmax = max(tpls)
            if tmax < 4 or 2 * tmax > 1024:   # (a row pair is the smallest workgroup of the row pass)
                continue
            line = n * 1.1
            # Column tile: 8 columns (a whole 64-byte column block per row) where two workgroups' tiles share a CU's LDS (<= 72 KiB each).  LONG
            # columns: the widest tile that fits ONE workgroup per CU (image + tables <= 154 KiB) if that workgroup has >= 600 threads to cover its
            # own latencies -- narrower tiles move 32- or 16-byte pieces of every row (measured: 2000-point columns 2.09 -> 1.27 ms per 128 x 1500
            # columns with 8 instead of 4 per tile, 1600 points 1.00 -> 0.79, 1500 points 1.62 -> 1.32; 1200 points, 480 threads: 0.78 -> 0.87) --,
            # else the widest one of which two fit.
            worst = 1 + max([1 / r for r in fac if r % 2 == 0 and r > 2] or [0])   # padding of the LDS image: one slot per (even) first radix
            tables = 1.3 * n * ELEM if ELEM == 8 else 0   # (the double-precision passes read their tables from memory)
            small = next((t for t in (8, 4, 2, 1) if t * tmax <= 1024 and t * line * ELEM <= (72 if ELEM == 8 else 78) * 1024), 0)
            big = next((t for t in (8, 4, 2) if t * tmax <= 1024 and t * tmax >= (600 if small >= 4 else 450) and t * n * ELEM * worst + tables <= 154 * 1024), 0)
            colt = big if big > small else small
            if not colt:
                continue
            regs = (2 * max(ppts) + 2 * max(fac) + 24) * ELEM // 8
            idle = sum(1 - t / tmax for t in tpls)
            score = (len(fac) * 1000 + sum(dft_cost(r) / r for r in fac) * 4 + sum(abs(p - PPT_TARGET) for p in ppts) * 3 + idle * 60
                     + (200 if regs > (128 if ELEM == 8 else 256) else 0) + (150 if colt < 8 and n <= 1100 else 0) + (0 if (colt * tmax) % 64 == 0 else 10))
            pairs = list(zip(fac, [c[0] for c in combo]))
            # forward order: an odd radix first where there is one (its scatter has an odd stride without padding), the largest radix last
            # (the last pass scatters nothing; it is the inverse transform's first)
            orders = set(itertools.permutations(pairs))
            order = min(orders, key=lambda o: (o[0][0] % 2 == 0, -o[-1][0], o))
            cand = (score, colt, order)
            if best is None or cand < best:
                best = cand
    return best


def row_lines(n, tmax):
    """rows per workgroup of the row pass: an even number (rows are worked on in pairs), ~256 threads, at most ~44 KiB of line images (three
    workgroups per CU beside the twiddle tables), the candidate that fills its wavefronts best.  (Larger workgroups would amortise the copy of
    the twiddle tables -- ~1.25 N entries per workgroup -- but leave one or two workgroups per CU, whose phases then no longer cover each other.)
    --f64: any number of rows (they are not paired there), 16-byte elements."""
    if ELEM == 16:   # double precision: small workgroups, many of them per CU (1024-point rows: one row per workgroup 0.67 ms, two 0.70, four 0.70)
        return max(1, -(-128 // tmax))
    best = None
    step = 2 if ELEM == 8 else 1
    for lb in range(step, 65, step):
        thr = lb * tmax
        if thr > 1024 or (lb > step and (lb * n * ELEM * 1.1 > (44 if ELEM == 8 else 78) * 1024 or thr > 320)):
            break
        fill = thr / (64 * -(-thr // 64))
        cand = (-(fill > 0.9), -thr if fill > 0.9 else -round(fill, 2), -lb)
        if best is None or cand < best[0]:
            best = (cand, lb)
    return best[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=96)
    ap.add_argument("--max", type=int, default=4096)
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--only", type=int, nargs="*")
    ap.add_argument("--parts", type=int, default=8, help="translation units the instantiations are spread over (#if P3D_MIX_PART == k)")
    ap.add_argument("--min13", type=int, default=480, help="lengths with prime factors 11 and 13 from this one on (below: 7-smooth lengths only)")
    ap.add_argument("--two-pass-below", type=int, default=1024,
                    help="lengths up to this one may use radices up to 32: two-pass plans, one exchange per transform (measured: 960-point columns 0.41 -> 0.24 ms, "
                         "900 x 900 1.05 -> 0.91 ms per iteration of 128 slices; 768-point rows lose 8 %%)")
    ap.add_argument("--f64", action="store_true", help="plans of the double-precision passes (csrc/p3d_mix64_plans.inc): 16-byte elements, 8 ... 20 points per "
                    "thread, powers of two included; --only / the default list of lengths")
    ap.add_argument("--all-smooth", action="store_true", help="with --f64: plans for every 7-smooth length min ... max, not only the default list")
    ap.add_argument("--roots", action="store_true", help="write the root tables Roots<R> (csrc/p3d_mix_roots.inc) instead of the plan list")
    args = ap.parse_args()
    if args.roots:
        import math
        print("// generated by tools/gen_mix_plans.py --roots -- do not edit.  cos / sin of 2 pi q / R, evaluated in double precision, rounded once")
        for r in range(2, 33):
            if not smooth(r, PRIMES13):
                continue
            def fmt(x):
                return "%.9ef" % (0.0 if abs(x) < 1e-15 else x)
            cs = ", ".join(fmt(math.cos(2 * math.pi * q / r)) for q in range(r))
            sn = ", ".join(fmt(math.sin(2 * math.pi * q / r)) for q in range(r))
            print(f"template <> struct Roots<{r}> {{\n    static constexpr float c[{r}] = {{{cs}}};\n    static constexpr float s[{r}] = {{{sn}}};\n}};")
            def fmtd(x):
                return "%.17e" % (0.0 if abs(x) < 1e-17 else x)
            csd = ", ".join(fmtd(math.cos(2 * math.pi * q / r)) for q in range(r))
            snd = ", ".join(fmtd(math.sin(2 * math.pi * q / r)) for q in range(r))
            print(f"template <> struct RootsD<{r}> {{\n    static constexpr double c[{r}] = {{{csd}}};\n    static constexpr double s[{r}] = {{{snd}}};\n}};")
        return
    if args.f64:
        global PPT_LO, PPT_HI, PPT_TARGET, ELEM
        PPT_LO, PPT_HI, PPT_TARGET, ELEM = 6, 16, 8, 16   # (measured on 1024-point lines: 8 points per thread 0.70 ms per iteration, 16 points 1.0-1.2 ms)
        args.two_pass_below = 0
    F64_LENGTHS = [64, 128, 256, 512, 1024, 2048, 4096, 500, 600, 720, 768, 800, 900, 960, 1000, 1200, 1280, 1440, 1500, 1536, 1600, 1800, 1920, 2000, 2400, 3000, 3072]
    if args.f64 and args.all_smooth:   # every 7-smooth length of the float32 engine's range, and the powers of two
        F64_LENGTHS = sorted(set(F64_LENGTHS) | {n for n in range(args.min, args.max + 1) if smooth(n)})
    lengths = args.only or (sorted(F64_LENGTHS) if args.f64 else
                            [n for n in range(args.min, args.max + 1) if n & (n - 1) and (smooth(n) or (n >= args.min13 and smooth(n, PRIMES13)))])
    rows = []
    forced = {}   # experiments: P3D_GEN_FORCE="1000:10x2,10x2,10x2;768:8x3,8x3,12x2" (forward order, radix x butterflies per thread)
    for item in filter(None, __import__("os").environ.get("P3D_GEN_FORCE", "").split(";")):
        nn, plan = item.split(":")
        forced[int(nn)] = tuple(tuple(int(v) for v in ps.split("x")) for ps in plan.split(","))
    for n in lengths:
        if n in forced:
            order = forced[n]
            tmax = max(n // (r * nb) for r, nb in order)
            colt = int(__import__("os").environ.get("P3D_GEN_FORCE_COLT", "0")) or next(t for t in (8, 4, 2, 1) if t * tmax <= 1024 and t * n * 1.1 * ELEM <= 76 * 1024)
            rows.append((n, colt, int(__import__("os").environ.get("P3D_GEN_FORCE_LB", "0")) or row_lines(n, tmax), order))
            continue
        b = best_plan(n, 32 if n <= args.two_pass_below else None)
        if b is None:
            continue
        _, colt, order = b
        tmax = max(n // (r * nb) for r, nb in order)
        rows.append((n, colt, row_lines(n, tmax), order))
    if args.list:
        for n, colt, lb, order in rows:
            print(f"{n:5d}  col tile {colt}  rows/wg {lb:2d}  passes " + "  ".join(f"{r}x{nb} ({n // (r * nb)} thr)" for r, nb in order))
        return
    print("// generated by tools/gen_mix_plans.py -- do not edit.  X(N, COLT, ROWLB, NPASS, R0, B0, R1, B1, R2, B2, R3, B3): forward pass p has radix R_p")
    print(f"// and B_p butterflies per thread.  The instantiations are spread over {args.parts} translation units (heaviest lengths dealt round-robin).")
    order_by_cost = sorted(rows, key=lambda r: -r[0] * len(r[3]))
    part = {r[0]: i % args.parts for i, r in enumerate(order_by_cost)}
    print(f"#define P3D_MIX_PARTS {args.parts}")
    for k in range(args.parts):
        print(f"#if P3D_MIX_PART < 0 || P3D_MIX_PART == {k}")
        for n, colt, lb, order in rows:
            if part[n] == k:
                rb = [x for pr in order for x in pr] + [1, 1] * (4 - len(order))
                print(f"X({n}, {colt}, {lb}, {len(order)}, {', '.join(map(str, rb))})")
        print("#endif")


if __name__ == "__main__":
    main()
