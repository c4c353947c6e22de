#!/usr/bin/env python3
"""Counter budget of ONE `rocprofv3 --pmc` pass on gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

    block   slots per pass   cost of the derived counters
    SQ      8                1 each
    TCC     4                FETCH_SIZE 3, WRITE_SIZE 2, every other TCC_* 1 (the `_sum` forms too: one slot, summed over channels)
    GRBM    2                1 each

    python3 tools/pmc_slots.py FETCH_SIZE                      -> exit 0, prints the pass
    python3 tools/pmc_slots.py FETCH_SIZE WRITE_SIZE           -> exit 2 (5 of 4 TCC slots), prints the split into passes
    python3 tools/pmc_slots.py --split FETCH_SIZE WRITE_SIZE   -> exit 0, one line per pass that fits

A request beyond the budget is not refused cleanly by rocprofv3 on this image: profiles/r03_pmc_counter_budget.txt (abort with
"error code 38: Request exceeds the capabilities of the hardware to collect" at the first dispatch in one run, a silent hang until the
box's silence guard in another).  tools/pmc_kernels.sh and tools/profile.sh pass every counter list through this check first.
"""
import sys

SLOTS = {"SQ": 8, "TCC": 4, "GRBM": 2}
COST = {"FETCH_SIZE": ("TCC", 3), "WRITE_SIZE": ("TCC", 2)}


def cost(name):
    if name in COST:
        return COST[name]
    for block in SLOTS:
        if name.startswith(block + "_"):
            return block, 1
    if name.startswith(("TCP_", "TA_", "TD_", "SPI_", "CPC_", "CPF_", "GDS_")):
        return name.split("_")[0], 1
    raise SystemExit(f"pmc_slots: unknown counter {name!r} (add its block and cost to tools/pmc_slots.py before using it)")


def used(counters):
    tot = {}
    for c in counters:
        block, n = cost(c)
        tot[block] = tot.get(block, 0) + n
    return tot


def fits(counters):
    return all(n <= SLOTS.get(b, 4) for b, n in used(counters).items())


def split(counters):
    """Greedy first-fit: passes in the order given, a counter opens a new pass when it does not fit any earlier one."""
    passes = []
    for c in counters:
        for p in passes:
            if fits(p + [c]):
                p.append(c)
                break
        else:
            passes.append([c])
    return passes


def main(argv):
    do_split = "--split" in argv
    counters = [a for a in argv if not a.startswith("--")]
    if not counters:
        raise SystemExit(__doc__)
    if fits(counters):
        print(" ".join(counters))
        return 0
    passes = split(counters)
    if do_split:
        for p in passes:
            print(" ".join(p))
        return 0
    over = {b: f"{n} of {SLOTS.get(b, 4)}" for b, n in used(counters).items() if n > SLOTS.get(b, 4)}
    sys.stderr.write(f"pmc_slots: {' '.join(counters)} does not fit one pass ({over} slots); run these passes instead:\n")
    for p in passes:
        sys.stderr.write("    " + " ".join(p) + "\n")
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
