"""Register spills of the compiled kernels, read from the code objects on the CPU (no GPU needed).

The hot kernels sit right at their register budget (128 VGPRs for 16 wavefronts per CU); an innocent-looking edit that makes the compiler spill a
register or two costs a few per cent and shows up in no functional test -- the early-exit rework of round 3 did exactly that to `row_pipe64_kernel<1024>`
(2.3 % on the headline cube, found only by an A/B run).  This test pins the set of kernels that use scratch memory at all: a NEW name in it fails the test
(fix the spill, or add the kernel here with the reason), a name that disappears should be taken off the list."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "pseudo-3d-interpolation_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels that are allowed to touch scratch, and why
KNOWN_SCRATCH = {
    r"resident_kernel<128, 128, \d>": "128 x 128 slices are NOT routed to the single-kernel path (p3d_resident.hpp: RESIDENT_MAX_POINTS); kept for the record of why",
    r"row_pipe_kernel<\d+, true, \d, (true|false), true>": "generic per-lane persistent row pass (APOCS with the early exit on short rows, odd row counts): 168-register budget, not a default path",
    r"row_real_kernel<(256|512|1024), 1, false>": "row-pair steady state WITHOUT the sparse shortcut (P3D_NO_SPARSE / dense spectra): the sparse form, the default, is clean",
    r"row_real_kernel<(2048|4096), \d, (true|false)>": "row pairs at 2048 / 4096 samples: no VGPR spills since their register budget follows their LDS-bound occupancy; 80 B of scratch for spilled SCALAR registers (the mask / base tables of two units)",
    r"row_pipe64_kernel<128, 1, false, 0, false, false>": "float32 cube through the complex passes at 128-sample rows, dense form",
    r"col_pipe_kernel<256, 32, \d, false>": "persistent column pass at 256 points (an experiment behind P3D_FORCE_COLPIPE)",
    r"flex_col_kernel<false, false>": "SGPR spills of the run-time radix dispatch (one tile per workgroup form)",
    r"chirp_col_kernel<2048, 0>": "chirp-z column iteration at 2048 points: 3 registers over the 128 of a 512-thread workgroup pair",
    r"mix64::(col|shear_col|shear_col_pair)_kernel<mix::MixPlan<\d+, (8|4|2), .*> >": "double-precision column tiles on the register engine (p3d_mix64.hip; shear_col: the SHEARLET "
    "loop's column pass, same tile): whole 64-byte-per-row tiles (or the widest that fits) need 600-1024 threads, i.e. 128 VGPRs, and the last pass of complex128 "
    "butterflies spills 4-86 of them; the narrower tile that does not spill measured SLOWER (1024-point columns: 0.74 vs 0.71 ms per iteration of 32 slices, "
    "tools/mix64_try_plans.sh), and every length is 1.8-4 x faster than on the LDS-image passes (tools/f64_sweep.sh), so the spill is the accepted price",
    r"mix64::gather_row_kernel<mix::MixPlan<(2800|3200|3920|4000|4032), .*> >": "the SHEARLET loop's gather pass at the longest rows: 16-20 points and as many accumulators "
    "of complex128 per thread, 2-12 registers over 256",
}


def _kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        fb, co = os.path.join(d, "f.fatbin"), os.path.join(d, "f.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
    return out


@pytest.mark.skipif(not glob.glob(os.path.join(BUILD, "*.o")) or not os.path.exists(f"{LLVM}/clang-offload-bundler") or not shutil.which("c++filt"),
                    reason="needs the object files of the library build (python -c 'import __graft_entry__ as g; g.build()') and the ROCm LLVM tools")
def test_no_new_kernel_spills_registers():
    offenders, seen_known, nkernels, dirty = [], set(), 0, []
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        for mangled, res in _kernels(obj).items():
            nkernels += 1
            if res["vgpr_spill_count"] == 0 and res["private_segment_fixed_size"] == 0:
                continue
            name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("p3d::", "")
            dirty.append(short)
            hit = [pat for pat in KNOWN_SCRATCH if re.fullmatch(pat, short)]
            if hit:
                seen_known.add(hit[0])
            else:
                offenders.append((os.path.basename(obj), short, res))
    assert nkernels > 500                                   # (the whole library was looked at)
    assert not offenders, "kernels that spill registers / use scratch and are not on the list: %r" % (offenders,)
    stale = set(KNOWN_SCRATCH) - seen_known
    assert not stale, "no kernel matches these entries of KNOWN_SCRATCH any more (take them off the list): %r" % (sorted(stale),)
    # the kernels of the metric's configuration and of the other BASELINE configurations, by name: every instantiation of them is clean
    hot = ("row_pipe32_kernel", "row_pipe64_kernel<1024", "col_kernel<1024, 8", "row_pipe64_kernel<512", "col_kernel<512, 16", "row_real_kernel<1024, 1, true", "col_shear_pair_kernel<2048",
           "row_kernel<1024, 3", "row_kernel<1024, 4", "wfuse1_kernel", "wcoarse_kernel", "dwt2_tile_kernel", "idwt2_tile_kernel", "chirp_row_kernel",
           "col64_kernel", "row64_kernel")
    assert not [n for n in dirty if n.startswith(hot)], [n for n in dirty if n.startswith(hot)]


# kernels whose speed rests on a number of workgroups per CU that their LDS image allows and their registers must not take away again:
# (pattern of the demangled name, threads per workgroup, workgroups per CU) -> at most 512 / (wavefronts per SIMD) registers, in steps of 8
OCCUPANCY = [
    (r"wfuse1_kernel<float, 32, (4|8), (true|false)>", 512, 3),            # three 44-KiB tiles per CU: 24 wavefronts (profiles/r04_wavelet_workgroup_size.txt)
    (r"wfuse1_kernel<p3d::c32, 32, (4|8), (true|false)>", 1024, 1),
    (r"dwt2_tile_kernel<float, 32, (4|8)>", 512, 4),
    (r"idwt2_tile_kernel<float, 32, (4|8)>", 512, 4),
    (r"row_pipe32_kernel<.*>", 512, 1),
    (r"col_kernel<1024, 8, \d+>", 512, 2),
    (r"col64_kernel<\d, (true|false)>", 512, 2),
    (r"row64_kernel<\d, (true|false)>", 512, 2),
]


@pytest.mark.skipif(not glob.glob(os.path.join(BUILD, "*.o")) or not os.path.exists(f"{LLVM}/clang-offload-bundler") or not shutil.which("c++filt"),
                    reason="needs the object files of the library build and the ROCm LLVM tools")
def test_register_budgets_of_the_occupancy_bound_kernels():
    seen = set()
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        if os.path.basename(obj) not in ("wavelet.o", "f64.o", "inst_1024.o"):
            continue
        for mangled, res in _kernels(obj).items():
            name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
            short = short.replace("p3d::row", "row").replace("p3d::col", "col")
            for pat, threads, per_cu in OCCUPANCY:
                if re.fullmatch(pat, short):
                    seen.add(pat)
                    waves_per_simd = threads // 64 * per_cu / 4
                    budget = int(512 // waves_per_simd) // 8 * 8
                    assert res["vgpr_count"] <= min(budget, 512), (short, res["vgpr_count"], budget)
    assert seen == {pat for pat, _, _ in OCCUPANCY}, sorted({pat for pat, _, _ in OCCUPANCY} - seen)
