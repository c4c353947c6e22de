"""CPU only (oracle).  complex64 cubes + hard threshold + 'inverse_proportional': why single slices come out 1e-2 ... 1e-1 apart between
two evaluations of the SAME reference code (NumPy complex64 vs complex128; the device is a third), and why that is the reference's
conditioning and not a defect (the float32-cube variant of the story is tools/diag_real_tie.py).

The model's first threshold is meant to be max|X| itself (POCS.py:251-274):
    a = n^q (hi - lo) / (n^q - 1),  b = (n^q lo - hi) / (n^q - 1),  tau_1 = a / 1^q + b        (hi = max|X|, lo = min|X|)
In exact arithmetic tau_1 = hi.  In floating point it is hi, or a little above, or a little below -- decided by the last bits of hi
and lo: by an ulp of double for a complex128 spectrum, by ~1e-7 relative for a complex64 one (hi, lo, a, b are float32 scalars there).
The hard operator then zeroes where |X| < tau_1 (threshold_operator.py:110-112, np.less in float64 because tau is a float64 scalar):
    tau_1 <= hi  ->  the largest coefficient of the slice survives the first iteration,
    tau_1 >  hi  ->  it is zeroed (and with it most of the slice's energy for that iteration).
Which of the two happens is a coin flip on rounding; two evaluations whose spectra differ in the last bit (complex64 / complex128
pocketfft, the device's FFT) flip it independently, and a flipped slice differs by 1e-2 ... 1e-1 after a few iterations.

Prints, for a handful of slices, the coin under NumPy's complex64 and complex128 spectra and the rel-L2 between the two runs of the
oracle (= the reference's algorithm; pinned bit-for-bit against the reference for this model by tests/golden)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pocs_oracle as orc  # noqa: E402


def coin(x, niter):
    """What the reference's own arithmetic does to the largest coefficient in iteration 1 (schedule by the oracle's restatement of
    get_threshold_decay -- for a complex64 spectrum hi, lo, a and b are float32 scalars and tau = a / k^q + b is float64: tau_1 is
    then off max|X| by float32 rounding, ~1e-7 relative, in either direction)."""
    X = np.fft.fft2(x)
    tau = orc.threshold_schedule("inverse_proportional", niter, "FFT", 0.99, 1e-2, X, "values")
    mag = np.abs(X)
    kept = int(np.count_nonzero(~np.less(mag, tau[0])))
    rel = (float(tau[0]) - float(mag.max())) / float(mag.max())
    return f"{'kept  ' if kept else 'ZEROED'} (tau_1/max|X| - 1 = {rel:+.1e})"


def main():
    kw = dict(niter=6, thresh_op="hard", thresh_model="inverse_proportional", eps=0.0, alpha=1.0, p_max=0.99, p_min=1e-2)
    print("shape      seed  largest coefficient, complex64 run           largest coefficient, complex128 run          rel-L2 between the two runs")
    agree = differ = 0
    for nil, nxl in ((32, 32), (60, 100), (128, 64), (286, 100)):
        mask = orc.synthetic_mask(nil, nxl, 0.5)
        for seed in range(6):
            x = (orc.synthetic_slice(nil, nxl, seed) * mask).astype(np.complex64)
            c32, c64 = coin(x, kw["niter"]), coin(x.astype(np.complex128), kw["niter"])
            a = orc.pocs_slice(x, mask, **kw)
            b = orc.pocs_slice(x.astype(np.complex128), mask, **kw)
            err = float(np.linalg.norm(a - b) / np.linalg.norm(b))
            same = c32.split()[0] == c64.split()[0]
            agree += same
            differ += not same
            print(f"{nil:4d}x{nxl:<5d} {seed:3d}   {c32:42s}  {c64:42s}  {err:.2e}{'' if same else '   <-- the coin fell differently'}")
    print(f"coins agree on {agree} slices (runs then agree to rounding / ordinary flips), differ on {differ} (1e-2 ... 1e-1)")


if __name__ == "__main__":
    main()
