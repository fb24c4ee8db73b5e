"""SURVEY section 8(c)'s acceptance bar, literally: the error of the HIP path against the float64 oracle may be at most
FOUR TIMES the error of the fp32 restatement of the reference's own op order (oracle tier 2: the numpy / torch-CPU float32
evaluation, the stand-in for the TF1 CPU path) against the same float64 result, on the same inputs and the same measure.

    check(test, quantity, err_hip, err_f32)      asserts err_hip <= max(4 * err_f32, FLOOR) and records the margin

FLOOR = 4 * 2^-24: where the float32 restatement happens to be exact to below one rounding of the result (an output of one
or three channels, a contraction of a handful of terms) "four times its error" would demand more than float32 can hold.

With CAPE_PARITY_MARGINS=<file> every comparison appends one line (test, quantity, both errors, their ratio) to that file:
`tools/parity_margins.py` turns it into profiles/rNN_parity_margins.txt.  TEST INFRASTRUCTURE ONLY."""
import os

# CAPE_PARITY_FACTOR: only for the NON-default arithmetic legs of tests/test_gpu_knobs.py, where it is set next to the measured
# ratio that motivates it; the default path is always held to 4
FACTOR = float(os.environ.get("CAPE_PARITY_FACTOR", "4.0"))
FLOOR = 4.0 * 2.0 ** -24

RECORDS = []
COLLECTED = 0        # comparisons that were recorded WITHOUT being asserted (CAPE_PARITY_COLLECT=1)


def check(test, quantity, err_hip, err_f32, also_below=None, factor=FACTOR):
    err_hip, err_f32 = float(err_hip), float(err_f32)
    ratio = err_hip / max(err_f32, 2.0 ** -24)
    RECORDS.append((test, quantity, err_hip, err_f32, ratio))
    path = os.environ.get("CAPE_PARITY_MARGINS")
    if path:
        leg = os.environ.get("CAPE_PARITY_LEG", "default")          # which arithmetic the library ran (tests/test_gpu_knobs.py)
        with open(path, "a") as f:
            f.write("%s\t%s\t%s\t%.4e\t%.4e\t%.3f\n" % (leg, test, quantity, err_hip, err_f32, ratio))
    if os.environ.get("CAPE_PARITY_COLLECT") == "1":                 # measurement runs: record every margin, judge afterwards
        global COLLECTED                                             # (tests/conftest.py fails the SESSION loudly when this happened)
        COLLECTED += 1
        return ratio
    bar = max(factor * err_f32, FLOOR)
    assert err_hip <= bar, "%s / %s: HIP error %.3e vs float64 exceeds %g x the fp32 restatement's %.3e" % (test, quantity, err_hip, factor, err_f32)
    if also_below is not None:
        assert err_hip < also_below, "%s / %s: HIP error %.3e exceeds the absolute bar %.1e" % (test, quantity, err_hip, also_below)
    return ratio
