#!/usr/bin/env python3
"""Format the margins that tests/parity_bar.py recorded (CAPE_PARITY_MARGINS=<tsv>) into the committed summary:
    python tools/parity_margins.py gpurun_out/r05_parity_margins.tsv > profiles/r05_parity_margins.txt
One line per comparison: arithmetic leg, test, quantity, error of the HIP path against float64, error of the fp32 restatement
against float64, their ratio (SURVEY 8(c) allows 4)."""
import sys


def main(path):
    rows = [l.rstrip("\n").split("\t") for l in open(path) if l.strip()]
    rows = [(r[0], r[1], r[2], float(r[3]), float(r[4]), float(r[5])) for r in rows if len(r) == 6]
    seen, uniq = set(), []
    for r in rows:                       # a test that ran twice (plain run + knob-matrix subprocess) keeps its last record
        seen.discard((r[0], r[1], r[2]))
    for r in reversed(rows):
        if (r[0], r[1], r[2]) not in seen:
            seen.add((r[0], r[1], r[2]))
            uniq.append(r)
    uniq.reverse()
    print("# SURVEY 8(c): err(HIP vs fp64) <= 4 x err(fp32 restatement vs fp64); %d comparisons, floor 4 * 2^-24 = 2.4e-07" % len(uniq))
    for leg in sorted(set(r[0] for r in uniq)):
        sel = [r for r in uniq if r[0] == leg]
        worst = max(sel, key=lambda r: r[5] if r[3] > 2.4e-7 else 0.0)
        print("# leg %-18s %4d comparisons, largest ratio above the floor %.2f (%s / %s)" % (leg, len(sel), worst[5], worst[1], worst[2]))
    print("%-18s %-58s %-36s %10s %10s %7s" % ("leg", "test", "quantity", "err_hip", "err_fp32", "ratio"))
    for r in uniq:
        print("%-18s %-58s %-36s %10.3e %10.3e %7.2f" % r)


if __name__ == "__main__":
    main(sys.argv[1])
