#!/usr/bin/env python3
"""Compare the per-shape checksums and timings printed by tools/ubench/gemm_bench / dw_bench runs.
    python tools/ab_compare.py reference.txt candidate.txt [candidate2.txt ...]
Every line "<shape>  <us> us <TF> TF  sum <s> abs <a>" of the reference is matched with the same shape in the
candidates; the absolute-value checksum must agree to 2e-6 relative (fp32 summation-order noise), the signed one to
2e-6 of the absolute checksum.  Exit status 1 on any mismatch."""
import re
import sys

PAT = re.compile(r"^(?P<shape>.*?)\s+(?P<us>[0-9.]+) us\s+(?P<tf>[0-9.]+) TF\s+sum (?P<sum>\S+) abs (?P<abs>\S+)")


def parse(path):
    out = []
    for line in open(path):
        m = PAT.match(line.rstrip())
        if m:
            out.append((" ".join(m.group("shape").split()), float(m.group("us")), float(m.group("sum")), float(m.group("abs"))))
    return out


def main(ref_path, *cand_paths):
    ref = parse(ref_path)
    bad = 0
    cands = [{r[0]: r for r in parse(p)} for p in cand_paths]
    print("%-34s %9s" % ("shape", "ref us") + "".join(" %12s" % p.split("/")[-1][:12] for p in cand_paths))
    for i, (shape, us, s, a) in enumerate(ref):
        row = "%-34s %9.1f" % (shape, us)
        for c in cands:
            if shape not in c:
                row += " %12s" % "MISSING"
                bad += 1
                continue
            _, cus, cs, ca = c[shape]
            ok = abs(ca - a) <= 2e-6 * max(abs(a), 1e-30) and abs(cs - s) <= 2e-6 * max(abs(a), 1e-30)
            row += " %8.1f %3s" % (cus, "ok" if ok else "BAD")
            bad += 0 if ok else 1
        print(row)
    tot = sum(r[1] for r in ref)
    print("%-34s %9.1f" % ("TOTAL", tot) + "".join(" %8.1f    " % sum(c[r[0]][1] for r in ref if r[0] in c) for c in cands))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
