#!/usr/bin/env python3
"""One steady-state training step out of a rocprofv3 rocpd .db (kernel trace): finds the period of the
graph-replay dispatch sequence, then prints per-kernel totals of ONE step and the ordered dispatch list.
    python tools/rocpd_step_seq.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(s):
    s = re.sub(r'\(anonymous namespace\)::', '', s)
    s = re.sub(r'void ', '', s)
    s = re.sub(r'at::native::', '', s)
    return s


def tag(n):
    if n.startswith('Cijk'):
        return 'rocblas:' + re.search(r'MT\d+x\d+x\d+', n).group(0)
    m = re.match(r'(\w+)(<[^(]*>)?', n)
    base = m.group(1)
    if base in ('gconv_fwd_kernel', 'gemm_plain_kernel', 'gemm_split_kernel', 'gconv_dw_kernel', 'dw_plain_kernel', 'dw_packed_kernel'):
        return base.replace('_kernel', '') + (m.group(2) or '')
    f = re.search(r'(CUDAFunctor_add|CUDAFunctorOnSelf_add|MulFunctor|addcmul|NormTwo|sum_functor|FillFunctor|pow_tensor|'
                  r'leaky_relu_backward|leaky_relu|reciprocal|exp_kernel|clamp|direct_copy|gather|sqrt|DivFunctor)', n)
    return base[:26] + (':' + f.group(1) if f else '')


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,start,end from kernels order by start"))
    names = [r[0] for r in rows]
    n = len(names)
    best = None
    for p in range(60, 900):
        run = mx = end = 0
        for i in range(n - p):
            if names[i] == names[i + p]:
                run += 1
                if run > mx:
                    mx, end = run, i
            else:
                run = 0
        if mx >= 3 * p:
            best = (p, mx, end)
            break
    if best is None:
        raise SystemExit("no periodic dispatch sequence found")
    p, mx, end = best
    start = end - mx + 1
    seq = rows[start + p: start + 2 * p]
    span = (seq[-1][2] - seq[0][1]) / 1e3
    busy = sum(e - s for _, s, e in seq) / 1e3
    lines = ["# %s: %d dispatches per step, span %.1f us, kernel time %.1f us" % (db, p, span, busy)]
    agg = {}
    for nm, s, e in seq:
        a = agg.setdefault(tag(short(nm)), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    lines.append("%5s %10s %8s  kernel" % ("calls", "total_us", "avg_us"))
    for k, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%5d %10.1f %8.1f  %s" % (cnt, t, t / cnt, k))
    lines.append("# ordered dispatches (kernel us)")
    items = ["%s %.0f" % (tag(short(nm)), (e - s) / 1e3) for nm, s, e in seq]
    for i in range(0, len(items), 5):
        lines.append("%3d | " % i + " | ".join(items[i:i + 5]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
