#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc rocpd database: per kernel, the average of each counter per dispatch."""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    return re.sub(r'\((?!anonymous).*', '', n)[:80]


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                     "group by kernel_name, counter_name")
    agg = collections.defaultdict(dict)
    for k, cn, v, n in rows:
        agg[short(k)][cn] = dict(avg_per_dispatch=v / n, dispatches=n)
    keep = {k: v for k, v in agg.items() if any(t in k for t in ('gconv', 'gemm_', 'dw_', 'spmm', 'bwd_prep', 'edge', 'vert_', 'fc_', 'cond_coef', 'momentum', 'gradnorm', 'gn_', 'cheb'))}
    json.dump(keep, open(out, 'w'), indent=1, sort_keys=True)
    print("wrote", out, len(keep), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
