#!/usr/bin/env python3
"""Per (kernel, grid size) averages of every counter of a rocprofv3 --pmc rocpd database -- the launches of one kernel
template at different layer shapes stay apart (tools/pmc_summary.py merges them).
    python tools/pmc_by_grid.py results.db [more.db ...] > table.json"""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\((?!anonymous).*', '', n)[:70]


def main(dbs):
    agg = collections.defaultdict(dict)
    for db in dbs:
        c = sqlite3.connect(db)
        for k, grid, wg, cn, v, n in c.execute(
                "select kernel_name, grid_size, workgroup_size, counter_name, sum(value), count(*) from counters_collection "
                "group by kernel_name, grid_size, workgroup_size, counter_name"):
            agg["%s grid=%d wg=%d" % (short(k), grid // max(wg, 1), wg)][cn] = round(v / n, 1)
            agg["%s grid=%d wg=%d" % (short(k), grid // max(wg, 1), wg)]["dispatches"] = n
    json.dump(agg, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
