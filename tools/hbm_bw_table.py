#!/usr/bin/env python3
"""Effective HBM bandwidth per kernel: HBM bytes per dispatch from the PMC summary (2 x FETCH_SIZE + WRITE_SIZE) divided
by the average kernel duration of the rocprofv3 kernel-trace summary of the same build.
    python tools/hbm_bw_table.py profiles/r01_pmc_summary.json profiles/r01_bench_kernel_stats.txt"""
import json
import re
import sys


def main(pmc_path, stats_path):
    pmc = json.load(open(pmc_path))
    stats = {}
    for line in open(stats_path):
        m = re.match(r'(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%', line)
        if m:
            name = m.group(1).replace('(anonymous namespace)::', '').replace('void ', '').strip()
            stats[re.sub(r'\((?!anonymous).*', '', name)] = (int(m.group(2)), float(m.group(4)))
    rows = []
    for k, v in pmc.items():
        if 'hbm_bytes_per_dispatch' not in v:
            continue
        s = next((val for name, val in stats.items() if name.startswith(k[:60]) or k.startswith(name[:60])), None)
        if s:
            rows.append((v['hbm_bytes_per_dispatch'] / s[1] / 1e6, k, s[1], v['hbm_bytes_per_dispatch'] / 1e6, s[0]))
    print("# %s + %s; peak 8 TB/s quoted, ~6.3 TB/s achievable copy rate" % (pmc_path, stats_path))
    print("%-52s %9s %10s %8s" % ("kernel", "avg us", "HBM MB", "TB/s"))
    for bw, k, us, mb, calls in sorted(rows, reverse=True):
        print("%-52s %9.1f %10.1f %8.2f" % (k[:52], us, mb, bw))


if __name__ == "__main__":
    main(*sys.argv[1:3])
