#!/usr/bin/env python3
"""Dump the per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd .db (what
`rocprofv3 --kernel-trace --stats` collected) as text -- the form committed under profiles/."""
import sqlite3
import sys


def main(db, out=None, top=45):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# source: %s  (rocprofv3 --kernel-trace --stats; durations in microseconds)" % db,
             "%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in rows[:top]:
        lines.append("%-100s %8d %14.1f %12.2f %6.2f%%" % (name[:100], calls, total, avg, pct))
    lines.append("# %d distinct kernels, %d dispatches, %.1f us total kernel time"
                 % (len(rows), sum(r[1] for r in rows), sum(r[2] for r in rows)))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
