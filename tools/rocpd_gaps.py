import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,stream_id,queue_id from kernels order by start"))
rows = rows[-1400:]
prev_end = None
out = []
for nm, s, e, st, q in rows:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    short = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', nm)[:50]
    if gap > 30 or 'ccl' in nm.lower() or 'AllReduce' in nm or 'copyBuffer' in nm and (e - s) > 20000:
        out.append("gap %7.1f us before %-50s dur %7.1f us stream %s queue %s" % (gap, short, (e - s) / 1e3, st, q))
    prev_end = max(prev_end or 0, e)
print("\n".join(out[-40:]))
