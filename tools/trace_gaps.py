"""Per-kernel timeline of the last solve in a rocprofv3 rocpd database: durations and the idle gaps between consecutive
kernels on the device (launch-bound loops show up as gaps comparable to the kernel durations)."""
import sqlite3
import sys
from collections import defaultdict


def main(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # last conv kernel marks the start of the last solve
    last = max(i for i, r in enumerate(rows) if "conv_normalize" in r[0])
    seq = rows[last:]
    t0 = seq[0][1]
    busy = defaultdict(float)
    cnt = defaultdict(int)
    gaps = 0.0
    for a, b in zip(seq[:-1], seq[1:]):
        gaps += max(0, b[1] - a[2])
    for r in seq:
        nm = r[0].split("(")[0][-60:]
        busy[nm] += (r[2] - r[1]) * 1e-3
        cnt[nm] += 1
    span = (seq[-1][2] - t0) * 1e-3
    print("columns:", cols)
    print("kernels in last solve: %d, span %.1f us, sum of gaps %.1f us" % (len(seq), span, gaps * 1e-3))
    for k, v in sorted(busy.items(), key=lambda x: -x[1]):
        print("%-62s calls %5d total %10.1f us avg %8.2f us" % (k, cnt[k], v, v / cnt[k]))
    # gap histogram after the conv kernel
    g = sorted(max(0, b[1] - a[2]) * 1e-3 for a, b in zip(seq[1:-1], seq[2:]))
    if g:
        print("gap us: median %.2f p90 %.2f max %.2f n=%d" % (g[len(g) // 2], g[int(len(g) * 0.9)], g[-1], len(g)))


if __name__ == "__main__":
    main(sys.argv[1])
