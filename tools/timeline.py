"""Timeline of the last solve in a rocprofv3 rocpd database (--kernel-trace): every kernel's start relative to the Step-1 kernel's start,
its duration and whether it ran while Step 1 was running -- shows whether the constraint set-up really overlaps Step 1.
    python tools/timeline.py results.db [max_rows]"""
import sqlite3, sys
def main(db, max_rows=400):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    convs = [i for i, r in enumerate(rows) if "conv_" in r[0] and "kernel" in r[0]]
    last_end = rows[convs[-1]][2]
    # the Step-1 launches of the last solve: walk back while the gaps between conv launches are short
    first = convs[-1]
    for i in reversed(convs[:-1]):
        if rows[first][1] - rows[i][2] > 1e5: break   # > 0.1 ms apart: previous solve (the chunk launches of one Step 1 follow each other within microseconds)
        first = i
    t0, t1 = rows[first][1], last_end
    print("Step 1 of the last solve: %.3f ms (%d launches)" % ((t1 - t0) * 1e-6, sum(1 for i in convs if i >= first)))
    seq = [r for r in rows if r[1] >= t0 - 5e6]
    inside = sum(min(r[2], t1) - max(r[1], t0) for r in seq if "conv_" not in r[0] and r[1] < t1 and r[2] > t0)
    print("other kernels' time inside Step 1's span: %.3f ms" % (inside * 1e-6))
    n = 0
    for r in seq:
        if "conv_" in r[0] and n > 0 and r[1] > t0: pass
        nm = r[0].split("(")[0].replace("void shm::", "")[:70]
        print("%9.3f ms  %9.1f us  %s%s" % ((r[1] - t0) * 1e-6, (r[2] - r[1]) * 1e-3, "" if r[1] < t1 else "after: ", nm))
        n += 1
        if n >= max_rows: break
if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
