"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2)
into the per-kernel statistics table that the CSV writer would have produced: calls, total, avg, min, max, %."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                     "max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-100s %8s %14s %12s %12s %12s %6s %5s %5s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "sgpr", "lds_B")]
    for r in rows:
        name = r[0] if len(r[0]) <= 100 else r[0][:97] + "..."
        lines.append("%-100s %8d %14d %12.0f %12d %12d %6.2f %5d %5d %8d" % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8]))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
