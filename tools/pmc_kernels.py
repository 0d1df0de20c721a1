"""Average PMC counter values per kernel from a rocprofv3 --pmc rocpd database."""
import sqlite3
import sys
from collections import defaultdict


def main(db, pat=""):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print("tables:", tabs)
        return
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view).fetchall()]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, counter_name, avg(value), count(*) from %s group by %s, counter_name" % (kn, view, kn)).fetchall()
    acc = defaultdict(dict)
    for k, cn, v, n in rows:
        if pat in k:
            acc[k][cn] = (v, n)
    for k, d in acc.items():
        print(k[:110])
        for cn, (v, n) in sorted(d.items()):
            print("    %-28s %16.1f  (n=%d)" % (cn, v, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
