#!/usr/bin/env python3
"""Per-kernel duration table from a rocprofv3 results.db (rocpd SQLite): name, launches, avg/min/max us."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start)/1000., min(end-start)/1000., max(end-start)/1000., "
                  "sum(end-start)/1000. from kernels group by name order by sum(end-start) desc limit %d"
                  % (int(sys.argv[2]) if len(sys.argv) > 2 else 30)).fetchall()
print("%-90s %8s %9s %9s %9s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us"))
for r in rows:
    print("%-90s %8d %9.1f %9.1f %9.1f %10.1f" % (r[0][:90], r[1], r[2], r[3], r[4], r[5]))
