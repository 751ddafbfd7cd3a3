"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the plain-text per-kernel summary kept under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof/<host>/<pid>_results.db > profiles/rNN_xxx.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# rocprofv3 --kernel-trace --stats   (durations in microseconds)")
print("%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0]
    print("%-110s %8d %14.1f %12.1f %6.2f%%" % (short[:110], calls, tot, avg, pct))
