"""Turn a rocprofv3 rocpd .db (kernel trace) into the per-kernel stats table committed under profiles/."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
    f.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for n, calls, tot, avg, pct in rows:
        f.write("%-110s %8d %14.1f %12.1f %8.2f\n" % (n[:110], calls, tot, avg, pct))
print(open(out).read()[:3000])
