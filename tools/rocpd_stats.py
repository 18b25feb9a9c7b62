"""Kernel statistics (name, calls, total/avg/min/max ns, share) from a rocprofv3 rocpd database (--kernel-trace),
written as the CSV `--stats` would give.  usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3, sys, csv
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = db.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
for r in rows:
    out.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 2)])
