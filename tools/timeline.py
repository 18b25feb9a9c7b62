"""print the kernel timeline of one batch in a rocprofv3 kernel trace CSV (from an init_roots_kernel to the next);
usage: timeline.py t_kernel_trace.csv [batches back from the last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("init_roots")]
i0 = idx[-1 - (int(sys.argv[2]) if len(sys.argv) > 2 else 0)]
i1 = idx[idx.index(i0) + 1] if idx.index(i0) + 1 < len(idx) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
tot = {}
for r in rows[i0:i1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
    tot[nm] = tot.get(nm, 0) + d
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  dur {d:8.1f}  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} wgs  {nm}")
print({k: round(v, 1) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}, "wall", (int(rows[i1-1]['End_Timestamp'])-t0)/1e3)
