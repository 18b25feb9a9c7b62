"""durations (us) of the kernels whose name contains argv[2] in a rocprofv3 kernel-trace CSV: min / median / max / calls"""
import csv, statistics, sys
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"])
print(f"{sys.argv[2]}: min {d[0]:.1f} med {statistics.median(d):.1f} max {d[-1]:.1f} us over {len(d)} launches" if d else "none")
