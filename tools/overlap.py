"""How much of the small-region kernels' run time lies inside the run time of the one-wavefront expand kernel
(kernel trace of a run with the expand classes on their own streams).  usage: overlap.py <t_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
c1 = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("void expand_kernel<64"))
sm = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "expand_small_kernel" in r["Kernel_Name"])
tot = sum(b - a for a, b in sm); inside = 0; j = 0
for a, b in sm:
    for x, y in c1:
        if y <= a: continue
        if x >= b: break
        inside += max(0, min(b, y) - max(a, x))
print(f"small kernels: {len(sm)} launches, {tot/1e6:.2f} ms, of which {inside/1e6:.2f} ms ({100.0*inside/max(tot,1):.0f} %) while the one-wavefront kernel runs "
      f"({len(c1)} launches, {sum(b-a for a,b in c1)/1e6:.2f} ms)")
