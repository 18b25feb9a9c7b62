"""What runs when, in a rocprofv3 kernel trace of the pipelined bench loop: share of the wall time in which k kernels are in flight, the
share in which the dominant kernel runs, and for every kernel family the time in which it runs ALONE (nothing else in flight) - how much of
the chip's time the small latency-bound kernels really hold.  usage: concurrency.py <t_kernel_trace.csv> [skip_first_fraction [skip_last_fraction]]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
names = {}
for i, r in enumerate(rows):
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    nm = nm.split("<")[0] + ("<" + nm.split("<")[1].split(",")[0] + ">" if "<" in nm else "")
    names[i] = nm
    ev.append((int(r["Start_Timestamp"]), 1, i)); ev.append((int(r["End_Timestamp"]), -1, i))
ev.sort()
t_first, t_last = ev[0][0], ev[-1][0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
t_from = t_first + skip * (t_last - t_first)                     # (warm-up steps and start-up out of the picture)
t_to = t_last - (float(sys.argv[3]) if len(sys.argv) > 3 else 0.2) * (t_last - t_first)      # (... and the drain of the last waves)
live = set()
conc = collections.Counter(); alone = collections.Counter(); present = collections.Counter()
prev = None
for t, d, i in ev:
    if prev is not None and t > prev and prev >= t_from and t <= t_to:
        dt = t - prev
        conc[min(len(live), 8)] += dt
        fam = {names[j] for j in live}
        for f in fam: present[f] += dt
        if len(fam) == 1: alone[next(iter(fam))] += dt
    if d > 0: live.add(i)
    else: live.discard(i)
    prev = t
wall = t_to - t_from
print(f"wall {wall/1e6:.1f} ms; kernels in flight: " + "  ".join(f"{k}{'+' if k == 8 else ''}: {100*v/wall:.1f}%" for k, v in sorted(conc.items())))
for f, v in sorted(present.items(), key=lambda kv: -kv[1]):
    print(f"{f:36s} present {100*v/wall:5.1f}% of the wall   alone {100*alone[f]/wall:5.1f}%")
