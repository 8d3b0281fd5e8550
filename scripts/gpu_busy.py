"""GPU-side view of the last otg_assemble_files job in a rocprofv3 kernel trace: how much of the job's span had a kernel running, and the idle gaps.
usage: python scripts/gpu_busy.py <kernel_trace.csv> <batches per job> [min gap ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2])
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "K_region_prepare" in e[2]]
i0 = marks[-nb]
t0 = ev[i0][0]
busy = 0
cur_s, cur_e = ev[i0][0], ev[i0][1]
gaps = []
last_name = ev[i0][2]
conc = 0      # time with >= 2 kernels running
ends = []
for s, e, n in ev[i0 + 1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if (s - cur_e) / 1e6 >= ming:
            gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e6, last_name, n))
        cur_s, cur_e = s, e
        last_name = n
    elif e > cur_e:
        cur_e = e
        last_name = n
busy += cur_e - cur_s
span = cur_e - t0
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
print("span %.1f ms, a kernel running %.1f ms (%.1f %%), idle %.1f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
print("gaps >= %.1f ms: %d, total %.1f ms" % (ming, len(gaps), sum(g[1] for g in gaps)))
for at, d, a, b in gaps:
    print("  at %8.1f ms  idle %6.2f ms  after %-44s before %s" % (at, d, short(a), short(b)))
# per-kernel totals in the window
tot = {}
for s, e, n in ev[i0:]:
    k = short(n)
    tot[k] = tot.get(k, 0) + (e - s)
print("kernel time summed over both contexts (top 14):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-44s %8.1f ms" % (k, v / 1e6))
