"""GPU view of profiles/e2e_timeline.py from its rocprofv3 kernel trace: the timed region is what follows the longest gap of the trace;
prints the fraction of the region in which at least one kernel runs, and the kernels' total time by name."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
gap_at, gap, last_end = 0, 0, iv[0][1]
for i, (s, e, _) in enumerate(iv):
    if s - last_end > gap:
        gap, gap_at = s - last_end, i
    last_end = max(last_end, e)
reg = iv[gap_at:]
t0, t1 = reg[0][0], max(e for _, e, _ in reg)
busy, cur_s, cur_e = 0, reg[0][0], reg[0][1]
for s, e, _ in reg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = defaultdict(float)
for s, e, n in reg:
    tot[n.split("(")[0][:60]] += (e - s) / 1e6
print(f"region {1e-6 * (t1 - t0):.1f} ms, some kernel running {1e-6 * busy:.1f} ms = {busy / (t1 - t0):.3f}; sum of kernel durations {sum(tot.values()):.1f} ms")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {v:9.1f} ms  {n}")
