"""Kernel trace of single_window_trace.py -> per optimize(): kernel time, gaps between consecutive kernels, by kernel name."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in rows)
# the timed region: after the longest gap (the sleep)
gap_at, gap, last = 0, 0, iv[0][1]
for i, (s, e, _) in enumerate(iv):
    if s - last > gap: gap, gap_at = s - last, i
    last = max(last, e)
reg = iv[gap_at:]
busy = sum(e - s for s, e, _ in reg)
span = reg[-1][1] - reg[0][0]
gaps = [reg[i + 1][0] - reg[i][1] for i in range(len(reg) - 1)]
small = [g for g in gaps if g < 200000]
print(f"{len(reg)} kernels over {span / 1e6:.2f} ms: running {busy / 1e6:.2f} ms ({busy / span:.2f}), {len(small)} gaps below 0.2 ms: mean {sum(small) / max(len(small), 1) / 1e3:.1f} us, total {sum(small) / 1e6:.2f} ms")
tot, cnt = defaultdict(float), defaultdict(int)
for s, e, n in reg: tot[n] += (e - s) / 1e3; cnt[n] += 1
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]: print(f"  {v / 10:8.1f} us per optimize  {cnt[n] // 10:3d} x {v / cnt[n]:7.1f} us  {n}")
