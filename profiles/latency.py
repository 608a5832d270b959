"""Single-window latencies of the live-SLAM call pattern: one local-BA window (BASELINE.json configs[1]) and one LocalInertialBA
window (configs[3]) at a time; per-kernel HIP-event times of one optimize()."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from orb_slam3_study_kr_amd import lba, synth
from orb_slam3_study_kr_amd import synth_inertial as si
sv = lba.LbaSolver(0)
w = synth.make_config2(100)
sv.upload([w]); sv.optimize()
t0 = time.perf_counter()
for _ in range(10): sv.optimize()
print("config2 single-window optimize ms", (time.perf_counter()-t0)/10*1e3)
sv.set_profiling(True)
for _ in range(5): sv.optimize()
prof = sv.profile()
print(json.dumps({k: (v[0] // 5, round(v[1]/5, 4)) for k, v in prof.items() if v[0]}))
r = sv.download()[0]
print("iterations", r.iterations, "trials", r.trials)
sv.set_profiling(False)
ws = [si.make_inertial_window(11)]
sv.solve_inertial(ws)
t0 = time.perf_counter()
for _ in range(10): res = sv.solve_inertial(ws)
print("liba single ms", (time.perf_counter()-t0)/10*1e3, "its", res[0].iterations)
