"""One config-2 window, optimize() ten times: run under `rocprofv3 --kernel-trace` and feed the trace to single_window_gaps.py for the
time the GPU is busy inside one optimize() of the live-SLAM call pattern (kernel time vs the gaps between dependent launches)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from orb_slam3_study_kr_amd import lba, synth
sv = lba.LbaSolver(0)
w = synth.make_config2(100)
sv.upload([w]); sv.optimize()
time.sleep(0.2)
t0 = time.perf_counter()
for _ in range(10): sv.optimize()
print("optimize ms", (time.perf_counter() - t0) / 10 * 1e3)
sv.close()
