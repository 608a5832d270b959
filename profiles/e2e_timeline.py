"""End-to-end pipeline of bench.py (upload + optimize + download of whole batches on several solver contexts) with a host time stamp
after every phase of every batch: where a batch spends its time when the contexts share the GPU and the host cores.

  python3 profiles/e2e_timeline.py [contexts] [batches per context] [upload threads]

Under `rocprofv3 --kernel-trace` the kernel trace of the same run gives the GPU's view (profiles/gpu_coverage.py): the script sleeps
0.5 s between the warm-up and the timed region so that the region can be found in the trace."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from orb_slam3_study_kr_amd import lba, synth  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n_win = int(os.environ.get("E2E_WINDOWS", "512"))
os.environ["ORBSLAM3_HIP_UPLOAD_THREADS"] = str(threads)
base = [synth.make_config2(100 + k) for k in range(8)]
windows = [base[k % 8] for k in range(n_win)]
solvers = [lba.LbaSolver(0) for _ in range(n_ctx)]
prepared = [sv.prepare(windows) for sv in solvers]


import contextlib
import threading

LOCKS = os.environ.get("E2E_STAGE_LOCKS", "1") != "0"
stage = [threading.Lock() if LOCKS else contextlib.nullcontext() for _ in range(3)]
if LOCKS and int(os.environ.get("E2E_OPT_PERMITS", "1")) > 1:
    stage[1] = threading.Semaphore(int(os.environ["E2E_OPT_PERMITS"]))   # several batches may optimise at once


def drive(k, n):
    sv = solvers[k]
    probs, res, _ = prepared[k]
    rows = []
    for _ in range(n):
        with stage[0]:
            t0 = time.perf_counter()
            sv.upload_prepared(windows, probs)
            t1 = time.perf_counter()
        up = sv.upload_times()
        with stage[1]:
            t1b = time.perf_counter()
            sv.optimize()
            t2 = time.perf_counter()
        with stage[2]:
            t2b = time.perf_counter()
            sv.download_prepared(res)
            t3 = time.perf_counter()
        rows.append((t0, t1, t2 - (t1b - t1), t3 - (t2b - t2) - (t1b - t1), up["pack_ms"], up["copy_ms"]))
    return rows


pool = ThreadPoolExecutor(n_ctx)
list(pool.map(lambda k: drive(k, 1), range(n_ctx)))
# one batch alone
alone = drive(0, 2)[1]
time.sleep(0.5)
t_start = time.perf_counter()
rows = sum(pool.map(lambda k: drive(k, per_ctx), range(n_ctx)), [])
wall = time.perf_counter() - t_start
for sv in solvers:
    sv.close()
r = np.array(rows)
print(f"{n_ctx} contexts x {per_ctx} batches of {n_win} windows, {threads} staging threads per context, stage locks {LOCKS}: {wall / (n_ctx * per_ctx) * 1e3:.1f} ms per batch")
print("phase (ms)            alone   shared(mean)")
print(f"upload: host pass   {alone[4]:7.1f} {r[:, 4].mean():9.1f}")
print(f"upload: H2D+kernels {alone[5]:7.1f} {r[:, 5].mean():9.1f}")
print(f"optimize            {(alone[2] - alone[1]) * 1e3:7.1f} {((r[:, 2] - r[:, 1]) * 1e3).mean():9.1f}")
print(f"download            {(alone[3] - alone[2]) * 1e3:7.1f} {((r[:, 3] - r[:, 2]) * 1e3).mean():9.1f}")
json.dump(dict(wall=wall, batches=n_ctx * per_ctx), open("/tmp/e2e_timeline.json", "w"))
