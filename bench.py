#!/usr/bin/env python3
"""bench.py -- local-BA windows/sec (+ ORB matches/sec) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 via torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM: osh_lba_optimize() on `--windows` independent config-2 windows (BASELINE.json configs[1]:
50 optimisable + 10 fixed keyframes, ~10k landmarks, ~75k stereo edges, <=10 LM iterations with
the reference's early-stop rules).  Multi-GPU: weak scaling, window w -> rank w mod G, no
data-path collective; barrier + max-over-ranks timing via torch.distributed (RCCL).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from orb_slam3_study_kr_amd import dist as osh_dist  # noqa: E402
from orb_slam3_study_kr_amd import launch as osh_launch  # noqa: E402
from orb_slam3_study_kr_amd import synth  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, AMD product figure; v_mfma_f64_16x16x4 measured at 64 cycles
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _make_cfg2(seed):
    return synth.make_config2(seed)


def generate_windows(seeds, workers=8):
    if len(seeds) <= 2 or workers <= 1:
        return [_make_cfg2(s) for s in seeds]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(workers, len(seeds))) as pool:
        return pool.map(_make_cfg2, seeds)


# Steps of the reference loop -> the device kernels that implement them (names of capi.KERNEL_NAMES).
# Linearisation and Schur complement are ONE step here: k_schur_fused forms the pose side of buildSystem and the Schur
# products in the same pass (the 6x3 blocks Hpl exist only in registers), so their times cannot be told apart.
STEP_KERNELS = {
    "lin_schur": ["linearize", "lin_aux", "lin_pose", "pose_hess", "schur", "schur_cross", "schur_reduce"],
    "solve": ["solve"], "backsub_residual": ["backsub", "residual"],   # k_backsub ends with the trial residual of its chunk (round 3): one pass
}


def kernel_algorithmic_bytes(windows, results, plan=None):
    """Bytes each step of the loop has to move over one optimize().

    "survey": SURVEY.md 8(d)'s per-window byte formulas x the number of times the reference loop runs the step (linearise:
    once per iteration; Schur / solve / back-substitution / trial residual: once per LM trial).  Those formulas charge the
    144-byte Hpl block of every optimisable edge to linearise (write), Schur (read) and back-substitution (read).
    "impl": what THIS implementation's kernels need at least: Hpl is never stored, each of those passes reads the 32-byte
    observation record + 8 bytes of indices per edge instead and the landmark factor (72 B) per landmark; the Schur products
    travel from k_schur_fused to k_schur_reduce as one 288-byte contribution per (item, pose pair) -- written once and read once per
    trial -- and every item reads its 32-byte landmark records (`plan`: the totals of the uploaded plan)."""
    sv = dict(lin_schur=0, solve=0, backsub_residual=0)
    im = dict(lin_schur=0, solve=0, backsub_residual=0)
    for w, r in zip(windows, results):
        b = w.algorithmic_bytes()
        P, F, L, E, Ef = w.n_free, w.n_fixed, w.n_points, w.n_edges, w.n_free_edges
        it, tr = int(r.iterations), int(r.trials)
        solve = (6 * P) * (6 * P + 1) * 8 + 2 * 6 * P * 8 + 2 * P * 56
        sv["lin_schur"] += it * b["lin"] + tr * b["schur"]
        sv["solve"] += tr * solve
        sv["backsub_residual"] += tr * (b["back"] + 2 * L * 24 + b["resid"])
        poses = (P + F) * 96
        im["lin_schur"] += it * (E * 40 + L * 24 + poses + L * 144 + P * 216) + tr * (Ef * 32 + L * 96 + (6 * P) * (6 * P + 1) * 8)
        im["solve"] += tr * solve
        im["backsub_residual"] += tr * (E * 40 + L * (72 + 24 + 24 + 24) + 6 * P * 8 + 2 * poses)   # every edge record once, both pose sets
    if plan is not None and len(windows):
        per_window = (plan["contributions"] * 288 * 2 + plan.get("records", 0) * 32) / len(windows)
        im["lin_schur"] += per_window * float(sum(int(r.trials) for r in results))
    return sv, im


def make_lba_inputs(args, rank, world):
    """Pure-numpy input generation; runs BEFORE anything touches the GPU (it forks worker processes)."""
    my = osh_dist.shard_indices(args.windows * world, rank, world)   # window w -> rank w mod G
    seeds = [100 + w for w in my]
    cache = Path(args.cache_inputs + (f".rank{rank}" if world > 1 else "")) if args.cache_inputs else None   # one file per rank
    if cache is not None and cache.exists():
        import pickle
        with open(cache, "rb") as f:          # written by this script (own file), see --prepare-only
            got = pickle.load(f)
        if got["seeds"] == seeds:
            return got["windows"]
    workers = args.workers if args.workers > 0 else max(1, min(16, usable_cores() // max(1, world)))   # (the cgroup's share, not the machine's cores)
    windows = generate_windows(seeds, workers=workers)
    if cache is not None:
        import pickle
        with open(cache, "wb") as f:
            pickle.dump({"seeds": seeds, "windows": windows}, f)
    return windows


def measured_traffic(step, windows_per_gpu):
    """HBM bytes per round of the kernels of `step` from the committed rocprofv3 PMC summary (profiles/traffic.json:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate --pmc passes), or None."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text())
    if t.get("windows_per_gpu") != windows_per_gpu:
        return None
    per = t.get("bytes_per_launch", {})
    # k_lin_lm's factor-only launch and k_schur_fused's mode-0 pass share their kernel names with the main launches
    names = [k for k in STEP_KERNELS[step] if k not in ("lin_aux", "lin_pose")]
    if any(k not in per for k in names):
        return None
    return float(sum(per[k] for k in names))


def measured_other_traffic(kernel, launch_key, launch_value):
    """HBM bytes per launch of a kernel outside the local-BA loop (profiles/traffic.json, "other_bytes_per_launch": uniform launches under
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/other_kernels.py --uniform), or None when the bench launch is another size."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text())
    if launch_key != launch_value:
        return None
    for k, v in t.get("other_bytes_per_launch", {}).items():
        if k.startswith(kernel):
            return float(v)
    return None


def run_lba(args, info, windows):
    """Timed region: every window resident in HBM; `--streams` solver contexts (own HIP stream each, driven by a host
    thread each) run osh_lba_optimize concurrently so the latency-bound kernels of one half overlap the other half."""
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from orb_slam3_study_kr_amd import lba
    ns = max(1, min(args.streams, len(windows)))
    parts = [windows[k::ns] for k in range(ns)]
    solvers = [lba.LbaSolver(info.local_rank) for _ in range(ns)]
    t0 = time.perf_counter()
    for sv, part in zip(solvers, parts):
        sv.upload(part)
    upload_s = time.perf_counter() - t0
    pool = ThreadPoolExecutor(ns)

    def step():
        list(pool.map(lambda sv: sv.optimize(), solvers))   # each call returns after its own stream is idle

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    elapsed_local = time.perf_counter() - t0
    elapsed = osh_dist.all_reduce_max(info, elapsed_local)
    results_parts = [sv.download() for sv in solvers]
    results = [None] * len(windows)
    for k, rp in enumerate(results_parts):
        results[k::ns] = rp
    for sv in solvers:
        sv.close()
    pool.shutdown()
    # per-kernel HIP-event timing: one untimed, un-overlapped step of ALL windows on a single stream
    solver = lba.LbaSolver(info.local_rank)
    solver.upload(windows)
    solver.optimize()
    solver.set_profiling(True)
    solver.optimize()
    prof = solver.profile()
    plan = solver.plan_stats()
    solver.set_profiling(False)
    solver.close()
    alg, alg_impl = kernel_algorithmic_bytes(windows, results, plan)
    return dict(windows=windows, results=results, elapsed=elapsed, elapsed_local=elapsed_local, upload_s=upload_s, prof=prof, alg=alg, alg_impl=alg_impl, plan=plan)


def run_batch_sweep(args, info, windows):
    """windows/s against the batch size B in {1, 8, 64, 256}: the HBM-resident optimize() and the whole
    osh_lba_solve (upload + optimize + download) of one context, one batch at a time; B = 1 is the live-SLAM call pattern
    (LocalMapping.cc:154-160), its times are the single-window latencies."""
    from orb_slam3_study_kr_amd import lba
    out = {}
    with lba.LbaSolver(info.local_rank) as sv:
        for B in (1, 8, 64, 256):
            ws = windows[:B]
            if len(ws) < B:
                break
            probs, res, outs = sv.prepare(ws)     # `outs` owns the result arrays `res` points into: keep it alive
            sv.upload_prepared(ws, probs)
            sv.optimize()
            reps = 5 if B <= 8 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                sv.optimize()
            opt_ms = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            for _ in range(reps):
                sv.upload_prepared(ws, probs)
                sv.optimize()
                sv.download_prepared(res)
            call_ms = (time.perf_counter() - t0) / reps * 1e3
            out[str(B)] = dict(optimize_ms=opt_ms, windows_per_s_resident=B / (opt_ms * 1e-3), solve_call_ms=call_ms,
                               windows_per_s_end_to_end=B / (call_ms * 1e-3), packed_on="device" if sv.pack_profile()["on_device"] else "host")
            del outs
    return out


def run_orb_sweep(args, info):
    """frame pairs/s of osh_orb_match_local_points against the batch size B in {1, 64, 1024}."""
    from orb_slam3_study_kr_amd import orb
    base = synth.make_orb_pair(7, 2000, 2000)
    rng = np.random.Generator(np.random.PCG64(7100))
    out = {}
    for B in (1, 64, 1024):
        pairs = []
        for k in range(min(B, 64)):      # 64 distinct query sets, cycled
            q = base.query_desc ^ np.packbits(rng.uniform(0, 1, (2000, 256)) < 0.01, axis=1)
            pairs.append(synth.OrbPair(np.ascontiguousarray(q), base.train_desc, base.train_level))
        pairs = [pairs[k % len(pairs)] for k in range(B)]
        m = orb.OrbMatcher(info.local_rank)
        m.upload(pairs)
        m.match_local_points()
        reps = 10 if B <= 64 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            n_matches, _, _, _ = m.match_local_points()
        dt = (time.perf_counter() - t0) / reps
        m.close()
        out[str(B)] = dict(ms_per_call=dt * 1e3, frame_pairs_per_s=B / dt, matches_per_s=float(n_matches.sum()) / dt, pair_evals_per_s=B * 4.0e6 / dt)
    return out


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a 16-core share
    of a 256-thread host to each GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_all_cores(windows, budget_s=10.0):
    """SURVEY.md 8(d) CPU baseline (ii): every host core solves its own window (the oracle is one thread per solve; ctypes releases
    the GIL), for a bounded time."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    try:
        ob.load(native=True)
        native = True
    except Exception:
        native = False
    n_cores = usable_cores()
    deadline = time.perf_counter() + budget_s

    def work(k):
        n = 0
        while time.perf_counter() < deadline:
            ob.lba_solve(windows[(k + n * n_cores) % len(windows)], native=native)
            n += 1
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_cores) as pool:
        done = sum(pool.map(work, range(n_cores)))
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="windows/s", cores=n_cores, kind="port", nproc=os.cpu_count(), usable_cores=n_cores, cpu_model=cpu_model(),
                sample=f"{done} solves of the same config-2 windows, one window per core on {n_cores} threads ({dt:.1f} s)",
                march="native" if native else "x86-64-v3")


def run_end_to_end(args, info, windows):
    """Upload + optimize + download of whole batches, the call pattern of a caller that hands over HOST buffers: several solver
    contexts (own stream, own pinned staging), each driven by its own host thread, work on different batches, so the host packing /
    H2D copy of one batch overlaps the optimisation of another on the device.  Every batch is packed from scratch (sort,
    Schur plan, landmark renumbering) -- the work SparseOptimizer::initializeOptimization + BlockSolver::buildStructure do
    inside the call this replaces.  Only C-ABI calls are inside the timed region."""
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from orb_slam3_study_kr_amd import lba
    n_ctx = max(1, args.e2e_contexts)
    solvers = [lba.LbaSolver(info.local_rank) for _ in range(n_ctx)]
    prepared = [sv.prepare(windows) for sv in solvers]
    per_ctx = max(2, args.e2e_batches)
    n_batches = n_ctx * per_ctx

    import threading
    stage = [threading.Lock() for _ in range(3)]   # upload | optimize | download: one batch in each stage at a time

    def drive(k, n):
        """One host thread owns one context and runs its batches one after the other (a context is not thread safe).  The three
        stages of a batch use different resources (host cores + PCIe down, the GPU, PCIe up + host copies); a lock per stage keeps
        ONE batch in each stage, so the contexts form a pipeline instead of all uploading, then all optimising, at the same time
        (measured without the locks: the GPU ran no kernel for a third of the time, profiles/e2e_timeline.py)."""
        sv = solvers[k]
        probs, res, _ = prepared[k]
        ups = []
        for _ in range(n):
            with stage[0]:
                sv.upload_prepared(windows, probs)
            ups.append(sv.upload_times())
            with stage[1]:
                sv.optimize()
            with stage[2]:
                sv.download_prepared(res)
        return ups

    # staging threads of an upload: one batch is staged at a time, so it may take the host cores of this rank (all ranks of a node
    # share them: divide by the world size)
    user_threads = os.environ.get("ORBSLAM3_HIP_UPLOAD_THREADS")
    e2e_threads = int(user_threads) if user_threads else max(2, min(16, usable_cores()) // max(1, info.world))
    os.environ["ORBSLAM3_HIP_UPLOAD_THREADS"] = str(e2e_threads)
    pool = ThreadPoolExecutor(n_ctx)
    list(pool.map(lambda k: drive(k, 1), range(n_ctx)))            # warm-up: staging buffers, device buffers
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    t0 = time.perf_counter()
    ups = sum(pool.map(lambda k: drive(k, per_ctx), range(n_ctx)), [])
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    elapsed = osh_dist.all_reduce_max(info, time.perf_counter() - t0)
    # one batch alone, phases timed one after the other (no overlap, all packing threads): where an end-to-end batch spends its time
    if user_threads is None:
        del os.environ["ORBSLAM3_HIP_UPLOAD_THREADS"]
    sv = solvers[0]
    probs, res, outs = prepared[0]
    t = [time.perf_counter()]
    sv.upload_prepared(windows, probs); t.append(time.perf_counter())
    sv.optimize(); t.append(time.perf_counter())
    sv.download_prepared(res); t.append(time.perf_counter())
    up = sv.upload_times()
    for s_ in solvers:
        s_.close()
    pool.shutdown()
    return dict(elapsed=elapsed, n_batches=n_batches, n_ctx=n_ctx, upload_threads=e2e_threads, serial_ms=dict(upload=(t[1] - t[0]) * 1e3, optimize=(t[2] - t[1]) * 1e3,
                                                                       download=(t[3] - t[2]) * 1e3, upload_pack=up["pack_ms"], upload_copy=up["copy_ms"]),
                pack_ms_mean=float(np.mean([u["pack_ms"] for u in ups])), copy_ms_mean=float(np.mean([u["copy_ms"] for u in ups])))


def run_orb(args, info):
    from orb_slam3_study_kr_amd import orb
    n_pairs = args.orb_pairs
    my = osh_dist.shard_indices(n_pairs * info.world, info.rank, info.world)
    base = synth.make_orb_pair(7, 2000, 2000)
    rng = np.random.Generator(np.random.PCG64(7000 + info.rank))
    pairs = []
    for _ in my:  # distinct query sets per pair (cheap variation of the config-3 pair)
        q = base.query_desc ^ np.packbits(rng.uniform(0, 1, (2000, 256)) < 0.01, axis=1)
        pairs.append(synth.OrbPair(np.ascontiguousarray(q), base.train_desc, base.train_level))
    m = orb.OrbMatcher(info.local_rank)
    m.upload(pairs)
    import torch
    for _ in range(max(1, args.warmup)):
        m.match_local_points()
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    steps = max(args.steps, 5)
    t0 = time.perf_counter()
    for _ in range(steps):
        n_matches, _, _, rounds = m.match_local_points()    # search + sequential slot occupancy, all on the device; D2H of the assignment included
    torch.cuda.synchronize()
    osh_dist.barrier(info)
    elapsed = osh_dist.all_reduce_max(info, time.perf_counter() - t0)
    accepted = int(n_matches.sum())
    m.match()
    res = m.download()
    m.set_profiling(True)
    for _ in range(3):
        m.match_local_points()
    launches, ms = m.profile()
    rl, rms = m.resolve_profile()
    m.close()
    tot = osh_dist.all_reduce_sum(info, [float(accepted), float(len(pairs))])
    per_step_s = elapsed / steps
    sbp = search_by_projection_end_to_end() if info.rank == 0 else None
    fuse = fuse_end_to_end() if info.rank == 0 else None
    return dict(matches_per_s=tot[0] / per_step_s, pair_evals_per_s=tot[1] * 4.0e6 / per_step_s,
                frame_pairs_per_s=tot[1] / per_step_s, accepted_per_pair=tot[0] / max(tot[1], 1),
                kernel_ms=ms / max(launches, 1), resolve_ms=rms / max(rl, 1), rounds=rounds, pairs_per_gpu=len(pairs), pair0=pairs[0], res=res, sbp=sbp, fuse=fuse)


def search_by_projection_end_to_end(n_kp=2000, n_mp=2000, reps=20):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) as Tracking::SearchLocalPoints calls it, through the
    drop-in ORBmatcher.cc: gather of the projected points, H2D, the windowed device search, D2H and the host replay of the
    sequential slot occupancy -- the whole call on the wall clock, one frame at a time (the reference's call pattern)."""
    from orb_slam3_study_kr_amd import host
    rng = np.random.Generator(np.random.PCG64(5))
    xy = np.stack([rng.uniform(5, synth.IMG_W - 5, n_kp), rng.uniform(5, synth.IMG_H - 5, n_kp)], axis=1).astype(np.float32)
    octave = rng.integers(0, synth.N_LEVELS, n_kp).astype(np.int32)
    desc = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
    src = rng.permutation(n_kp)[:n_mp]
    mp_desc = desc[src] ^ np.packbits(rng.uniform(0, 1, (n_mp, 256)) < 0.06, axis=1)
    proj = (xy[src] + rng.normal(0, 2.0, (n_mp, 2))).astype(np.float32)
    level = np.clip(octave[src] + rng.integers(0, 2, n_mp), 0, synth.N_LEVELS - 1).astype(np.int32)
    viewcos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    f = host.HostFrame(xy, octave, desc)
    try:
        n, _ = f.search_local_points(mp_desc, proj, level, viewcos, nnratio=0.8, th=3.0)
        ms = 0.0
        for _ in range(reps):
            f.search_local_points(mp_desc, proj, level, viewcos, nnratio=0.8, th=3.0)
            ms += f.lib.osh_host_last_call_ms()    # the ORBmatcher::SearchByProjection call alone, timed inside the C++ wrapper
        ms /= reps
    finally:
        f.close()
    return dict(call="ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th=3, ...) through the drop-in ORBmatcher.cc",
                keypoints=n_kp, map_points=n_mp, matches=int(n), ms_per_call=ms, matches_per_s=n / (ms * 1e-3),
                includes="the C++ call alone (steady_clock around it inside the harness wrapper: no Python, no construction of the map points): "
                         "host gather + H2D + windowed device search + occupancy rounds + D2H")


def fuse_end_to_end(n_kp=2000, n_mp=2000, reps=10):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) as LocalMapping::SearchInNeighbors calls it, through the drop-in ORBmatcher.cc:
    projection gates and candidate lists on the host, one batched device search, the ordered replace / add replay -- wall clock."""
    from orb_slam3_study_kr_amd import host
    rng = np.random.Generator(np.random.PCG64(6))
    f32 = np.float32
    xy = np.stack([rng.uniform(5, synth.IMG_W - 5, n_kp), rng.uniform(5, synth.IMG_H - 5, n_kp)], axis=1).astype(f32)
    octave = rng.integers(0, synth.N_LEVELS, n_kp).astype(np.int32)
    desc = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
    depth_kp = rng.uniform(4, 10, n_kp)
    uright = np.where(rng.uniform(0, 1, n_kp) < 0.6, xy[:, 0] - float(synth.BF) / depth_kp, -1.0).astype(f32)
    src = rng.integers(0, n_kp, n_mp)
    noisy = xy[src] + rng.normal(0, 0.8, (n_mp, 2))
    depth = depth_kp[src]
    pos = np.stack([(noisy[:, 0] - float(synth.CX)) / float(synth.FX) * depth, (noisy[:, 1] - float(synth.CY)) / float(synth.FY) * depth, depth], axis=1).astype(f32)
    mp_desc = desc[src] ^ np.packbits(rng.uniform(0, 1, (n_mp, 256)) < 0.05, axis=1)
    maxd = (depth * synth.SCALE_FACTORS[octave[src]].astype(np.float64)).astype(f32)
    mind = (maxd / f32(synth.SCALE_FACTORS[-1]) * f32(0.5)).astype(f32)
    normal = (pos / np.linalg.norm(pos, axis=1, keepdims=True)).astype(f32)
    nobs = rng.integers(1, 8, n_mp).astype(np.int32)
    n_res = n_kp // 2
    slot_res = -np.ones(n_kp, dtype=np.int32)
    slot_res[rng.permutation(n_kp)[:n_res]] = np.arange(n_res)
    res_nobs = rng.integers(0, 8, n_res).astype(np.int32)
    f = host.HostFrame(xy, octave, desc, uright=uright)
    try:
        args = (pos, mp_desc, np.stack([mind, maxd], axis=1), normal, nobs, slot_res, res_nobs)
        n, _ = f.fuse(*args)
        ms = 0.0
        for _ in range(reps):
            f.fuse(*args)
            ms += f.lib.osh_host_last_call_ms()    # the ORBmatcher::Fuse call alone, timed inside the C++ wrapper
        ms /= reps
    finally:
        f.close()
    return dict(call="ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th=3) through the drop-in ORBmatcher.cc", keypoints=n_kp, map_points=n_mp,
                fused=int(n), ms_per_call=ms,
                includes="the C++ call alone (steady_clock around it inside the harness wrapper): projection gates + candidate lists + H2D + device "
                         "search + D2H + ordered replace / add replay on the map")


def make_inertial_inputs(args):
    from orb_slam3_study_kr_amd import synth_inertial as si
    # BASELINE.json configs[3] / SURVEY.md 8(d) config 4: 10 temporal + 1 + 20 fixed keyframes, ~2 000 landmarks (3 600 candidates, the visible ones stay)
    base = [si.make_inertial_window(11 + k, n_points=3600) for k in range(8)]
    return [base[k % len(base)] for k in range(args.inertial_windows)]


def run_inertial(args, info, windows):
    """BASELINE.json configs[3]: LocalInertialBA windows through osh_liba_solve (one launch; a group of thread blocks per window: 32
    for a single window, fewer as the batch grows).  The call includes the H2D upload and the D2H download (the reference calls it
    one window at a time)."""
    from orb_slam3_study_kr_amd import lba
    solver = lba.LbaSolver(info.local_rank)
    solver.solve_inertial(windows[:1])
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        solver.solve_inertial(windows[:1])
    single_ms = (time.perf_counter() - t0) / reps * 1e3
    solver.solve_inertial(windows)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = solver.solve_inertial(windows)
    batch_s = (time.perf_counter() - t0) / reps
    # the same kernel on the problems of Optimizer::FullInertialBA (every keyframe of a small map optimisable, lambda 1e-5, optimize(7) as the
    # loop closer calls it, src/LoopClosing.cc:2291) and MergeInertialBA (12 temporal + 31 pose-only keyframes, lambda 1e3, optimize(8))
    import dataclasses
    from orb_slam3_study_kr_amd import synth_inertial as si
    map_ba = {}
    for name, n_opt, lam, its in (("full_inertial_ba_100_keyframes", 100, 1e-5, 7), ("full_inertial_ba_400_keyframes", 400, 1e-5, 7),
                                  ("merge_inertial_ba_43_keyframes", 43, 1e3, 8)):
        w = si.make_inertial_window(900 + n_opt, n_opt=n_opt, n_fixed=0, n_points=40 * n_opt, large=True)
        w = dataclasses.replace(w, lambda_init=lam, max_iterations=its, link_robust=np.ones_like(w.link_robust))
        solver.solve_inertial([w])
        t0 = time.perf_counter()
        r = solver.solve_inertial([w])[0]
        map_ba[name] = dict(ms=(time.perf_counter() - t0) * 1e3, keyframes=n_opt, landmarks=w.n_points, edges=w.n_edges, lm_iterations=int(r.iterations))
    solver.close()
    out = dict(metric=f"LocalInertialBA windows/sec (10 temporal KF + 21 fixed, {int(np.mean([w.n_points for w in windows]))} landmarks, "
                      f"{int(np.mean([w.n_edges for w in windows]))} stereo edges, IMU preintegration edges)",
               map_sized=map_ba,
               windows_per_s=len(windows) / batch_s, windows_per_batch=len(windows), single_window_latency_ms=single_ms,
               lm_iterations_mean=float(np.mean([r.iterations for r in res])), includes="H2D upload + D2H download", dtype="f64 (+f32 preintegration getters)")
    # SURVEY.md 8(d) byte model applied to the visual part of the inertial window (d = 3, P = the temporal keyframes): per iteration and
    # trial (lin + resid) + (schur + back + update + resid).  A window is a few MB and its optimisation a chain of barrier-separated
    # phases on at most one XCD, half of the time in a 150x150 LDL^T on one CU: it is latency bound and the
    # fraction of the HBM peak is tiny by construction; it is reported as a yardstick.
    alg = 0.0
    for w, r in zip(windows, res):
        E, L, P, F = w.n_edges, w.n_points, w.n_opt, w.n_fixed + w.n_fixed_imu
        Ef = int((w.edge_pose < w.n_opt).sum())
        resid = E * (8 * 3 + 16) + L * 24 + (P + F) * 56
        lin = resid + Ef * 144 + L * 72 + P * 216
        schur = Ef * 144 + L * 72 + (6 * P) * (6 * P + 1) * 8
        back = Ef * 144 + L * 96 + 6 * P * 8
        upd = 2 * (P * 56 + L * 24)
        alg += r.iterations * (lin + resid) + r.trials * (schur + back + upd + resid)
    out["roofline"] = dict(bound="hbm", kernel="k_liba<24>", achieved=alg / batch_s / 1e9, peak=8000.0, unit="GB/s", frac=alg / batch_s / 8e12,
                           traffic=measured_other_traffic("k_liba", len(windows), 128),
                           note="whole call incl. upload / download; a group of thread blocks per window, latency bound (DESIGN.md 4b)")
    if info.rank == 0 and info.world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 4.0:
            ob.liba_solve(windows[n % len(windows)])
            n += 1
        out["cpu_baseline"] = dict(value=n / (time.perf_counter() - t0), unit="windows/s", cores=1, kind="port",
                                   sample=f"{n} solves of the same windows by oracle/liba_oracle.c, one thread")
    return out


def cpu_baseline(windows, budget_s=12.0):
    """The oracle (kind "port": CPU restatement of the g2o path, oracle/lba_oracle.c) timed single
    threaded on this host, rebuilt here with -O3 -march=native like the reference's own flags."""
    from oracle import binding as ob
    try:
        ob.load(native=True)
        native = True
    except Exception:
        native = False
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:        # bounded sample: ~12 s of single-thread CPU work
        ob.lba_solve(windows[n % len(windows)], native=native)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="windows/s", cores=1, kind="port",
                sample=f"{n} solves over the same config-2 windows (cycled), one at a time on one thread ({dt:.1f} s)",
                march="native" if native else "x86-64-v3")


def stub_main(args):
    """The rank plumbing of main() with the solver stubbed out (tests/test_bench_launch_cpu.py, gloo on CPU): same
    sharding, barriers, max-over-ranks timing and rank-0 JSON line; a step is a 10 ms sleep."""
    info = osh_dist.init_from_env(backend="gloo")
    osh_launch.check_world(args.gpus, info.world)
    my = osh_dist.shard_indices(args.windows * info.world, info.rank, info.world)
    for _ in range(args.warmup):
        time.sleep(0.01)
    osh_dist.barrier(info)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01)
    osh_dist.barrier(info)
    elapsed_local = time.perf_counter() - t0
    elapsed = osh_dist.all_reduce_max(info, elapsed_local)
    done = osh_dist.all_reduce_sum(info, [float(len(my))])[0]
    ranks_seen = int(round(osh_dist.all_reduce_sum(info, [1.0])[0]))
    per_rank = osh_dist.all_gather_floats(info, len(my) / (elapsed_local / args.steps))
    if info.rank == 0:
        ms = elapsed / args.steps * 1e3
        print(json.dumps({"metric": "stub", "stub": True, "value": done / (ms * 1e-3), "unit": "windows/s", "n_gpus": info.world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "scaling": "weak",
                          "ranks_seen": ranks_seen, "windows_per_s_by_rank": per_rank,
                          "config": {"windows_per_gpu": args.windows, "global_windows": int(done)}}))
    osh_dist.finalize(info)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=512, help="config-2 windows resident per GPU (one step solves them all)")
    ap.add_argument("--orb-pairs", type=int, default=64, help="2000x2000 frame pairs per GPU per ORB step")
    ap.add_argument("--streams", type=int, default=2, help="solver contexts (HIP streams) sharing the GPU in the timed region")
    ap.add_argument("--workers", type=int, default=0, help="input-generation processes (0 = auto; use 1 under rocprofv3: "
                    "the profiler's signal handler hangs on the pool's worker teardown)")
    ap.add_argument("--cache-inputs", default="", help="pickle file to store / reuse the generated windows")
    ap.add_argument("--prepare-only", action="store_true", help="generate (and cache) the inputs, then exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-orb", action="store_true")
    ap.add_argument("--no-sweeps", action="store_true", help="skip the batch-size sub-tables (config 2: B = 1, 8, 64, 256; config 3: B = 1, 64, 1024)")
    ap.add_argument("--inertial-windows", type=int, default=128, help="config-4 windows per osh_liba_solve call (0 = skip)")
    ap.add_argument("--e2e-contexts", type=int, default=4, help="solver contexts (each with its own host thread, stream and pinned staging) of the end-to-end run")
    ap.add_argument("--e2e-batches", type=int, default=6, help="batches per solver context in the end-to-end (upload + optimize + download) run; 0 = skip")
    ap.add_argument("--stub-solver", action="store_true", help="CPU rehearsal of the rank plumbing (gloo): no GPU work, the "
                    "timed step is a fixed sleep; the JSON line is marked \"stub\" and is not a measurement")
    args = ap.parse_args()

    if osh_launch.needs_spawn(args.gpus):
        # `python bench.py --gpus N` outside a rendezvous: start the N ranks as a CHILD process (never exec; this parent
        # has not touched the GPU) and hand its exit code back
        raise SystemExit(osh_launch.spawn_ranks(str(Path(__file__).resolve()), args.gpus, sys.argv[1:]))
    env_rank, env_world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    osh_launch.check_world(args.gpus, env_world)
    if args.stub_solver:
        return stub_main(args)
    windows = make_lba_inputs(args, env_rank, env_world)
    if args.prepare_only:
        return
    inertial_windows = make_inertial_inputs(args) if args.inertial_windows > 0 else None
    info = osh_dist.init_from_env()
    osh_launch.check_world(args.gpus, info.world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(info.local_rank)

    lba_out = run_lba(args, info, windows)
    e2e_out = run_end_to_end(args, info, windows) if args.e2e_batches > 0 else None
    sweep_out = run_batch_sweep(args, info, windows) if (info.rank == 0 and not args.no_sweeps) else None
    orb_sweep_out = run_orb_sweep(args, info) if (info.rank == 0 and not args.no_sweeps and not args.no_orb) else None
    orb_out = None if args.no_orb else run_orb(args, info)
    inertial_out = run_inertial(args, info, inertial_windows) if inertial_windows else None

    n_gpus = info.world
    ms_per_step = lba_out["elapsed"] / args.steps * 1e3
    value = args.windows * n_gpus / (ms_per_step * 1e-3)
    # what the process group saw: an all-reduce of 1 over RCCL (= the number of ranks that took part) and every rank's own rate
    ranks_seen = int(round(osh_dist.all_reduce_sum(info, [1.0])[0]))
    per_rank = osh_dist.all_gather_floats(info, args.windows / (lba_out["elapsed_local"] / args.steps))

    # roofline of the dominant step of the loop (largest total HIP-event time in one optimize()); a step is one or
    # more kernels (STEP_KERNELS), its launch time the sum of theirs, its algorithmic bytes SURVEY.md 8(d)'s figure
    from orb_slam3_study_kr_amd import lba
    prof, alg, alg_impl = lba_out["prof"], lba_out["alg"], lba_out["alg_impl"]
    step_ms = {st: sum(prof[k][1] for k in ks) for st, ks in STEP_KERNELS.items()}
    dom = max(step_ms, key=step_ms.get)
    launches = max(prof[k][0] for k in STEP_KERNELS[dom])      # LM rounds (k_lin_lm's factor-only launch and the mode-0 pass run once)
    total_ms = step_ms[dom]
    avg_ms = total_ms / max(launches, 1)
    achieved = (alg[dom] / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    achieved_impl = (alg_impl[dom] / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = dict(bound="hbm", kernel=dom + " = " + " + ".join(lba.kernel_symbol(k) for k in STEP_KERNELS[dom]),
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(dom, args.windows),
                    frac_measured=(measured_traffic(dom, args.windows) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (measured_traffic(dom, args.windows) and avg_ms > 0) else None,
                    avg_launch_ms=avg_ms, launches=launches, algorithmic_bytes_per_launch=alg[dom] / max(launches, 1),
                    bytes_this_implementation_needs_per_launch=alg_impl[dom] / max(launches, 1), frac_of_needed_bytes=achieved_impl / HBM_PEAK_GBS,
                    avg_ms_by_kernel={lba.kernel_symbol(k): prof[k][1] / max(prof[k][0], 1) for k in STEP_KERNELS[dom]},
                    note="achieved = SURVEY.md 8(d) bytes of linearise + Schur (they charge the 144-byte Hpl block of every edge, "
                         "which this implementation never stores) / sum of the step's mean kernel times; the step is FP64-issue "
                         "bound (fp64 below), not HBM bound")
    kernels = {k: dict(launches=prof[k][0], total_ms=round(prof[k][1], 4)) for k in prof}
    # FP64 view of the dominant step: the Schur products run on the matrix cores (v_mfma_f64_16x16x4_f64 = 2048 flop; every launch
    # of k_schur_fused issues the plan's MFMA count for each still-active window), the edge descriptions on the vector unit.
    # On gfx950 FP64 MFMA and FP64 VALU share one pipe (profiles/ubench/f64_overlap.hip: their times ADD), so the bound is
    # total FP64 issue; peak = AMD's 78.6 TFLOP/s FP64 figure for MI355X (vector = matrix rate).
    res_all = lba_out["results"]
    mfma_total = lba_out["plan"]["mfma_per_pass"] / max(len(res_all), 1) * float(sum(int(r.trials) for r in res_all))
    mfma_ms = prof["schur"][1] + prof["schur_cross"][1]
    mfma = dict(kernel="k_schur_fused<true> + k_schur_fused<false>", issued_TFLOPs=(mfma_total * 2048 / (mfma_ms * 1e-3) / 1e12) if mfma_ms > 0 else 0.0,
                peak_TFLOPs=FP64_MFMA_PEAK_TFLOPS, useful_fraction_of_issued=lba_out["plan"]["useful_blocks"] * 108 * 2 / max(lba_out["plan"]["mfma_per_pass"] * 2048, 1))
    mfma["frac"] = mfma["issued_TFLOPs"] / FP64_MFMA_PEAK_TFLOPS
    # SURVEY.md 8(d) algorithmic flops: ~530 per edge and linearisation, Schur 87 MFLOP per trial at config 2 = sum over landmarks of
    # k(k+1)/2 * 216 + k * 108 + 60
    flops = 0.0
    for w, r in zip(lba_out["windows"], res_all):
        k = np.bincount(w.edge_point[w.edge_pose < w.n_free], minlength=w.n_points).astype(np.float64)
        flops += int(r.iterations) * 530.0 * w.n_edges + int(r.trials) * float((k * (k + 1) / 2 * 216 + k * 108 + 60).sum())
    roofline["fp64"] = dict(algorithmic_TFLOPs=flops / (step_ms[dom] * 1e-3) / 1e12 if step_ms[dom] > 0 else 0.0, peak_TFLOPs=FP64_MFMA_PEAK_TFLOPS,
                            frac=(flops / (step_ms[dom] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if step_ms[dom] > 0 else 0.0, mfma=mfma)
    steps = {}
    for st in STEP_KERNELS:
        ms = step_ms[st]
        steps[st] = dict(total_ms=round(ms, 4), survey_GBps=(alg[st] / (ms * 1e-3) / 1e9) if ms > 0 else None,
                         needed_GBps=(alg_impl[st] / (ms * 1e-3) / 1e9) if ms > 0 else None)
        assert ms <= 0 or alg_impl[st] / (ms * 1e-3) / 1e9 <= HBM_PEAK_GBS, f"step {st}: more bytes than the HBM peak allows -- wrong byte model"
    whole_bytes = sum(alg.values())
    res = lba_out["results"]
    out = {
        "metric": "local-BA windows/sec (50 KF, 10k pts, 10 LM iters) + ORB matches/sec; 1/2/4/8 GPU",
        "value": value, "unit": "windows/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: synthetic stereo local BA, 50 optimisable + 10 fixed KF, "
                               "10k landmarks (~75k stereo edges), optimize(10) with the reference's stop rules",
                   "windows_per_gpu": args.windows, "global_windows": args.windows * n_gpus, "streams_per_gpu": args.streams,
                   "parallelism": f"independent windows, w mod {n_gpus}", "lm_iterations_mean": float(np.mean([r.iterations for r in res])),
                   "lm_trials_mean": float(np.mean([r.trials for r in res]))},
        "roofline": roofline,
        "kernels": kernels, "step_times": steps,
        "whole_job_alg_GBps_per_gpu": whole_bytes / (ms_per_step * 1e-3) / 1e9,
        "upload_s_per_batch": lba_out["upload_s"],
        "ranks_seen": ranks_seen, "windows_per_s_by_rank": per_rank,
    }
    if sweep_out:
        out["config2_batch_sweep"] = sweep_out
        if "1" in sweep_out:
            out["single_window_latency_ms"] = {"local_ba_optimize": sweep_out["1"]["optimize_ms"], "local_ba_solve_call": sweep_out["1"]["solve_call_ms"]}
    if e2e_out is not None:
        e2e_windows = args.windows * n_gpus * e2e_out["n_batches"]
        out["value_end_to_end"] = e2e_windows / e2e_out["elapsed"]
        out["end_to_end"] = {"what": "osh_lba_upload (float32 staging of the caller's arrays + H2D + sort / Schur plan / record build as HIP kernels) + "
                                     "osh_lba_optimize + osh_lba_download of whole batches of host-resident windows; several solver contexts, each "
                                     "driven by its own host thread, form a three-stage pipeline (one batch uploading, one optimising, one "
                                     "downloading at a time)",
                             "contexts": e2e_out["n_ctx"],
                             "batches": e2e_out["n_batches"], "windows_per_batch": args.windows, "ms_per_batch": e2e_out["elapsed"] / e2e_out["n_batches"] * 1e3,
                             "fraction_of_resident": (e2e_windows / e2e_out["elapsed"]) / value,
                             "one_batch_serial_ms": e2e_out["serial_ms"], "pack_ms_mean": e2e_out["pack_ms_mean"], "copy_ms_mean": e2e_out["copy_ms_mean"],
                             "upload_threads_per_context": e2e_out["upload_threads"]}
    if orb_out is not None:
        out["orb"] = {"metric": "ORB matches/sec (SearchByProjection 256-bit Hamming, 2000x2000 per frame pair)",
                      "matches_per_s": orb_out["matches_per_s"], "pair_evals_per_s": orb_out["pair_evals_per_s"],
                      "frame_pairs_per_s": orb_out["frame_pairs_per_s"], "accepted_per_pair": orb_out["accepted_per_pair"],
                      "kernel_ms_per_launch": orb_out["kernel_ms"], "occupancy_rounds_ms_per_call": orb_out["resolve_ms"], "occupancy_rounds": orb_out["rounds"],
                      "pairs_per_gpu": orb_out["pairs_per_gpu"], "dtype": "u32 popcount",
                      "what": "osh_orb_match_local_points per step: unrestricted search + the sequential slot occupancy of src/ORBmatcher.cc:84-139 "
                              "resolved in fixed-point rounds on the device + D2H of the slot assignment; matches are the function's return values"}
        # integer-VALU bound: a 256-bit Hamming distance is 8 v_xor_b32 + 8 v_bcnt_u32_b32 (the add is folded into bcnt) per pair;
        # the best / second-best bookkeeping on top is overhead.  Peak = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz (the FP32 vector
        # peak of MI355X_MICROARCH.md, 157.3 TFLOP/s, is this rate x 2 flop x 2 packed).  Lane-ops per pair: 19.25 = the 77 vector
        # instructions of the loop body over 4 train descriptors (8 xor + 8 chained v_bcnt + v_lshl_or + v_med3 + v_min each, one
        # v_mov; round 2: 23.4 by SQ_INSTS_VALU, the counts summed by a tree of v_add3 and the second-best kept by max + min).
        pe = orb_out["pairs_per_gpu"] * 4.0e6 / (orb_out["kernel_ms"] * 1e-3)
        out["orb"]["roofline"] = {"bound": "valu_int", "kernel": "k_orb_bruteforce", "achieved": pe * 16 / 1e12, "peak": 39.3,
                                  "unit": "T lane-ops/s", "frac": pe * 16 / 39.3e12, "ops_per_pair_eval_algorithmic": 16,
                                  "ops_per_pair_eval_measured": 19.25, "frac_of_issue_slots_measured": pe * 19.25 / 39.3e12,
                                  "pair_evals_per_s_kernel": pe, "traffic": measured_other_traffic("k_orb_bruteforce", orb_out["pairs_per_gpu"], 64)}
        if orb_sweep_out:
            out["orb"]["config3_batch_sweep"] = orb_sweep_out
        if orb_out.get("sbp"):
            out["orb"]["search_by_projection_end_to_end"] = orb_out["sbp"]
        if orb_out.get("fuse"):
            out["orb"]["fuse_end_to_end"] = orb_out["fuse"]
    if inertial_out is not None:
        out["inertial"] = inertial_out
        out.setdefault("single_window_latency_ms", {})["local_inertial_ba_call"] = inertial_out["single_window_latency_ms"]
    if info.rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(lba_out["windows"])
        # like for like: the CPU path builds its structure inside the timed call, so the speed-up is quoted on the end-to-end rate
        out["speedup_vs_cpu_1thread"] = (out.get("value_end_to_end") or value) / out["cpu_baseline"]["value"]
        out["speedup_vs_cpu_1thread_resident_loop_only"] = value / out["cpu_baseline"]["value"]
        out["cpu_baseline"]["flags"] = "gcc -O3 -march=native -ffp-contract=off (the reference builds with -O3 -march=native, which allows contraction)"
        out["cpu_baseline"]["nproc"] = os.cpu_count()
        out["cpu_baseline"]["usable_cores"] = usable_cores()
        out["cpu_baseline"]["cpu_model"] = cpu_model()
        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(lba_out["windows"])
        out["speedup_vs_cpu_all_cores"] = (out.get("value_end_to_end") or value) / out["cpu_baseline_all_cores"]["value"]
        if orb_out is not None:
            from oracle import binding as ob
            p = orb_out["pair0"]
            t0 = time.perf_counter()
            ob.orb_search(p.query_desc, p.train_desc, p.train_level)
            dt = time.perf_counter() - t0
            out["orb"]["cpu_baseline"] = dict(value=4.0e6 / dt, unit="pair-evals/s", cores=1, kind="port",
                                              sample="one 2000x2000 frame pair, SWAR popcount restatement")
    if info.rank == 0:
        print(json.dumps(out))
    osh_dist.finalize(info)


if __name__ == "__main__":
    main()
