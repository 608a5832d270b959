"""Python driver over the local-BA C-ABI (include/orbslam3_hip.h, osh_lba_*).

Thin plumbing only: every number is computed by the HIP kernels in csrc/lba_device.hip.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .synth import LbaResultArrays, LbaWindow


class LbaSolver:
    """Owns one ``osh_lba_ctx`` (one HIP device + stream)."""

    def __init__(self, device: int = 0):
        self.lib = capi.load_library()
        self.ctx = C.c_void_p()
        capi.check(self.lib.osh_lba_create(device, C.byref(self.ctx)), "osh_lba_create", self.lib)
        self._windows: list[LbaWindow] = []
        self._problems = None

    def close(self):
        if self.ctx:
            self.lib.osh_lba_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- batch life cycle -------------------------------------------------------------------
    def upload(self, windows: list[LbaWindow]):
        self._windows = list(windows)
        arr = (capi.LbaProblem * len(windows))()
        for i, w in enumerate(windows):
            arr[i] = w.as_struct()
        self._problems = arr
        capi.check(self.lib.osh_lba_upload(self.ctx, len(windows), arr), "osh_lba_upload", self.lib)

    def optimize(self):
        capi.check(self.lib.osh_lba_optimize(self.ctx), "osh_lba_optimize", self.lib)

    def set_pack_mode(self, mode: int):
        """0: uploads are packed by HIP kernels (default), 1: by host threads (csrc/lba_pack.h), -1: default rules."""
        capi.check(self.lib.osh_lba_set_pack_mode(self.ctx, mode), "osh_lba_set_pack_mode", self.lib)

    def pack_compare(self, windows) -> dict:
        """Packs ``windows`` with the device packer and the host packer and compares the two layouts byte by byte."""
        arr = (capi.LbaProblem * len(windows))()
        for i, w in enumerate(windows):
            arr[i] = w.as_struct()
        st = np.zeros(4, dtype=np.int64)
        capi.check(self.lib.osh_lba_pack_compare(self.ctx, len(windows), arr, capi.ptr(st, capi.c_int64_p)), "osh_lba_pack_compare", self.lib)
        return dict(bytes=int(st[0]), sections=int(st[1]), items=int(st[2]), records=int(st[3]))

    # -- the same life cycle on prepared ctypes arrays (bench.py's end-to-end pipeline: nothing but the C-ABI calls is timed) --
    def prepare(self, windows: list[LbaWindow]):
        """(problem array, result array, result holders) for upload_prepared / download_prepared."""
        n = len(windows)
        probs = (capi.LbaProblem * n)()
        for i, w in enumerate(windows):
            probs[i] = w.as_struct()
        outs = [LbaResultArrays(w) for w in windows]
        res = (capi.LbaResult * n)()
        for i, o in enumerate(outs):
            o.bind(res[i])
        return probs, res, outs

    def upload_prepared(self, windows, probs):
        self._windows = windows
        self._problems = probs
        capi.check(self.lib.osh_lba_upload(self.ctx, len(windows), probs), "osh_lba_upload", self.lib)

    def download_prepared(self, res):
        capi.check(self.lib.osh_lba_download(self.ctx, len(res), res), "osh_lba_download", self.lib)

    def upload_times(self) -> dict:
        ms = np.zeros(2, dtype=np.float64)
        capi.check(self.lib.osh_lba_get_upload_times(self.ctx, capi.ptr(ms, capi.c_double_p)), "osh_lba_get_upload_times", self.lib)
        return dict(pack_ms=float(ms[0]), copy_ms=float(ms[1]))

    def pack_profile(self) -> dict:
        ms = np.zeros(30, dtype=np.float64)
        capi.check(self.lib.osh_lba_get_pack_profile(self.ctx, capi.ptr(ms, capi.c_double_p)), "osh_lba_get_pack_profile", self.lib)
        names = ("histogram", "offsets", "scatter", "order", "units", "", "", "", "unit_keys", "grouping", "compaction", "key_sort", "ranks", "greedy_merge",
                 "place_units", "renumber", "chunks", "slot_bytes", "arena0", "contrib_counts", "items_records", "contrib_slots", "", "")
        return dict(h2d_ms=float(ms[0]), pre1_ms=float(ms[1]), pre2_ms=float(ms[2]), post_ms=float(ms[3]), staged_bytes=int(ms[4]), on_device=bool(ms[5]),
                    kcycles={n: round(float(c) / 1e3, 1) for n, c in zip(names, ms[6:]) if n})

    def download(self) -> list[LbaResultArrays]:
        n = len(self._windows)
        outs = [LbaResultArrays(w) for w in self._windows]
        arr = (capi.LbaResult * n)()
        for i, o in enumerate(outs):
            o.bind(arr[i])
        capi.check(self.lib.osh_lba_download(self.ctx, n, arr), "osh_lba_download", self.lib)
        return [o.read_scalars(arr[i]) for i, o in enumerate(outs)]

    def solve(self, windows: list[LbaWindow]) -> list[LbaResultArrays]:
        self.upload(windows)
        self.optimize()
        return self.download()

    # -- local inertial BA (one persistent block per window, csrc/liba_device.hip) ---------------
    def solve_inertial(self, windows):
        from .synth_inertial import LibaResultArrays
        n = len(windows)
        probs = (capi.LibaProblem * n)()
        for i, w in enumerate(windows):
            probs[i] = w.as_struct()
        outs = [LibaResultArrays(w) for w in windows]
        arr = (capi.LibaResult * n)()
        for i, o in enumerate(outs):
            o.bind(arr[i])
        capi.check(self.lib.osh_liba_solve(self.ctx, n, probs, arr), "osh_liba_solve", self.lib)
        return [o.read_scalars(arr[i]) for i, o in enumerate(outs)]

    def inertial_profile(self):
        """(blocks per window, shader-clock cycles per phase of window 0) of the last solve_inertial on this thread."""
        grp = C.c_int32(0)
        cyc = np.zeros(8, dtype=np.int64)
        capi.check(self.lib.osh_liba_get_profile(C.byref(grp), capi.ptr(cyc, capi.c_int64_p)), "osh_liba_get_profile", self.lib)
        names = ("linearise", "assembly", "dinv", "schur", "ldlt", "backsub", "errors", "outputs")
        return int(grp.value), dict(zip(names, (int(c) for c in cyc)))

    # -- parity / debug aids ------------------------------------------------------------------
    def linearize(self, window: int = 0) -> dict:
        w = self._windows[window]
        P, L, E = w.n_free, w.n_points, w.n_edges
        out = dict(Hpp=np.zeros((P, 6, 6)), bp=np.zeros((P, 6)), Hll=np.zeros((L, 3, 3)), bl=np.zeros((L, 3)),
                   Hpl=np.zeros((E, 6, 3)), chi2=np.zeros(E))
        rc = C.c_double(0)
        d = capi.c_double_p
        capi.check(self.lib.osh_lba_linearize(self.ctx, window, capi.ptr(out["Hpp"], d), capi.ptr(out["bp"], d),
                                              capi.ptr(out["Hll"], d), capi.ptr(out["bl"], d), capi.ptr(out["Hpl"], d),
                                              capi.ptr(out["chi2"], d), C.cast(C.byref(rc), d)),
                   "osh_lba_linearize", self.lib)
        out["robust_chi2"] = rc.value
        return out

    def debug_trial(self, window: int, lam: float):
        w = self._windows[window]
        n = 6 * w.n_free
        S, bs, x = np.zeros((n, n)), np.zeros(n), np.zeros(n + 3 * w.n_points)
        d = capi.c_double_p
        capi.check(self.lib.osh_lba_debug_trial(self.ctx, window, lam, capi.ptr(S, d), capi.ptr(bs, d), capi.ptr(x, d)),
                   "osh_lba_debug_trial", self.lib)
        return S, bs, x

    # -- profiling ------------------------------------------------------------------------------
    def optimize_poses(self, frames):
        """``osh_pose_optimize``: Optimizer::PoseOptimization's four rounds for every frame of the batch, one block per frame."""
        from .synth import PoseResultArrays
        n = len(frames)
        probs = (capi.PoseProblem * n)(*[f.as_struct() for f in frames])
        res = [PoseResultArrays(f) for f in frames]
        rs = (capi.PoseResult * n)()
        for r, a in zip(rs, res):
            a.bind(r)
        capi.check(self.lib.osh_pose_optimize(self.ctx, n, probs, rs), "osh_pose_optimize", self.lib)
        return [a.read_scalars(r) for r, a in zip(rs, res)]

    def optimize_poses_inertial(self, frames):
        """``osh_posei_optimize``: Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame for every frame of the batch."""
        from .synth_inertial import PoseiResultArrays
        n = len(frames)
        probs = (capi.PoseiProblem * n)(*[f.as_struct() for f in frames])
        res = [PoseiResultArrays(f) for f in frames]
        rs = (capi.PoseiResult * n)()
        for r, a in zip(rs, res):
            a.bind(r)
        capi.check(self.lib.osh_posei_optimize(self.ctx, n, probs, rs), "osh_posei_optimize", self.lib)
        return [a.read(r, f.mode) for r, a, f in zip(rs, res, frames)]

    def plan_stats(self) -> dict:
        st = np.zeros(8, dtype=np.int64)
        capi.check(self.lib.osh_lba_get_plan_stats(self.ctx, capi.ptr(st, capi.c_int64_p)), "osh_lba_get_plan_stats", self.lib)
        return dict(items=int(st[0]), sym_items=int(st[1]), mfma_per_pass=int(st[2]), useful_blocks=int(st[3]), contributions=int(st[4]),
                    reduce_entries=int(st[5]), records=int(st[6]), rhs_contributions=int(st[7]))

    def set_profiling(self, enable: bool):
        capi.check(self.lib.osh_lba_set_profiling(self.ctx, int(enable)), "osh_lba_set_profiling", self.lib)

    def profile(self) -> dict:
        """{short kernel name: (launches, total ms)} measured with HIP events on the solver's stream."""
        launches = np.zeros(capi.OSH_K_COUNT, dtype=np.int64)
        ms = np.zeros(capi.OSH_K_COUNT, dtype=np.float64)
        capi.check(self.lib.osh_lba_get_profile(self.ctx, capi.ptr(launches, capi.c_int64_p), capi.ptr(ms, capi.c_double_p)),
                   "osh_lba_get_profile", self.lib)
        return {capi.KERNEL_NAMES[k]: (int(launches[k]), float(ms[k])) for k in range(capi.OSH_K_COUNT)}


def kernel_symbol(short_name: str) -> str:
    """Device kernel name (as rocprofv3 prints it) behind a short name of ``capi.KERNEL_NAMES``."""
    lib = capi.load_library()
    return lib.osh_lba_kernel_name(capi.KERNEL_NAMES.index(short_name)).decode()
