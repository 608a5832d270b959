"""Device packer of osh_lba_upload (csrc/lba_pack_device.hip): HIP-event times of its three kernels and of the H2D copy, the host
staging pass, and the whole upload, for 512 / 64 / 1 config-2 windows; the host packer (csrc/lba_pack.h) beside it."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from orb_slam3_study_kr_amd import lba, synth  # noqa: E402

base = [synth.make_config2(100 + k) for k in range(8)]
with lba.LbaSolver(0) as sv:
    sv.set_profiling(True)
    for n in (512, 64, 1):
        ws = [base[k % 8] for k in range(n)]
        probs, res, outs = sv.prepare(ws)
        for mode, name in ((0, "device"), (1, "host")):
            sv.set_pack_mode(mode)
            sv.upload_prepared(ws, probs)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                sv.upload_prepared(ws, probs)
            ms = (time.perf_counter() - t0) / reps * 1e3
            up = sv.upload_times()
            pp = sv.pack_profile()
            print(f"{n:4d} windows, {name:6s} packer: upload {ms:7.2f} ms  host pass {up['pack_ms']:6.2f} ms  rest {up['copy_ms']:6.2f} ms", end="")
            if mode == 0:
                print(f"  | H2D {pp['h2d_ms']:.2f} ms ({pp['staged_bytes'] / 1e6:.0f} MB)  k_pack_pre1 {pp['pre1_ms']:.3f}  k_pack_pre2 {pp['pre2_ms']:.3f}  k_pack_post {pp['post_ms']:.3f} ms")
                print("      kilo-cycles per phase (mean over windows):", pp["kcycles"])
            else:
                print()
