"""Which library faults at process exit under rocprofv3?  Runs one small piece of the path (argv[1]: none | lba | orb | liba | pose |
torch+lba), dumps /proc/self/maps from a Python atexit hook (it runs before the C exit handlers) so the program counters of the
fault report can be resolved to libraries, and exits."""
import atexit
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
what = sys.argv[1] if len(sys.argv) > 1 else "lba"
out = Path(sys.argv[2]) if len(sys.argv) > 2 else Path("/tmp/maps.txt")
atexit.register(lambda: out.write_text(open("/proc/self/maps").read()))
if what.startswith("torch"):
    import torch  # noqa: F401
    torch.cuda.is_available()
from orb_slam3_study_kr_amd import capi, lba, orb, synth  # noqa: E402
from orb_slam3_study_kr_amd import synth_inertial as si  # noqa: E402

lib = capi.load_library()
if what.endswith("lba") and not what.endswith("liba"):
    with lba.LbaSolver(0) as sv:
        sv.solve([synth.make_window(3, n_free=6, n_fixed=2, n_points=400, stereo=True)])
elif what == "orb":
    m = orb.OrbMatcher(0)
    m.upload([synth.make_orb_pair(7, 500, 500)])
    m.match()
    m.close()
elif what == "liba":
    with lba.LbaSolver(0) as sv:
        sv.solve_inertial([si.make_inertial_window(11)])
elif what == "pose":
    with lba.LbaSolver(0) as sv:
        sv.optimize_poses([synth.make_pose_frame(300)])
elif what == "devcount":
    lib.osh_device_count()
print("done", what, flush=True)
