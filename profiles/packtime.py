"""Host-side cost of osh_lba_upload's packer (csrc/lba_pack.h) on this machine's cores: ms per window per thread at 1 / 8 / 16 threads
over 128 config-2 windows (osh_lba_pack_check times the second, warm pass)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from orb_slam3_study_kr_amd import synth  # noqa: E402
from test_lba_pack_cpu import pack_check  # noqa: E402

ws = [synth.make_config2(100 + k) for k in range(8)] * 16
for nt in (1, 8, 16):
    for _ in range(2):
        rc, msg, st, ms = pack_check(ws, threads=nt)
        print(nt, "threads:", round(ms, 1), "ms,", round(ms / len(ws) * nt, 2), "ms/window/thread", flush=True)
