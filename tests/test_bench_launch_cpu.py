"""bench.py's N>1 launch path (BASELINE.json configs[4]) on CPU: `python bench.py --gpus 2` outside a rendezvous must start
two ranks as a child process and report n_gpus == 2; a rendezvous of the wrong size must fail instead of printing a line."""
import json
import os
import subprocess
import sys
from pathlib import Path

from orb_slam3_study_kr_amd import launch

ROOT = Path(__file__).resolve().parent.parent


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_needs_spawn_only_outside_a_rendezvous():
    assert launch.needs_spawn(2, {}) and launch.needs_spawn(8, {"PATH": "x"})
    assert not launch.needs_spawn(1, {})
    assert not launch.needs_spawn(2, {"WORLD_SIZE": "2", "RANK": "0"})


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    cmd = launch.launch_command("bench.py", 4, ["--gpus", "4", "--steps", "2"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "2"]


def test_bench_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "5",
                        "--stub-solver"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["stub"] is True
    assert out["config"]["global_windows"] == 10           # 5 windows per rank, both ranks counted
    assert abs(out["value"] - 10 / (out["ms_per_step"] * 1e-3)) < 1e-6
    # what the process group itself saw: an all-reduce of 1 counted both ranks, the all-gather returned one rate per rank
    assert out["ranks_seen"] == 2
    assert len(out["windows_per_s_by_rank"]) == 2 and all(v > 0 for v in out["windows_per_s_by_rank"])


def test_wrong_world_size_fails_loudly():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--stub-solver"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
