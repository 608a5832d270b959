"""world_size-2 gloo test of the multi-GPU plumbing: windows shard w mod G, no data-path
collective, barrier + max-over-ranks timing + summed work are what bench.py's N>1 run uses."""
import json
import socket
import subprocess
import sys
from pathlib import Path

from orb_slam3_study_kr_amd import dist as osh_dist

ROOT = Path(__file__).resolve().parent.parent


def test_shard_indices_partition_the_units():
    for world in (1, 2, 3, 8):
        shards = [osh_dist.shard_indices(21, r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(21))
        assert all(all(w % world == r for w in s) for r, s in enumerate(shards))


def test_two_rank_gloo_run(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "dist"
    env = dict(**__import__("os").environ, OSH_DIST_OUT=str(out))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.loads(Path(f"{out}.{k}").read_text()) for k in range(2)]
    assert [x["rank"] for x in res] == [0, 1] and all(x["world"] == 2 for x in res)
    assert res[0]["mine"] == [0, 2, 4, 6, 8] and res[1]["mine"] == [1, 3, 5, 7, 9]
    for x in res:
        assert abs(x["t_max"] - 0.2) < 1e-12            # MAX over ranks
        assert x["tot"] == [10.0, 45.0]                 # whole-job totals
