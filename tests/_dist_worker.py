"""Worker for tests/test_dist_cpu.py: the N>1 plumbing of bench.py on the gloo backend (no GPU)."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from orb_slam3_study_kr_amd import dist as osh_dist  # noqa: E402

info = osh_dist.init_from_env(backend="gloo")
n_total = 10
mine = osh_dist.shard_indices(n_total, info.rank, info.world)
osh_dist.barrier(info)
# every rank "solves" its shard: elapsed differs per rank, the job time is the max, the work is the sum
elapsed = 0.1 * (info.rank + 1)
t_max = osh_dist.all_reduce_max(info, elapsed)
tot = osh_dist.all_reduce_sum(info, [float(len(mine)), float(sum(mine))])
osh_dist.barrier(info)
out = dict(rank=info.rank, world=info.world, mine=mine, t_max=t_max, tot=tot)
Path(os.environ["OSH_DIST_OUT"] + f".{info.rank}").write_text(json.dumps(out))
osh_dist.finalize(info)
