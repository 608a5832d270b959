"""Starting the N ranks of a multi-GPU bench run (BASELINE.json configs[4]).

`python bench.py --gpus N` without a torch.distributed environment has to become N ranks, one per GPU.
The parent never touches the GPU and never exec()s: it starts `python -m torch.distributed.run` as a CHILD
process and returns the child's exit code.  When the environment already carries a rendezvous (the driver's
own `python -m torch.distributed.run ... bench.py --gpus N`), the rank count must equal --gpus.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def in_rendezvous(environ=None) -> bool:
    """True when this process was started by torch.distributed.run (or an equivalent launcher)."""
    environ = os.environ if environ is None else environ
    return "WORLD_SIZE" in environ and "RANK" in environ


def needs_spawn(gpus: int, environ=None) -> bool:
    return gpus > 1 and not in_rendezvous(environ)


def launch_command(script: str, gpus: int, argv: list[str], port: int | None = None) -> list[str]:
    """The child command: one rank per GPU of this node, rendezvous on 127.0.0.1 (the hostname may not resolve)."""
    port = free_port() if port is None else port
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]


def spawn_ranks(script: str, gpus: int, argv: list[str], timeout: float | None = None) -> int:
    """Runs the ranks as a child process group; rank 0's JSON line goes to our stdout.  Returns the child's rc."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(launch_command(script, gpus, argv), env=env, timeout=timeout)
    return int(proc.returncode)


def check_world(gpus: int, world: int) -> None:
    """A SCALE line must never silently be a different rank count than the one asked for."""
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but the rendezvous has WORLD_SIZE={world}; refusing to report a "
                         f"line for the wrong GPU count")
