"""Runs the FullInertialBA / MergeInertialBA host tests four times in one process (their reference varies from run to run with the
heap addresses of the map, see the test's docstring): `python profiles/repeat_check.py` on the GPU box."""
import pytest
rcs = [int(pytest.main(["tests/test_gpu_host.py", "-q", "-m", "gpu", "-k", "full_inertial or merge_inertial", "-p", "no:cacheprovider"])) for _ in range(4)]
print("RCS", rcs)
raise SystemExit(max(rcs))
