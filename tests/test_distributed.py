"""N > 1 path on CPU: two gloo ranks run the row-block sharded commitment of
sandstorm_amd/sharding.py (sub-tree per rank, all-gather of the roots, top levels on the
host) and must reproduce the single-process Merkle root for every tree flavour."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_commit_gloo(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DIST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_partitions():
    from sandstorm_amd import sharding
    assert [len(p) for p in sharding.column_partition(10, 8)] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert sharding.row_block(1 << 20, 3, 8) == (3 << 17, 4 << 17)
    assert sharding.subtree_friendly_layers(22, 8) == 19 and sharding.subtree_friendly_layers(2, 8) == 0
