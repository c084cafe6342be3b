"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly): ONE
proof over 2, 4 and 8 ranks (torch.distributed / gloo, tests/dist_prove_worker_gpu.py) with every rank's kernels - column-sharded
extension, row-block hashing, constraint evaluation with its halo, DEEP on row blocks, the re-shards between them - executed by the
device code on the CPU: the bytes are the single-device proof's (tests/golden/), the real recursive AIR of the reference's example
included."""
import os

import pytest

from tests.test_gpu_sharded import GOLD, run_sharded_gpu


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,case", [("mini_proof_eth_log9.bin", "mini:9:4"), ("array_sum_recursive_eth.proof", "example")])
def test_sharded_proof_with_the_device_code_on_every_rank(world, name, case, tmp_path):
    with open(os.path.join(GOLD, name), "rb") as f:
        want = f.read()
    assert run_sharded_gpu(world, case, tmp_path, "gloo") == want
