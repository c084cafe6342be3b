"""GPU box probe behind bench.py's files -> proof leg: where should the trace generator's output columns live?  The C++ generator
(256 host threads) writes 4.8 GB per 2^20-step starknet trace; candidates: (a) torch pinned tensors (hipHostMalloc: one NUMA
node), (b) ordinary pages first touched by the generator's own threads (spread over the nodes), then pinned in place
(hipHostRegister).  Prints generation and upload times of both.  usage: python tools/e2e_probe.py [log_steps]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sandstorm_amd import binary, examples, hostlib   # noqa: E402

log_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 16 << log_steps
states, memory, xpi = examples.starknet_example(log_steps)
tb, mb = binary.write_register_states(states), binary.write_memory(memory)
del states, memory
dev = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(9)]


def run(name, views, tensors):
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hostlib.starknet_base_trace(tb, mb, xpi, out=views)
        t1 = time.perf_counter()
        for c in range(9):
            dev[c].copy_(tensors[c], non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            print("%-28s trace %.3f s   upload %.3f s" % (name, t1 - t0, t2 - t1))


pinned = [torch.empty((n, 4), dtype=torch.int64).pin_memory() for _ in range(9)]
run("hipHostMalloc (torch pinned)", [t.numpy().view("uint64") for t in pinned], pinned)
del pinned
plain = [np.empty((n, 4), dtype=np.uint64) for _ in range(9)]         # untouched pages: the generator's threads touch them first
hostlib.starknet_base_trace(tb, mb, xpi, out=plain)
rt = torch.cuda.cudart()
for a in plain:
    err = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    assert int(err) == 0, err
run("first touch + hipHostRegister", plain, [torch.from_numpy(a.view("int64")) for a in plain])
unreg = [np.empty((n, 4), dtype=np.uint64) for _ in range(9)]
hostlib.starknet_base_trace(tb, mb, xpi, out=unreg)
run("first touch, pageable", unreg, [torch.from_numpy(a.view("int64")) for a in unreg])
