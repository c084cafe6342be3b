"""Times the extension-trace scans (csrc/ext.hip) at the starknet 2^20-step shape: memory product over 2^23 items,
range-check product over 2^22, diluted product and aggregate over 2^21.  SS_SCAN_LOG_CHUNK / SS_INV_LOG_CHUNK in the
environment select the chunk sizes.  Prints one line: ms per call (best of 5)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be  # noqa: E402

dev = torch.device("cuda", 0)
ctx = be.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 24
g = torch.Generator(device=dev)
g.manual_seed(1)
cols = torch.randint(0, 2**63 - 1, (3, n, 4), dtype=torch.int64, device=dev, generator=g)
cols[:, :, 3] &= (1 << 59) - 1
out = torch.zeros((n, 4), dtype=torch.int64, device=dev)
z, a = cols[0, 5].cpu().numpy().view("uint64"), cols[0, 9].cpu().numpy().view("uint64")


def timed(fn):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


res = {
    "mem 2^23": timed(lambda: ctx.permutation_product((cols[0], 2, 0, 1), (cols[1], 2, 0, 1), n // 2, z, a, out, 2, 0, want_last=False)),
    "rc 2^22": timed(lambda: ctx.permutation_product((cols[2], 4, 0, -1), (cols[2], 4, 2, -1), n // 4, z, None, out, 4, 1, want_last=False)),
    "dc 2^21": timed(lambda: ctx.permutation_product((cols[2], 8, 1, -1), (cols[2], 8, 5, -1), n // 8, z, None, out, 8, 7, want_last=False)),
    "agg 2^21": timed(lambda: ctx.diluted_aggregate(cols[2], 8, 5, n // 8, z, a, out, 8, 3)),
}
print("scan=%s inv=%s  " % (os.environ.get("SS_SCAN_LOG_CHUNK", "default"), os.environ.get("SS_INV_LOG_CHUNK", "default"))
      + "  ".join("%s %.3f ms" % kv for kv in res.items()) + "  total %.3f ms" % sum(res.values()))
