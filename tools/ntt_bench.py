"""NTT micro-benchmark: batch of C coset LDEs (iNTT n + NTT 2n), resident columns.
usage: python tools/ntt_bench.py [log_n] [cols] [reps]"""
import sys, time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sandstorm_amd import backend as be

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
C = int(sys.argv[2]) if len(sys.argv) > 2 else 9
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << log_n
ctx = be.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(1)
src = torch.randint(0, 1 << 62, (C, n, 4), dtype=torch.int64, device="cuda", generator=g)
src[:, :, 3] &= (1 << 58) - 1
cols = [src[c].data_ptr() for c in range(C)]
offset = be.felt(3)
ev = [ctx.alloc(64 * n) for _ in range(C)]
co = [ctx.alloc(32 * n) for _ in range(C)]
for it in range(reps + 1):
    if it == 1:
        torch.cuda.synchronize(); ctx.profile(2 if os.environ.get("NTT_BENCH_CLOCK") else True); ctx.profile_reset(); t0 = time.perf_counter()
    ctx.lde(cols, log_n, 1, offset, ev, co)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
ms, launches = ctx.profile_read(be.PROF_NTT_PASS)
bfly = C * (n // 2 * log_n + n * (log_n + 1))
clock = ""
if os.environ.get("NTT_BENCH_CLOCK"):          # the shader clock the chip granted the transform kernels in THIS run (ss_profile_read_clock)
    cyc, ref = ctx.profile_read_clock(be.PROF_NTT_PASS)
    clock = ", %.3f GHz" % (cyc / ref * 0.1) if ref > 0 else ", clock not read"
print("log_n=%d cols=%d: %.2f ms per batch LDE (kernels %.2f ms, %d launches), %.1f G butterflies/s, %.1f Gfield-ops/s"
      % (log_n, C, dt * 1e3, ms / reps, launches // reps, bfly / (ms / reps * 1e-3) / 1e9, 3 * bfly / (ms / reps * 1e-3) / 1e9) + clock)
