"""The CPU port (oracle/: C + OpenMP, test infrastructure) on WHOLE proofs at the bench's full size, once, outside bench.py
(VERDICT r5 "next" 6a): what bench.py's cpu_baseline leg extrapolates to from its bounded sample, measured - the real AIR of the layout,
CLI-default options, synthetic columns, through the same Python host as the GPU runs (bench.python_host_proof) - together with the
sample sizes the leg uses, so that the n log n scaling it assumes can be read off.
python tools/cpu_full_size.py starknet 17 20   |   recursive 14 16 20      -> one line per size, flushed as it is known"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402  (sets OMP_NUM_THREADS to the CPUs the cgroup grants)
from oracle import oracle_py as oracle           # noqa: E402
from oracle.cpu_context import CpuContext        # noqa: E402

layout = sys.argv[1]
threads = int(os.environ.get("OMP_NUM_THREADS", bench.HOST_CPUS))
print("host: %d CPUs visible, quota %s, OMP threads %d; the port's Montgomery product: %.1f ns on one core (dependent chain)"
      % (os.cpu_count(), bench.HOST_CPUS, threads, min(oracle.mulmod_ns() for _ in range(3))), flush=True)
prev = None
for ls in [int(a) for a in sys.argv[2:]]:
    t0 = time.perf_counter()
    t = bench.python_host_proof(CpuContext(), layout, ls)
    lf = ls + 4
    line = "%s 2^%d steps: the port's whole proof %.2f s on %d threads (set-up + proof %.1f s)" % (layout, ls, t, threads, time.perf_counter() - t0)
    if prev:
        pls, pt = prev
        scale = float(1 << (ls - pls)) * (lf + 1) / (pls + 4 + 1)
        line += "; n log n from 2^%d would say %.2f s (x %.1f): measured / extrapolated = %.2f" % (pls, pt * scale, scale, t / (pt * scale))
    print(line, flush=True)
    prev = (ls, t)
