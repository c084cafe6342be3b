"""Host-only: times the C++ base-trace generator (host/trace_{starknet,recursive}.cpp) on the padded example run - no GPU involved.
python tools/time_trace.py [starknet|recursive] [log2 steps]; SSH_TRACE_TIMING=1 prints the generator's own section times on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sandstorm_amd import hostlib, examples, binary
layout = sys.argv[1] if len(sys.argv) > 1 else "starknet"
log_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if layout == "starknet":
    states, memory, pi = examples.starknet_example(log_steps); nb = 9
else:
    states, memory, pi = examples.recursive_example(log_steps); nb = 7
tb, mb = binary.write_register_states(states), binary.write_memory(memory)
del states, memory
n = 16 << log_steps
cols = [np.empty((n, 4), dtype=np.uint64) for _ in range(nb)]
for it in range(4):
    t0 = time.perf_counter()
    hostlib.base_trace_with_callback(layout, tb, mb, pi, None, cols, None)
    print("trace_gen %.4f s" % (time.perf_counter() - t0), flush=True)
