"""Statements to prove, for the benchmark, the smoke test and the tests: seeded synthetic columns and the
reference's shipped example run (`example/` of the reference: cairo-run's trace.bin / memory.bin / public input of array-sum,
2^14 steps; kept as data under tests/golden/) re-declared for a layout and padded to any power of two of steps - the program ends
in `jmp rel 0`, so repeating its final state is a valid run.  Host-side input preparation: nothing here is on the hot path."""
import copy
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE_DIR = os.path.join(ROOT, "tests", "golden", "example")
EXAMPLE_PUBLIC_INPUT = os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json")

MASK64 = (1 << 64) - 1
SEED0 = 0x53414E4453544F52


def splitmix64_stream(seed, count):
    """vectorised SplitMix64: `count` successive outputs for `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_column(n, column_index=0, seed=SEED0):
    """(n,4) uint64: uniformly random-ish Montgomery images < p.

    4 SplitMix64 draws per element, top limb masked to 59 bits and elements
    >= p folded by clearing the top bits (valid Montgomery images are just
    integers < p, so any value < p is a legal element)."""
    raw = splitmix64_stream(seed ^ column_index, 4 * n).reshape(n, 4).copy()
    raw[:, 3] &= np.uint64((1 << 59) - 1)          # < 2^251 < p
    return raw


def load_run(example_dir=EXAMPLE_DIR, public_input_json=EXAMPLE_PUBLIC_INPUT):
    """the example run: register states, memory, public input"""
    from sandstorm_amd import binary, public_input
    with open(os.path.join(example_dir, "trace.bin"), "rb") as f:
        states = binary.read_register_states(f.read())
    with open(os.path.join(example_dir, "memory.bin"), "rb") as f:
        memory = binary.read_memory(f.read())
    pi = public_input.AirPublicInput.from_json(public_input_json)
    return states, memory, pi


def _move_heap(memory, new_base):
    """the program's one heap segment moved to new_base (behind the re-declared builtin segments), the pointers into it with it"""
    heap = [a for a in range(len(memory)) if memory[a] is not None and a > 1000]
    mem = list(memory) + [None] * (new_base + 16 - len(memory))
    for a in heap:
        mem[a] = None
    for a in heap:
        mem[new_base + a - heap[0]] = memory[a]
    for a in range(1000):
        if a < len(memory) and mem[a] is not None and heap[0] <= mem[a] <= heap[-1] + 1:
            mem[a] += new_base - heap[0]
    return mem


def recursive_example(log_steps):
    """the example run as a 2^log_steps-step statement of the recursive layout: padded with its final state, the builtin segments
    re-declared for that step count back to back behind the execution segment, the program's one heap segment moved behind them"""
    from sandstorm_amd.layouts import recursive as rec
    states, memory, pi = load_run()
    if (1 << log_steps) == len(states):
        return states, memory, pi
    assert (1 << log_steps) > len(states)
    states = list(states) + [states[-1]] * ((1 << log_steps) - len(states))
    pi = copy.deepcopy(pi)
    pi.n_steps = 1 << log_steps
    seg = dict(pi.memory_segments)
    addr = seg["execution"][1]
    seg["output"] = (addr, addr)
    for name, ratio, cells in (("pedersen", rec.PEDERSEN_BUILTIN_RATIO, 3), ("range_check", rec.RANGE_CHECK_BUILTIN_RATIO, 1),
                               ("bitwise", rec.BITWISE_RATIO, 5)):
        seg[name] = (addr, addr)                     # begin = stop: the program uses nothing of the segment
        addr += cells * (pi.n_steps // ratio)
    pi.memory_segments = seg
    return states, _move_heap(memory, addr), pi


def starknet_example(log_steps=17):
    """the example run re-declared for the starknet layout (>= 2^17 steps: its diluted check needs them): padded with its final
    state, the builtin segments laid out after the execution segment, the heap segment moved behind them"""
    from sandstorm_amd.layouts import starknet as sk
    states, memory, pi = load_run()
    states = list(states) + [states[-1]] * ((1 << log_steps) - len(states))
    pi.n_steps = 1 << log_steps
    spi = sk.example_public_input(pi)
    new_base = spi.memory_segments["poseidon"][0] + 6 * (pi.n_steps // sk.POSEIDON_RATIO)
    return states, _move_heap(memory, new_base), spi
