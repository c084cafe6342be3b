"""The ALU roofline's static half: for every kernel of libsandstorm_hip.so, the vector-instruction mix of its gfx950 code
(llvm-objdump of the device code objects inside sandstorm_amd/_build/*.o) priced with the MEASURED issue cost of each mnemonic
(profiles/r04_ubench_instruction_rates.txt, tools/ubench.hip: cycles per wave-instruction per SIMD at the nominal 2.4 GHz) ->
profiles/alu_model.json: per kernel {valu, mads, weighted_cycles_per_inst}.  bench.py turns it, with the SQ_INSTS_VALU counts
of profiles/alu_counters_<workload>.json and its own HIP-event stage times, into `roofline.alu`:
    peak_wave_insts_per_s = 1024 SIMDs x 2.4e9 / weighted_cycles_per_inst,  frac = (SQ_INSTS_VALU per proof / stage seconds) / peak
The mix is the STATIC one of the kernel's text (every instruction once): the field kernels are straight-line bodies executed
whole, so it is close to the dynamic mix; what a launch really issued is the counter's business.
Usage: python tools/alu_model.py [ubench rates file]"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "sandstorm_amd", "_build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
DEFAULT_RATES = os.path.join(ROOT, "profiles", "r05_ubench_instruction_rates.txt")
if not os.path.exists(DEFAULT_RATES):
    DEFAULT_RATES = os.path.join(ROOT, "profiles", "r04_ubench_instruction_rates.txt")


OWN_CLOCK = [False]


def rates(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(v_\w+)\s+[\d.]+ ms\s+[\d.]+ T lane-ops/s\s+~\s*([\d.]+) cyc", line)
        own = re.search(r"own clock [\d.]+ GHz\s+->\s*([\d.]+) cyc", line)          # round 5: the probe stamps its own clock
        if m and m.group(1) != "v_cndmask_b32":      # (that probe chains on VCC and measures the hazard, not the issue cost: priced as a simple op)
            out[m.group(1)] = float(own.group(1)) if own else float(m.group(2))
            OWN_CLOCK[0] = OWN_CLOCK[0] or bool(own)
    return out


def cost_of(mnemonic, table):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mnemonic)
    if base in table:
        return table[base]
    simple = table.get("v_add_u32", 3.06)
    if base.startswith(("v_mad_u64", "v_mad_i64")):
        return table["v_mad_u64_u32"]
    if base.startswith(("v_mul_lo", "v_mul_hi")):
        return table.get("v_mul_lo_u32", 4.5)
    if base.endswith(("_b64", "_u64", "_i64", "_f64")):
        return table.get("v_lshl_add_u64", 4.6)
    if base.startswith(("v_alignbit", "v_perm", "v_bitop3", "v_bfi")):
        return table.get("v_alignbit_b32", 4.6)
    return simple                       # add / sub / logic / shifts / moves / compares / selects of 32 bits


def kernels_of(obj, tmp):
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], capture_output=True, text=True)
    co = [f for f in os.listdir(tmp) if f.startswith(os.path.basename(obj) + ".") and "gfx950" in f]
    out = {}
    for f in co:
        txt = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                out.setdefault(name, collections.Counter())
                continue
            m = re.match(r"^\s+(v_\w+)\b", line)
            if m and name:
                out[name][m.group(1)] += 1
        os.remove(os.path.join(tmp, f))
    os.remove(dst)
    return out


def main():
    table = rates(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_RATES)
    model = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(os.listdir(BUILD)):
            if not obj.endswith(".o"):
                continue
            for name, mix in kernels_of(os.path.join(BUILD, obj), tmp).items():
                valu = sum(mix.values())
                if valu < 200 or name.endswith(".kd"):
                    continue
                cyc = sum(n * cost_of(mn, table) for mn, n in mix.items())
                model[name] = {"object": obj, "valu": valu, "mads": sum(n for mn, n in mix.items() if mn.startswith("v_mad_u64")),
                               "weighted_cycles_per_inst": round(cyc / valu, 4)}
    out = {"rates_source": os.path.relpath(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_RATES, ROOT), "rates": table, "simds": 1024,
           "cycles_are": "cycles of the probe's own clock (s_memtime / s_memrealtime inside tools/ubench.hip)" if OWN_CLOCK[0] else "cycles at a nominal 2.4 GHz",
           "kernels": model}
    with open(os.path.join(ROOT, "profiles", "alu_model.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k in sorted(model, key=lambda k: -model[k]["valu"])[:14]:
        print("%-90s valu %7d mads %6d  %.3f cyc/inst" % (k[:90], model[k]["valu"], model[k]["mads"], model[k]["weighted_cycles_per_inst"]))


if __name__ == "__main__":
    main()
