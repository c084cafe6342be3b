"""gpurun_out/final/ (tools/final_round.sh) -> profiles/r01_final_*: bench JSON lines, kernel-trace summaries,
kernel stats, and the HBM-traffic tables + hbm_traffic_<workload>.json that bench.py reports as roofline.traffic.
Usage: python tools/collect_final.py [round tag, default r01_final]"""
import json
import os
import re
import shutil
import sys

import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    COMMIT = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
except OSError:
    COMMIT = None
SRC = os.path.join(ROOT, os.environ.get("COLLECT_SRC", os.path.join("gpurun_out", "final")))      # COLLECT_SRC: another run's directory
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def parse_pmc(path, counter):
    """tools/pmc_summary.py output -> {kernel: (dispatches, sum)}"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+disp=(\d+)\s*$", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            continue
        m = re.match(r"^\s+%s\s+mean\s+(\S+)\s+sum\s+(\S+)" % counter, line)
        if m and cur:
            out[cur[0]] = (cur[1], float(m.group(2)))
    return out


for name in sorted(os.listdir(SRC)):
    if name.startswith("bench_") and name.endswith(".json"):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, "%s_%s" % (TAG, name)))
for name in ("pytest_gpu.txt", "smoke.txt", "sq_counters_starknet_2p20.txt", "sq_counters_recursive_2p20.txt", "ubench.txt", "mfma_mulbench.txt", "mulbench.txt", "fma_mulbench.txt", "e2e_device.txt", "device_trace_kernel_stats.csv",
             "kernel_gaps_starknet_2p20.txt", "kernel_gaps_recursive_2p20.txt", "kernel_gaps_goldilocks_plain_2p20.txt"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, "%s_%s" % (TAG, name)))

for d in sorted(os.listdir(SRC)):
    if not d.startswith("prof_"):
        continue
    w = d[len("prof_"):]
    p = os.path.join(SRC, d)
    for src, dst in (("kernel_trace_summary.txt", "%s_kernel_trace_summary_%s.txt"), ("kernel_stats.csv", "%s_kernel_stats_%s.csv")):
        if os.path.exists(os.path.join(p, src)):
            shutil.copy(os.path.join(p, src), os.path.join(DST, dst % (TAG, w)))
    fpath, wpath = os.path.join(p, "pmc_fetch.txt"), os.path.join(p, "pmc_write.txt")
    if not (os.path.exists(fpath) and os.path.exists(wpath)):
        continue
    fetch, write = parse_pmc(fpath, "FETCH_SIZE"), parse_pmc(wpath, "WRITE_SIZE")
    rows = []
    for k in set(fetch) | set(write):
        disp = (fetch.get(k) or write.get(k))[0]
        rd = 2.0 * fetch.get(k, (0, 0.0))[1] * 1024        # MI355X_MICROARCH.md: FETCH_SIZE in KB, x2 on gfx950 for wide coalesced reads
        wr = write.get(k, (0, 0.0))[1] * 1024
        rows.append((k, disp, rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    txt = os.path.join(DST, "%s_hbm_traffic_%s.txt" % (TAG, w))
    with open(txt, "w") as f:
        f.write("# HBM-side traffic per kernel, %s bench, 2 proofs per run (python bench.py --workload %s --steps 1 --warmup 0 --no-cpu-baseline)\n" % (w, w))
        f.write("# separate passes: rocprofv3 --pmc FETCH_SIZE ... ; rocprofv3 --pmc WRITE_SIZE ...  (tools/profile_round.sh)\n")
        f.write("# bytes: read = 2 * FETCH_SIZE[KB] * 1024 (MI355X_MICROARCH.md gfx950 correction for wide coalesced reads), write = WRITE_SIZE[KB] * 1024 (uncalibrated);\n")
        f.write("# Infinity-Cache hits are counted (the guide), so re-reads served by the 256 MiB cache show up here\n")
        f.write("%-64s %5s %14s %14s %16s\n" % ("kernel", "disp", "read_GB", "write_GB", "GB_per_dispatch"))
        for k, disp, rd, wr in rows:
            f.write("%-64s %5d %14.3f %14.3f %16.4f\n" % (k[:60], disp, rd / 1e9, wr / 1e9, (rd + wr) / 1e9 / max(1, disp)))
    ntt = [r for r in rows if "ntt_pass_kernel" in r[0]]
    disp = sum(r[1] for r in ntt)
    total = sum(r[2] + r[3] for r in ntt)
    # per bench stage (bench.py stage_roofline reads `kernels`): the kernels whose launches the stage's HIP events bracket
    stages = {"ntt_pass": ["ntt_pass_kernel"], "quotient": ["quotient_"], "deep": ["deep_kernel", "ood_blocks", "ood_fold", "batch_inverse", "poly_reduce"],
              "hash_rows": ["keccak_rows", "blake2s_rows"], "merkle": ["_pairs_kernel", "pedersen_", "felt_pairs"], "fri_fold": ["fri_fold_kernel"]}
    per_stage = {}
    for stage, keys in stages.items():
        sel = [r for r in rows if any(k in r[0] for k in keys)]
        d, t = sum(r[1] for r in sel), sum(r[2] + r[3] for r in sel)
        if d:
            per_stage[stage] = {"dispatches": d, "bytes_per_launch": t / d, "bytes_per_proof": t / 2, "source": "profiles/%s_hbm_traffic_%s.txt" % (TAG, w)}
    with open(os.path.join(DST, "hbm_traffic_%s.json" % w), "w") as f:
        json.dump({"workload": w, "kernel": "ss::ntt_pass_kernel", "dispatches": disp, "bytes_per_launch": total / max(1, disp),
                   "bytes_per_proof": total / 2, "source": "profiles/%s_hbm_traffic_%s.txt" % (TAG, w), "commit": COMMIT, "kernels": per_stage}, f, indent=1)
    print(w, "ntt dispatches", disp, "GB/launch %.3f" % (total / max(1, disp) / 1e9))


# ---- the ALU roofline's measured half: SQ_INSTS_VALU per proof and stage (own --pmc pass: tools/final_round.sh), with the
# instruction-weighted issue cost of the stage's kernels from profiles/alu_model.json (tools/alu_model.py) -> alu_counters_<w>.json
STAGES = {"ntt_pass": ["ntt_pass_kernel"], "quotient": ["quotient_"], "deep": ["deep_kernel", "deep_rational", "ood_blocks", "ood_fold", "batch_inverse", "poly_reduce"],
          "hash_rows": ["keccak_rows", "blake2s_rows"], "merkle": ["_pairs_kernel", "pedersen_", "felt_pairs"], "fri_fold": ["fri_fold_kernel"],
          "extension_scans": ["scan_", "perm_", "dil_", "inverse_dense"]}
# once per process and device, not per proof (the Pedersen window table's two build kernels: 5e10 wave instructions that a run of two proofs
# would otherwise charge to the Merkle stage - the round-5 'merkle frac 1.79' of recursive_2p20 was this)
ONE_TIME = ("pedersen_build_windows", "pedersen_join_halves")
model_path = os.path.join(DST, "alu_model.json")
model_json = json.load(open(model_path)) if os.path.exists(model_path) else {}
model = model_json.get("kernels", {})


def parse_all(path):
    """tools/pmc_summary.py output -> {kernel: {counter: sum, "disp": n}}"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+disp=(\d+)\s*$", line)
        if m:
            cur = out.setdefault(m.group(1), {"disp": int(m.group(2))})
            continue
        m = re.match(r"^\s+(\w+)\s+mean\s+(\S+)\s+sum\s+(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
    return out


for w in ("starknet_2p20", "recursive_2p20"):
    src = os.path.join(SRC, "sq_counters_%s.txt" % w)
    if not os.path.exists(src):
        continue
    ks = parse_all(src)
    stages = {}
    for stage, keys in STAGES.items():
        sel = {k: v for k, v in ks.items() if any(key in k for key in keys) and "SQ_INSTS_VALU" in v and not any(o in k for o in ONE_TIME)}
        if not sel:
            continue
        valu = sum(v["SQ_INSTS_VALU"] for v in sel.values())
        cyc, priced = 0.0, 0.0
        per_kernel = {}
        for k, v in sel.items():
            hit = [m for name, m in model.items() if name[:60].startswith(k[:56]) or k[:56].startswith(name[:56])]
            c = hit[0]["weighted_cycles_per_inst"] if hit else None
            per_kernel[k] = {"dispatches": v["disp"], "SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "weighted_cycles_per_inst": c}
            if c:
                cyc += c * v["SQ_INSTS_VALU"]
                priced += v["SQ_INSTS_VALU"]
        stages[stage] = {"valu_wave_insts_per_proof": valu / 2, "weighted_cycles_per_inst": cyc / priced if priced else None,
                         "busy_cycles_per_proof": sum(v.get("SQ_BUSY_CYCLES", 0.0) for v in sel.values()) / 2,
                         "wave_cycles_per_proof": sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in sel.values()) / 2,
                         "wait_any_per_proof": sum(v.get("SQ_WAIT_ANY", 0.0) for v in sel.values()) / 2, "kernels": per_kernel}
    with open(os.path.join(DST, "alu_counters_%s.json" % w), "w") as f:
        json.dump({"workload": w, "commit": COMMIT, "proofs_in_run": 2, "source": "profiles/%s_sq_counters_%s.txt" % (TAG, w),
                   "model": "profiles/alu_model.json (tools/alu_model.py: static instruction mix x %s)" % model_json.get("rates_source"),
                   "cycles_are": model_json.get("cycles_are"),
                   "stages": stages}, f, indent=1)
    print(w, "alu counters:", {k: "%.3g" % v["valu_wave_insts_per_proof"] for k, v in stages.items()})


# ---- the clock each stage's kernels ran at (tools/valu_busy.sh's pass) -> profiles/effective_clock_<w>.json + the per-kernel table
for w in ("starknet_2p20", "recursive_2p20"):
    cc, kt = os.path.join(SRC, "valu_busy", "counters_%s.csv" % w), os.path.join(SRC, "valu_busy", "kernel_trace_%s.csv" % w)
    if os.path.exists(cc) and os.path.exists(kt):
        import subprocess
        with open(os.path.join(DST, "%s_valu_busy_%s.txt" % (TAG, w)), "w") as f:
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_busy.py"), cc, kt, os.path.join(DST, "effective_clock_%s.json" % w), w, str(COMMIT)],
                           stdout=f, check=True)
        print(w, "effective clocks written")
