"""Per kernel: the hardware's VALU utilisation and effective clock from one rocprofv3 --pmc pass (tools/valu_busy.sh).
usage: python tools/valu_busy.py counter_collection.csv [kernel_trace.csv [stages.json workload commit]]
With a fifth argument: the per-STAGE sums (tools/collect_final.py's stage table) as JSON for bench.py's `roofline.alu`
(profiles/effective_clock_<workload>.json): the clock a stage's kernels actually ran at - the chip clocks to its power budget."""
import collections
import csv
import sys

SIMDS = 1024
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.defaultdict(dict)                    # dispatch id -> counter -> value
name = {}
for r in rows:
    d = r.get("Dispatch_Id") or r.get("Dispatch_ID")
    disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    name[d] = r["Kernel_Name"]
dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        d = r.get("Dispatch_Id") or r.get("Dispatch_ID")
        dur[d] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, cs in disp.items():
    k = name[d][:70]
    for c, v in cs.items():
        agg[k][c] += v
    agg[k]["_n"] += 1
    agg[k]["_ns"] += dur.get(d, 0.0)
print("%-72s %5s %10s %12s %12s %12s %8s %8s %8s" % ("kernel", "disp", "ms", "INSTS_VALU", "ACTIVE_VALU", "GUI_ACTIVE", "cyc/inst", "busy", "GHz"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
    iv, av, gui = a.get("SQ_INSTS_VALU", 0.0), a.get("SQ_ACTIVE_INST_VALU", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0)
    if not iv:
        continue
    # GRBM_GUI_ACTIVE as reported: summed over the 8 XCDs when the csv carries no per-instance rows -> shown raw, and /8 in the ratios
    gui1 = gui / 8.0
    print("%-72s %5d %10.3f %12.4g %12.4g %12.4g %8.2f %8.3f %8.3f" % (k, a["_n"], a["_ns"] / 1e6, iv, av, gui, 4.0 * av / iv,
          4.0 * av / (SIMDS * gui1) if gui1 else 0.0, gui1 / a["_ns"] if a["_ns"] else 0.0))

if len(sys.argv) > 3:
    import json
    STAGES = {"ntt_pass": ["ntt_pass_kernel"], "quotient": ["quotient_"], "deep": ["deep_kernel", "deep_rational", "ood_blocks", "ood_fold", "batch_inverse", "poly_reduce"],
              "hash_rows": ["keccak_rows", "blake2s_rows"], "merkle": ["_pairs_kernel", "pedersen_", "felt_pairs"], "fri_fold": ["fri_fold_kernel"],
              "extension_scans": ["scan_", "perm_", "dil_", "inverse_dense"]}
    ONE_TIME = ("pedersen_build_windows", "pedersen_join_halves")       # the window table's build: once per process, not a proof's work
    out = {}
    for stage, keys in STAGES.items():
        sel = [a for k, a in agg.items() if any(key in k for key in keys) and a.get("SQ_INSTS_VALU") and not any(o in k for o in ONE_TIME)]
        ns, gui, iv = sum(a["_ns"] for a in sel), sum(a.get("GRBM_GUI_ACTIVE", 0.0) for a in sel) / 8.0, sum(a["SQ_INSTS_VALU"] for a in sel)
        if ns and gui:
            out[stage] = {"kernel_ms_in_run": ns / 1e6, "grbm_gui_active_per_xcd": gui, "effective_clock_ghz": gui / ns,
                          "valu_wave_insts_in_run": iv, "cycles_per_valu_inst_per_simd": SIMDS * gui / iv}
    json.dump({"workload": sys.argv[4] if len(sys.argv) > 4 else None, "commit": sys.argv[5] if len(sys.argv) > 5 else None,
               "how": "one rocprofv3 pass `--pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace` "
                      "(tools/valu_busy.sh); effective clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs in the csv: / 8) / the kernels' traced duration",
               "stages": out}, open(sys.argv[3], "w"), indent=1)
