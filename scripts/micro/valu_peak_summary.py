"""Joins valu_peak's own s_memtime figures with the rocprofv3 PMC counters of the same dispatches: for every (variant,
waves/SIMD) the profile formula  SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)  beside the true issue rate."""
import csv, glob, json, sys
out = sys.argv[1]
plain = json.load(open(out + "/plain.json"))
def rows(d):
    f = glob.glob(out + "/" + d + "/*/*_counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
acc = {}
for r in rows("pmc_sq") + rows("pmc_grbm"):
    if "valu_loop" not in r["Kernel_Name"] and "op_loop" not in r["Kernel_Name"]:
        continue
    acc.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
# dispatches come in the program's order: per (variant, wps) two launches (warm-up, timed); the two PMC runs number them alike
ids = sorted(acc)
res = []
for i, p in enumerate(plain["results"]):
    if 2 * i + 1 >= len(ids):
        break
    c = acc[ids[2 * i + 1]]
    e = dict(p)
    simd_cycles = 1024.0 * c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if simd_cycles > 0 and "SQ_ACTIVE_INST_VALU" in c:
        e["profile_formula_valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4)
        e["pmc_valu_wave_instr_per_cycle_per_simd"] = round(c["SQ_INSTS_VALU"] / simd_cycles, 4)
        e["active_inst_valu_quadcycles_per_valu_instr"] = round(c["SQ_ACTIVE_INST_VALU"] / max(1.0, c["SQ_INSTS_VALU"]), 4)
        e["lanes_per_valu_instr"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, c["SQ_ACTIVE_INST_VALU"]), 2)
    e["pmc"] = c
    res.append(e)
json.dump({"device": plain["device"], "results": res}, sys.stdout, indent=1)
