"""Condense gpurun_out/prof_<tag>/ (made by profile_bench.sh) into profiles/<name>/: bench.json, kernel_stats.csv,
pmc_sum_over_one_pass.json (per kernel: every counter summed over the launches of ONE bench pass) and profiles/traffic_latest.json
(HBM bytes per pass, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", name)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
# With two passes in flight the kernel durations of the trace overlap; the per-pass figure comparable with bench.py's
# roofline.kernel_ms is the steady-state distance between the ends of consecutive film kernels (one per pass).
tr = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
if tr:
    ends = sorted(int(r["End_Timestamp"]) for r in csv.DictReader(open(tr[0])) if r["Kernel_Name"].startswith("kz_film_gather"))
    gaps = sorted((b - a) / 1e6 for a, b in zip(ends, ends[1:]))
    if gaps:
        # the two passes of a call end close together (their film kernels are chained), so gaps alternate short / long: the mean over
        # the timed loop is the per-pass figure; the counting legs after the loop (megakernel: hundreds of ms) are left out
        med = gaps[len(gaps) // 2]
        loop = [g for g in gaps if g <= 2 * med]
        json.dump({"film_kernels": len(ends), "mean_ms_between_pass_ends": round(sum(loop) / len(loop), 3), "gaps_used": len(loop),
                   "note": "rocprofv3 --kernel-trace of bench.py --steps 4 --warmup 1: time from the end of one pass (its kz_film_gather) to the end of the next, "
                           "averaged over the timed loop; comparable with roofline.kernel_ms of bench.json"},
                  open(os.path.join(dst, "pass_span_from_trace.json"), "w"), indent=1)
        print("pass ends: n %d mean gap %.3f ms over %d gaps" % (len(ends), sum(loop) / len(loop), len(loop)))
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("kz_"):
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if (k, row["Dispatch_Id"]) not in seen:
            seen.add((k, row["Dispatch_Id"]))
    for k in {s[0] for s in seen}:
        launches[k] = max(launches[k], sum(1 for s in seen if s[0] == k))
out = {k: dict(v, launches=launches[k]) for k, v in sorted(acc.items())}
json.dump(out, open(os.path.join(dst, "pmc_sum_over_one_pass.json"), "w"), indent=1)
# the timed path only: no counting (STATS) variants; counters scaled to the launches of ONE pass (maxDepth = 5 on the bench workload)
per_pass = {"kz_wf_generate": 1, "kz_wf_primary_fix": 1, "kz_wf_trace<1, false, true>": 1, "kz_wf_trace<0, false, true>": 5, "kz_wf_trace<2, false, true>": 5,
            "kz_wf_shade<false, false>": 5, "kz_wf_shade<false, true>": 5, "kz_film_gather": 1}
wf = [k for k in out if k in per_pass]
for k in wf:
    f = per_pass[k] / max(1, out[k]["launches"])
    out[k] = {c: (v * f if c != "launches" else per_pass[k]) for c, v in out[k].items()}
json.dump({k: out[k] for k in wf}, open(os.path.join(dst, "pmc_sum_over_one_pass.json"), "w"), indent=1)
fetch = sum(out[k].get("FETCH_SIZE", 0) for k in wf)
write = sum(out[k].get("WRITE_SIZE", 0) for k in wf)
traffic = {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "fetch_size_kib_raw": fetch, "write_size_kib": write,
           "note": "sum over the wavefront path kernels of ONE pass (bench.json roofline.samples_per_launch samples), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; "
                   "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (our 16-B per-lane gathers are uncalibrated: raw figure kept)",
           "profile": "profiles/" + name}
json.dump(traffic, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
for k in wf:
    v = out[k]
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        busy = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
        lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v["SQ_ACTIVE_INST_VALU"], 1)
        print("%-34s launches %2d  ms %6.2f  VALU busy %4.0f%%  active lanes/instr %4.1f  VALU insts %6.0fM" %
              (k, v["launches"], v["GRBM_GUI_ACTIVE"] / 8 / 2.4e6, 100 * busy, lanes, v.get("SQ_INSTS_VALU", 0) / 1e6))
bound = {"profile": "profiles/" + name, "note": "VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU"}
for k in wf:
    v = out[k]
    if v.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_ACTIVE_INST_VALU" in v:
        bound[k] = {"valu_busy": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3),
                    "active_lanes_per_valu_inst": round(v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v["SQ_ACTIVE_INST_VALU"], 1), 1),
                    "l2_hit": round(v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)), 3)}
json.dump(bound, open(os.path.join(root, "profiles", "bound_latest.json"), "w"), indent=1)
print(json.dumps(traffic)[:200])
