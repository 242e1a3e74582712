"""Condense gpurun_out/prof_<tag>/ (made by profile_bench.sh) into profiles/<name>/:
  bench.json, kernel_stats.csv, pass_span_from_trace.json, pmc_per_launch.json (every kz_ kernel launch of the first bench step:
  counters, lanes, VALU busy), pmc_sum_over_one_pass.json (per kernel, summed over the launches of ONE pass),
and profiles/pmc_latest.json: the counter-derived facts bench.py quotes, stamped with the commit and the profile directory
(bench.py withholds them when its live per-pass time differs from the profiled build's by more than 5 %).
HBM bytes: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE.

    python scripts/summarize_profile.py <gpurun tag> <profile dir name>"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", name)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
st = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if st:
    shutil.copy(st[0], os.path.join(dst, "kernel_stats.csv"))
# With two passes in flight the kernel durations of the trace overlap; the per-pass figure comparable with bench.py's
# roofline.pass_ms_in_flight is the steady-state distance between the ends of consecutive film kernels (one per pass).
tr = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
if tr:
    ends = sorted(int(r["End_Timestamp"]) for r in csv.DictReader(open(tr[0])) if r["Kernel_Name"].replace("void ", "").startswith("kz_film_taps"))
    gaps = sorted((b - a) / 1e6 for a, b in zip(ends, ends[1:]))
    if gaps:
        med = gaps[len(gaps) // 2]
        loop = [g for g in gaps if g <= 2 * med]        # the counting legs after the loop (megakernel: hundreds of ms) are left out
        json.dump({"film_kernels": len(ends), "mean_ms_between_pass_ends": round(sum(loop) / len(loop), 3), "gaps_used": len(loop),
                   "note": "rocprofv3 --kernel-trace of bench.py --steps 4 --warmup 1: time from the end of one pass (its kz_film_taps) to the end of "
                           "the next, averaged over the timed loop; comparable with roofline.pass_ms_in_flight of bench.json"},
                  open(os.path.join(dst, "pass_span_from_trace.json"), "w"), indent=1)
        print("pass ends: n %d mean gap %.3f ms over %d gaps" % (len(ends), sum(loop) / len(loop), len(loop)))

def short(k):
    k = k.split("(")[0].replace("void ", "")
    return k
disp = defaultdict(dict); names = {}
def newest(pattern):
    """One file per counter group: a tag profiled twice leaves both runs' CSVs behind, and summing them doubles every count."""
    groups = defaultdict(list)
    for f in glob.glob(pattern):
        groups[os.path.dirname(f)].append(f)
    return [max(fs, key=os.path.getmtime) for fs in groups.values()]
for f in newest(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k.startswith("kz_"):
            continue
        i = int(row["Dispatch_Id"])
        disp[i][row["Counter_Name"]] = disp[i].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        names[i] = k
ids = sorted(disp)
# bench.py --steps 1 --warmup 0: the first kz_wf_generate .. kz_film_taps run is pass 1 of the step
first = next((i for i in ids if names[i] == "kz_wf_generate"), None)
last = next((i for i in ids if i > (first or 0) and names[i].startswith("kz_film_taps")), None)
one_pass = [i for i in ids if first is not None and last is not None and first <= i <= last]
def derived(c):
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    d = {"ms_at_2.4GHz": round(g / 2.4e6, 3)}
    if "SQ_ACTIVE_INST_VALU" in c and g > 0:
        d["valu_busy_formula"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / g, 3)
        d["lanes"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c["SQ_ACTIVE_INST_VALU"], 1), 1)
    if "SQ_INSTS_VALU" in c and g > 0:
        d["valu_wave_instr_per_cycle_per_simd"] = round(c["SQ_INSTS_VALU"] / 1024 / g, 4)
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) > 0:
        d["l1_hit"] = round(1.0 - c.get("TCP_TCC_READ_REQ_sum", 0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        d["wait_inst_any"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
        d["wait_any"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
    return d
json.dump([{"dispatch": i, "kernel": names[i], **derived(disp[i]), "counters": disp[i]} for i in one_pass],
          open(os.path.join(dst, "pmc_per_launch.json"), "w"), indent=1)
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(int)
for i in one_pass:
    k = names[i]
    launches[k] += 1
    for c, v in disp[i].items():
        acc[k][c] += v
out = {k: dict(v, launches=launches[k], **derived(v)) for k, v in sorted(acc.items())}
json.dump(out, open(os.path.join(dst, "pmc_sum_over_one_pass.json"), "w"), indent=1)
samples = bench["roofline"]["samples_per_launch"]
fetch = sum(v.get("FETCH_SIZE", 0) for v in out.values()); write = sum(v.get("WRITE_SIZE", 0) for v in out.values())
def group(prefix):
    ks = [k for k in out if k.startswith(prefix)]
    c = defaultdict(float)
    for k in ks:
        for n, v in out[k].items():
            if n == "launches" or (isinstance(v, float) and n.split("_")[0] in ("SQ", "TCC", "TCP", "GRBM", "FETCH", "WRITE")):
                c[n] += v
    return c
kernels = {}
for label, prefix in (("kz_wf_trace<0>", "kz_wf_trace<0"), ("kz_wf_trace<4> (shadow)", "kz_wf_trace<4"), ("kz_wf_trace<2> (deferred walk-through)", "kz_wf_trace<2"), ("kz_wf_shade", "kz_wf_shade"), ("kz_wf_trace_packet", "kz_wf_trace_packet"),
                      ("kz_wf_trace_list", "kz_wf_trace_list"), ("kz_wf_beam", "kz_wf_beam"), ("kz_film_resolve", "kz_film_resolve"), ("kz_film_taps", "kz_film_taps"), ("kz_film_apply", "kz_film_apply"), ("kz_wf_generate", "kz_wf_generate")):
    c = group(prefix)
    if not c.get("SQ_INSTS_VALU"):
        continue
    d = derived(c)
    kernels[label] = {"launches_per_pass": int(c["launches"]), "valu_wave_instr_per_sample": round(c["SQ_INSTS_VALU"] / samples, 3),
                      "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0) / c["SQ_INSTS_VALU"], 3), "lanes": d.get("lanes"),
                      "valu_busy_formula": d.get("valu_busy_formula"), "l1_hit": d.get("l1_hit"), "l2_hit": d.get("l2_hit"),
                      "ms_under_pmc_at_2.4GHz": d.get("ms_at_2.4GHz")}
try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception:
    commit = None
sys.path.insert(0, root)
from bench import source_hash
facts = {"profile": "profiles/" + name, "commit": commit, "source_sha16": source_hash(), "bench_value": bench["value"],
         "path_kernels_ms_one_pass_alone": bench["roofline"]["pass_ms_alone"], "samples_per_pass": samples,
         "hbm_bytes_per_sample": round((2 * fetch + write) * 1024 / samples, 1) if fetch else None,
         "fetch_size_kib_raw": fetch, "write_size_kib": write, "kernels": kernels,
         "note": "rocprofv3 --pmc, one counter group per run of `bench.py --steps 1 --warmup 0 --no-cpu-baseline`, summed over the launches of the "
                 "first pass; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (per-lane 16-B gathers are uncalibrated: raw kept)"}
json.dump(facts, open(os.path.join(root, "profiles", "pmc_latest.json"), "w"), indent=1)
for k, v in kernels.items():
    print("%-18s launches %d  VALU/sample %8.2f  lanes %s  busy %s  L1 %s  L2 %s" % (k, v["launches_per_pass"], v["valu_wave_instr_per_sample"], v["lanes"], v["valu_busy_formula"], v["l1_hit"], v["l2_hit"]))
print("HBM bytes/sample:", facts["hbm_bytes_per_sample"])
