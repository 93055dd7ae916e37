"""Turns the raw rocprofv3 output of tools/run_profiles.sh (gpurun_out/prof/...) into the files kept under profiles/:

  profiles/<round>_bench_train_b32_kernel_stats.csv      kernel-trace --stats of the training bench command
  profiles/<round>_bench_train_b32_one_stream_kernel_stats.csv  the same with RNET_WGRAD_STREAM=0 (one-stream backward:
                                                         kernel durations with the chip to themselves)
  profiles/<round>_bench_train1_infer30_kernel_stats.csv the same for the default bench command (train + inference)
  profiles/<round>_pmc_train_b32_per_kernel.csv          per kernel and counter: dispatches, mean, sum
  profiles/traffic.json                                  per kernel: HBM bytes per launch (FETCH_SIZE x2 per the gfx950
                                                         note in MI355X_MICROARCH.md + WRITE_SIZE, KB -> bytes) and the
                                                         MFMA-busy fraction of the SIMD cycles

python tools/summarize_profiles.py [--round r01] [--src gpurun_out/prof]"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    """device symbol without its argument list / `void` / anonymous-namespace prefix"""
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "prof"))
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    for sub, dst in (("train", f"{a.round}_bench_train_b32_kernel_stats.csv"),
                     ("train1s", f"{a.round}_bench_train_b32_one_stream_kernel_stats.csv"),
                     ("infer", f"{a.round}_bench_train1_infer30_kernel_stats.csv")):
        f = newest(os.path.join(a.src, sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(out, dst))
            print("copied", f, "->", dst)
    # round 5: BASELINE configs[3] / configs[4] (tools/run_profiles_r05.sh) — kernel stats + FETCH_SIZE per kernel
    for sub, dst in (("c3", f"{a.round}_train_1024_b16_kernel_stats.csv"), ("c4", f"{a.round}_effnet_b3_f16_kernel_stats.csv")):
        f = newest(os.path.join(a.src, sub, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(out, dst))
            print("copied", f, "->", dst)
    for sub, dst in (("c3_pmc_fetch", f"{a.round}_pmc_train_1024_b16_fetch_per_kernel.csv"),
                     ("c4_pmc_fetch", f"{a.round}_pmc_effnet_b3_f16_fetch_per_kernel.csv")):
        f = newest(os.path.join(a.src, sub, "**", "*counter_collection.csv"))
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        with open(os.path.join(out, dst), "w") as fo:
            fo.write("Kernel_Name,Counter_Name,Dispatches,Mean_KB,Sum_KB,HBM_read_bytes_per_launch(x2 gfx950 correction)\n")
            for k in sorted(acc, key=lambda k: -sum(acc[k])):
                v = acc[k]
                fo.write(f'"{k}",FETCH_SIZE,{len(v)},{sum(v) / len(v):.1f},{sum(v):.1f},{int(2000 * sum(v) / len(v))}\n')
        print("wrote", dst)
    for name in ("layers_b1.txt", "layers_b8.txt"):
        f = os.path.join(a.src, name)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(out, f"{a.round}_infer_{name}"))
    f = os.path.join(a.src, "bench_line.json")
    if os.path.exists(f) and os.path.getsize(f) > 100:
        shutil.copy(f, os.path.join(out, f"{a.round}_bench_line.json"))
    per = defaultdict(lambda: defaultdict(list))    # kernel -> counter -> values
    clock = defaultdict(list)                        # kernel -> (GRBM_GUI_ACTIVE, duration ns) per dispatch
    for sub in ("pmc_fetch", "pmc_write", "pmc_mfma"):
        f = newest(os.path.join(a.src, sub, "**", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                if dur > 0:
                    clock[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), dur))
    with open(os.path.join(out, f"{a.round}_pmc_train_b32_per_kernel.csv"), "w") as fo:
        fo.write("Kernel_Name,Counter_Name,Dispatches,Mean,Sum\n")
        for k in sorted(per):
            for c in sorted(per[k]):
                v = per[k][c]
                fo.write(f'"{k}",{c},{len(v)},{sum(v) / len(v)},{sum(v)}\n')
    traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / {SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE} in "
                       "three separate passes over `RNET_WGRAD_STREAM=0 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline` (training "
                       "step, B=32), tools/run_profiles.sh + tools/summarize_profiles.py; means over the launches of each "
                       "kernel; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md (128-B requests "
                       "tallied at 64 B), WRITE_SIZE uncalibrated; counter units KB -> bytes (x1000)",
               "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)",
               "kernels": {}}
    for k, c in per.items():
        e = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fm, wm = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            e.update(fetch_kb_mean=round(fm, 1), write_kb_mean=round(wm, 1), bytes_per_launch=int((2 * fm + wm) * 1000))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and sum(c["GRBM_GUI_ACTIVE"]) > 0:
            e["mfma_busy_frac_of_simd_cycles"] = round(sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) /
                                                       (sum(c["GRBM_GUI_ACTIVE"]) / 8 * 256 * 4), 4)
        if e and ("mfma" in k or "conv" in k or "wgrad" in k or "bn_" in k):
            traffic["kernels"][k] = e
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    # Effective core clock per kernel: GRBM_GUI_ACTIVE counts busy cycles summed over the 8 XCDs, so
    # GRBM_GUI_ACTIVE / 8 / (End_Timestamp - Start_Timestamp) is the clock a launch really ran at
    # (MI355X_MICROARCH.md, "DVFS give-back").  Launches shorter than 20 us are left out (timestamp granularity).
    clk = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over the one-stream training step "
                   "(tools/run_profiles.sh): per kernel, mean over its launches >= 20 us of GRBM_GUI_ACTIVE / 8 XCDs / dispatch "
                   "duration = effective core clock in GHz, next to the kernel's MFMA-busy fraction; the chip clocks to its "
                   "power budget: 2.4 GHz nominal",
           "kernels": {}}
    for k, v in clock.items():
        v = [(c, d) for c, d in v if d >= 20000]
        if not v:
            continue
        ghz = [c / 8.0 / d for c, d in v]
        e = {"launches": len(v), "effective_clock_ghz_mean": round(sum(ghz) / len(ghz), 3),
             "effective_clock_ghz_min": round(min(ghz), 3), "effective_clock_ghz_max": round(max(ghz), 3),
             "mean_duration_us": round(sum(d for _, d in v) / len(v) / 1e3, 1)}
        m = traffic["kernels"].get(k, {}).get("mfma_busy_frac_of_simd_cycles")
        if m is not None:
            e["mfma_busy_frac_of_simd_cycles"] = m
        clk["kernels"][k] = e
    json.dump(clk, open(os.path.join(out, f"{a.round}_clock_per_kernel.json"), "w"), indent=1)
    probe = os.path.join(a.src, "mfma_probe.json")
    if os.path.exists(probe):
        shutil.copy(probe, os.path.join(out, f"{a.round}_clock_mfma_probe.json"))
    print("kernels with counters:", len(traffic["kernels"]))


if __name__ == "__main__":
    main()
