#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_kernels.sh into profiles/rNN_pmc_kernels.json (NN = $GDL_ROUND, default 03) + a readable table."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
extra = sys.argv[2:]


def arg(name, default):
    return extra[extra.index(name) + 1] if name in extra else default


cnt = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> counter -> sum
launches = collections.defaultdict(int)
dur = collections.defaultdict(float)
nsteps = 0
for d in sorted(glob.glob(out + "/*/")):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    first_counter = None
    for f in files:
        rows = list(csv.DictReader(open(f)))
        names = sorted({r["Counter_Name"] for r in rows})
        first_counter = names[0] if names else None
        nsteps = max(nsteps, sum(1 for r in rows if "sgd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == first_counter))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE":
                launches[k] += 1
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
nsteps = max(nsteps, 1)
kernels = {}
tot_r = tot_w = 0.0
for k, c in cnt.items():
    n = max(launches[k], 1)
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reads half the bytes of a 16-byte-per-lane stream on gfx950
    # (MI355X_MICROARCH.md, HBM): doubled
    rd, wr = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024 / n, c.get("WRITE_SIZE", 0.0) * 1024 / n
    tot_r += rd * n / nsteps
    tot_w += wr * n / nsteps
    e = {"launches_per_step": round(n / nsteps, 2), "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
         "hbm_bytes_per_launch": round(rd + wr), "avg_us_under_pmc": round(dur[k] / n, 2) if dur[k] else None}
    if c.get("SQ_BUSY_CYCLES"):
        # MFMA busy cycles are summed over SIMDs; SQ_BUSY_CYCLES over shader engines -- report the raw ratio inputs
        e["mfma_busy_cycles_per_launch"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n)
        e["wave_cycles_x4_per_launch"] = round(4 * c.get("SQ_WAVE_CYCLES", 0.0) / n)
        e["gui_active_per_launch"] = round(c.get("GRBM_GUI_ACTIVE", 0.0) / n)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3)
            e["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
            e["active_inst_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
            e["mfma_frac_of_wave_cycles"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * wc), 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    kernels[k] = e
import bench  # noqa: E402  (src_hash)

doc = {"src_hash": bench.src_hash(), "dtype": arg("--dtype", "bf16"), "batch": int(arg("--batch", "64")),
       "workload": arg("--workload", "cremad"), "steps_seen": nsteps,
       "command": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-f32 "
                  + " ".join(extra),
       "note": "separate --pmc passes; under counter collection kernels run serialised, so per-launch numbers are each kernel "
               "ALONE; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (16-byte streaming reads)",
       "step_totals": {"hbm_read_GB": round(tot_r / 1e9, 3), "hbm_write_GB": round(tot_w / 1e9, 3)},
       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"]))}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "r%s_pmc_kernels.json" % os.environ.get("GDL_ROUND", "03")), "w"), indent=1)
print(f"steps seen {nsteps}; per step: read {tot_r / 1e9:.2f} GB, write {tot_w / 1e9:.2f} GB")
for k, e in list(doc["kernels"].items())[:30]:
    print(f"  {k[:60]:60s} n/step {e['launches_per_step']:5.1f}  rd {e['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB  wr "
          f"{e['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB  mfma/wave {e.get('mfma_frac_of_wave_cycles')}  wait {e.get('wait_any_frac')}")
