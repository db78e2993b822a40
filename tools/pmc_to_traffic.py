#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes of tools/final_profile.sh (--pmc FETCH_SIZE, --pmc WRITE_SIZE, separate
runs, `python bench.py --iters 20 --steps 1 --warmup 0 --no-cpu-baseline`) into
  profiles/<tag>_pmc_fetch_write_kb_per_launch.csv   per-kernel launch averages, and
  profiles/<round>_traffic.json                      HBM bytes per launch for bench.py's `roofline.traffic`,
with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: (2*FETCH_SIZE + WRITE_SIZE) * 1024.

    python tools/pmc_to_traffic.py gpurun_out/final r02_final
"""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
traffic_name = tag.split("_")[0] + "_traffic.json"      # r02_final -> profiles/r02_traffic.json (read by bench.py)
PHASE = [("mlp_bwd_denc_kernel", "image_mlp_bwd_denc"), ("mlp_bwd_dw_kernel", "image_mlp_bwd_dw"),
         ("motion_warp_dft_kernel", "motion_warp_dft"), ("motion_warp_bwd_dft_kernel", "motion_warp_bwd"),
         ("hashgrid_fwd_kernel<2", "image_encode_fwd"), ("hashgrid_fwd_kernel<3", "motion_encode_fwd"),
         ("mlp_fwd_mfma_kernel<256", "image_mlp_fwd"), ("mlp_fwd_mfma_kernel<64", "motion_mlp_fwd"),
         ("mlp_fwd_f16_kernel<256", "image_mlp_fwd"), ("mlp_fwd_f16_kernel<64", "motion_mlp_fwd"),
         ("mlp_bwd_f16_kernel<256", "image_mlp_bwd"), ("mlp_bwd_f16_kernel<64", "motion_mlp_bwd"),
         ("mlp_fwd_bf16x2_kernel<256", "image_mlp_fwd"), ("mlp_fwd_bf16x2_kernel<64", "motion_mlp_fwd"),
         ("mlp_bwd_bf16x2_kernel<256", "image_mlp_bwd"), ("mlp_bwd_bf16x2_kernel<64", "motion_mlp_bwd"),
         ("motion_warp_fwd_kernel", "motion_warp_fwd"), ("motion_warp_bwd", "motion_warp_bwd"),
         ("mlp_bwd_mfma_kernel<256", "image_mlp_bwd"), ("mlp_bwd_mfma_kernel<64", "motion_mlp_bwd"),
         ("csr_bwd_kernel<3", "motion_encode_bwd"), ("csr_bwd_kernel<2", "image_encode_bwd"),
         ("select_dc_seed_kernel", "select_dc_seed"), ("ge_loss_kernel", "image_grad_init_ge")]
agg = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    assert files, f"no counter_collection.csv under {src}/{sub}"
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if "adam_" in name:             # image and motion share the kernel: tell them apart by grid size
                name += f" [grid {row['Grid_Size']}]"
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    agg[counter] = {k: (n, s / n) for k, (n, s) in acc.items()}
rows = [(c, k, n, v) for c, d in agg.items() for k, (n, v) in d.items()]
rows.sort(key=lambda r: -r[3])
out_csv = os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write_kb_per_launch.csv")
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "launches", "avg_value_KB (FETCH_SIZE under-reports wide streaming reads by 2x on gfx950)"])
    for c, k, n, v in rows:
        w.writerow([c, k[:110], n, round(v, 1)])
kern = {}
adam = sorted((k for k in agg["FETCH_SIZE"] if "adam_" in k),
              key=lambda k: -agg["FETCH_SIZE"][k][1])
names = dict(PHASE)
for k in set(agg["FETCH_SIZE"]) | set(agg["WRITE_SIZE"]):
    phase = next((p for pat, p in PHASE if pat in k), None)
    if phase is None and k in adam[:2]:
        phase = "adam_motion" if k == adam[0] else "adam_image"
    if phase is None:
        continue
    fk = agg["FETCH_SIZE"].get(k, (0, 0.0))[1]
    wk = agg["WRITE_SIZE"].get(k, (0, 0.0))[1]
    kern[phase] = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                   "hbm_bytes_corrected": int((2 * fk + wk) * 1024)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --iters 20 "
                     "--steps 1 --warmup 0 --no-cpu-baseline; per-launch averages; hbm_bytes_corrected = "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, "
                     "MI355X_MICROARCH.md; uncalibrated for 8-byte gathers); tools/pmc_to_traffic.py",
           "kernels": dict(sorted(kern.items()))}, open(os.path.join(ROOT, "profiles", traffic_name), "w"), indent=1)
print("wrote", out_csv, "and profiles/" + traffic_name + ":", {k: v["hbm_bytes_corrected"] for k, v in kern.items()})
