#!/usr/bin/env python3
"""
The numbers table of DESIGN.md 7 from committed bench lines: python scripts/numbers_table.py [round tag, default r06]
(reads profiles/<tag>_final_bench_<workload>.json; prints markdown).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
rows = [("c1", "configs[0]"), ("c2", "configs[1]"), ("c3", "configs[2]"), ("c4share", "configs[3], one GPU's share"), ("batch", "configs[4], 8 tracks per GPU"),
        ("default", "reference default plan"), ("ov50", "C3 edges, overlap 0.5"), ("ov875", "C3 edges, overlap 0.875"),
        ("ov60", "C3 edges, overlap 0.6, 60 s"), ("wide65536", "chain_bands([0, 3000]), default max STFT")]
print("| workload | shape | ms per step | G samples/s | × real time | dominant kernel: frac of 8 TB/s (traffic ÷ algorithmic) | all launches, frac | CPU line (Msamples/s) |")
print("|---|---|---|---|---|---|---|---|")
for wl, what in rows:
    path = os.path.join(ROOT, "profiles", f"{tag}_final_bench_{wl}.json")
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    w = d["config"]["workload"]
    sizes = "[" + w.rsplit("STFT [", 1)[1].split("]")[0] + "]"
    per_gpu = w.rsplit(" per GPU", 1)[0]
    secs = per_gpu.rsplit(": ", 1)[1].split(" of ")[0] + (" of " + per_gpu.rsplit(": ", 1)[1].split(" of ")[1] if per_gpu.count(" of ") > 1 else "")
    secs = secs.replace(" of 48 kHz stereo", "").replace(" of 96 kHz stereo", " at 96 kHz")
    r = d["roofline"]
    kern = r["kernel"].replace("upx_band_kernel<upx::", "").replace("upx_zoom_synthesis_kernel<upx::", "syn ").replace("upx_zoom_analysis_kernel<upx::", "ana ")
    kern = kern.replace(", upx::", ", ")
    kern = kern[:-1] if kern.endswith(">") and not kern.startswith("unfused") else kern
    tr = f" ({r['traffic_ratio']}×)" if r.get("traffic_ratio") else ""
    cpu = d.get("cpu_baseline") or {}
    cpu_s = f"{cpu.get('value')} ({cpu.get('threads_used')} threads; serial {cpu.get('bands_serial_value')})" if cpu else "-"
    print(f"| `{wl}` {what} | {secs}, STFT {sizes} | **{d['ms_per_step']:.3f}** | {d['value'] / 1e3:.1f} | {d['config']['x_realtime']:,.0f} | "
          f"`{kern}` {r['frac']:.3f}{tr} | {d['all_bands_frac']:.3f} | {cpu_s} |")
