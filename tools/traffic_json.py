#!/usr/bin/env python
"""profiles/<rnd>_traffic.json from this round's PMC runs (HBM / fabric bytes per launch of the dominant kernels):
    python tools/traffic_json.py <gpu_traffic.sh tag> <round prefix> [gpu_gather.sh tag] [gpu_pmc_any.sh tag of prof_other_models.py standard]
FETCH_SIZE is doubled and WRITE_SIZE taken as is (MI355X_MICROARCH.md, HBM section: gfx950 reports half the bytes of
16-byte-per-lane coalesced loads), each counter from its own rocprofv3 --pmc pass with the kernel trace only."""
import csv
import datetime
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
gather_tag = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
std_tag = sys.argv[4] if len(sys.argv) > 4 else None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")


def mean_counter(root, passdir, name, pred):
    cc = glob.glob(os.path.join(root, passdir, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(cc[0])) if r["Counter_Name"] == name and pred(r)]
    return sum(vals) / len(vals) if vals else None


qkv = lambda r: "gemm_f32_kernel<2, 2," in r["Kernel_Name"] and r["Grid_Size"] == "2359296"  # noqa: E731
fetch, write = mean_counter(src, "pmc_fetch", "FETCH_SIZE", qkv), mean_counter(src, "pmc_write", "WRITE_SIZE", qkv)
hit, req = mean_counter(src, "pmc_tcc", "TCC_HIT_sum", qkv), mean_counter(src, "pmc_tcc", "TCC_REQ_sum", qkv)
rows_, D = 65500, 768
alg = rows_ * D * 4 + 3 * D * D * 4 + rows_ * 3 * D * 4
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "xnrs_amd/csrc"], capture_output=True, text=True).stdout.strip())
out = {
    "commit": commit + ("+uncommitted csrc changes" if dirty else ""),
    "date": datetime.date.today().isoformat(),
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB)",
    "qkv_launch": {"rows": rows_, "grid_threads": 2359296, "what": "one full 65 500-row pass (the last pass of a call is shorter)"},
    "source": f"tools/gpu_traffic.sh {tag} (python3 tools/prof_news.py 1310 4 50 768 16), MI355X",
    "kernel": "gemm_f32_kernel<2,2,false,false,true,5,16,true,4,false,2> grid 2359296 (fused Q/K/V projection of one 65 500-row pass)",
    "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
    "correction": "gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of 16-B-per-lane coalesced loads (MI355X_MICROARCH.md, HBM): "
                  "reads doubled; WRITE_SIZE taken as is.",
    "qkv_gemm_hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
    "algorithmic_bytes_per_launch": alg,
    "l2_hit_rate": (hit / req) if hit and req else None,
    "note": "FETCH_SIZE counts L2-miss (fabric) requests and includes Infinity-Cache hits; A (201 MB) + W (7 MB) fit the 256 MB "
            "Infinity Cache, so the excess over the algorithmic 208 MB of reads is L2 re-fetch of operand tiles served on-die, not HBM traffic.",
    "gather_rows_hbm_bytes_per_launch": None,
    "gather_note": "no tools/gpu_gather.sh run was passed: no counter traffic for the gather kernel",
}
if gather_tag:
    gsrc = os.path.join(ROOT, "gpurun_out", f"prof_{gather_tag}")
    gk = lambda r: "gather_rows_kernel<true>" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000  # noqa: E731
    gf, gw = mean_counter(gsrc, "uniform_pmc_fetch", "FETCH_SIZE", gk), mean_counter(gsrc, "uniform_pmc_write", "WRITE_SIZE", gk)
    if gf is not None and gw is not None:
        out["gather_rows_hbm_bytes_per_launch"] = int((2 * gf + gw) * 1024)
        out["gather_note"] = (f"gather_rows_kernel<true>, 28 160 uniform ids x 153 600-B news rows out of a 10-GB table: FETCH_SIZE {gf:.5g} KB (x2), "
                              f"WRITE_SIZE {gw:.5g} KB (tools/gpu_gather.sh {gather_tag}, this binary)")
if std_tag:
    ssrc = os.path.join(ROOT, "gpurun_out", f"prof_{std_tag}")
    # the attention-free encoders' dominant kernel: the one-launch additive encoder (or the RDOT GEMM) over 256 k-row passes
    cc = glob.glob(os.path.join(ssrc, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
    best = None
    if cc:
        by = {}
        for r in csv.DictReader(open(cc[0])):
            if r["Counter_Name"] == "FETCH_SIZE" and ("additive_fused" in r["Kernel_Name"] or "gemm_f32_kernel" in r["Kernel_Name"]):
                by.setdefault((r["Kernel_Name"][:90], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
        if by:
            best = max(by.items(), key=lambda kv: sum(kv[1]) / len(kv[1]))
    if best:
        (kname, grid), vals = best
        big = [v for v in vals if v >= 0.5 * max(vals)]  # the history launch (512 x 25 news); the candidates' launch is 5x smaller
        sf = sum(big) / len(big)
        wc = glob.glob(os.path.join(ssrc, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
        wv = [float(r["Counter_Value"]) for r in csv.DictReader(open(wc[0])) if r["Counter_Name"] == "WRITE_SIZE"
              and r["Kernel_Name"][:90] == kname and r["Grid_Size"] == grid] if wc else []
        wbig = [v for v in wv if v >= 0.5 * max(wv)] if wv else [0.0]
        sw = sum(wbig) / len(wbig)
        rows_h = 512 * 25 * 50
        out["standard_fc1"] = {"kernel": kname, "grid_threads": int(grid), "FETCH_SIZE_KB_raw": sf, "WRITE_SIZE_KB": sw,
                               "hbm_bytes_per_launch": int((2 * sf + sw) * 1024),
                               "algorithmic_bytes_per_launch": rows_h * 768 * 4 + 256 * 768 * 4 + 512 * 25 * 768 * 4,
                               "launch": "history tower of StandardRec, B = 512, H = 25, S = 50, D = 768: 640 000 token rows in ONE persistent launch",
                               "note": "2.2 x one read of x: the kernel reads every 256-row tile twice -- once through the fc1 K loop, once for the "
                                       "weighted sum of the rows when their scores are known -- and a tile (786 KB) fits neither LDS (160 KB) nor, "
                                       "with 256 workgroups in flight, L2 (4 MB per XCD); the two-kernel form (GEMM + pooling) reads x twice as well",
                               "source": f"tools/gpu_pmc_any.sh {std_tag} prof_other_models.py standard 3"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
