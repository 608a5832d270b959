#!/usr/bin/env python3
"""Turns one gpurun_out/prof_<tag>/ directory (rocprofv3 csv output) into the small committed summaries:
   profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, as is)
   profiles/<tag>_pmc_summary.csv    (mean counter value per launch, kernels x counters)
   profiles/traffic.json             (HBM bytes per launch per kernel: FETCH_SIZE*2 + WRITE_SIZE, KiB -> bytes)
usage: summarize.py <prof dir> <tag> <windows_per_gpu>"""
import glob
import json
import shutil
import sys
from pathlib import Path

import pandas as pd

src, tag, win = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3])
dst = Path(__file__).resolve().parent
stats = src / "trace" / "trace_kernel_stats.csv"
if stats.exists():
    shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
other = src / "other_trace" / "trace_kernel_stats.csv"
if other.exists():
    shutil.copy(other, dst / f"{tag}_other_kernel_stats.csv")
other_traffic = {}
otr = {}
for f in sorted(glob.glob(str(src / "other_pmc[23]" / "pmc_counter_collection.csv"))):   # the uniform FETCH_SIZE / WRITE_SIZE passes
    g = pd.read_csv(f).groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean()
    for (k, cn), val in g.items():
        otr.setdefault(k.split("(")[0].replace("osh::", "").replace("void ", "").strip(), {})[cn] = float(val)
for k, cv in otr.items():
    if "FETCH_SIZE" in cv and "WRITE_SIZE" in cv:
        other_traffic[k] = (2.0 * cv["FETCH_SIZE"] + cv["WRITE_SIZE"]) * 1024.0
opm = sorted(glob.glob(str(src / "other_pmc1" / "pmc_counter_collection.csv")))
if opm:
    od = pd.concat([pd.read_csv(f).groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index() for f in opm])
    od["Kernel_Name"] = od["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("osh::", "")
    od.pivot_table(index="Counter_Name", columns="Kernel_Name", values="Counter_Value").to_csv(dst / f"{tag}_other_pmc_summary.csv", float_format="%.6g")
rows = []
for f in sorted(glob.glob(str(src / "pmc*" / "pmc_counter_collection.csv"))):
    df = pd.read_csv(f)
    rows.append(df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index())
if rows:
    d = pd.concat(rows)
    d["Kernel_Name"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("osh::", "")
    piv = d.pivot_table(index="Counter_Name", columns="Kernel_Name", values="Counter_Value")
    piv.to_csv(dst / f"{tag}_pmc_summary.csv", float_format="%.6g")
    # rocprofv3 kernel name -> short name of capi.KERNEL_NAMES
    names = {"void k_lin_lm<false>": "linearize", "void k_lin_lm<true>": "linearize",
             "void k_lin_lm<false, true>": "linearize", "void k_lin_lm<false, false>": "linearize", "void k_lin_lm<true, true>": "linearize",
             "void k_lin_lm<true, false>": "linearize", "void k_residual<false, true>": "residual", "void k_residual<false, false>": "residual",
             "void k_residual<true, true>": "residual", "void k_residual<true, false>": "residual", "void k_backsub<false, true>": "backsub",
             "void k_backsub<false, false>": "backsub", "void k_backsub<true, true>": "backsub", "void k_backsub<true, false>": "backsub", "void k_schur_fused<true, false>": "schur",
             "void k_schur_fused<false, false>": "schur_cross", "void k_schur_fused<true, true>": "schur", "void k_schur_fused<false, true>": "schur_cross",
             "void k_lin_items<0, false>": "linearize", "void k_lin_items<1, false>": "lin_pose", "void k_lin_aux<false>": "lin_aux",
             "void k_lin_items<0, true>": "linearize", "void k_lin_items<1, true>": "lin_pose", "void k_lin_aux<true>": "lin_aux",
             "void k_residual<false>": "residual", "void k_residual<true>": "residual", "k_pose_reduce": "pose_hess",
             "void k_backsub<false>": "backsub", "void k_backsub<true>": "backsub",
             "void k_schur_items<true>": "schur", "void k_schur_items<false>": "schur_cross", "k_schur_reduce": "schur_reduce",
             "void k_schur_reduce<false>": "schur_reduce", "void k_schur_reduce<true>": "schur_reduce",
             "void k_solve<24, 256>": "solve", "void k_solve<12, 256>": "solve", "void k_solve<6, 256>": "solve", "void k_solve<24, 512>": "solve",
             "void k_solve<12, 512>": "solve", "void k_solve<6, 512>": "solve", "k_backsub": "backsub", "k_residual": "residual",
             "dpack::k_pack_pre1": "pack_pre1", "dpack::k_pack_pre2": "pack_pre2", "dpack::k_pack_post": "pack_post", "k_widen_rec": "widen_rec"}
    traffic = {}
    for k, short in names.items():
        if k in piv.columns and "FETCH_SIZE" in piv.index and "WRITE_SIZE" in piv.index:
            # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> double it
            traffic[short] = float((2.0 * piv.loc["FETCH_SIZE", k] + piv.loc["WRITE_SIZE", k]) * 1024.0)
    (dst / "traffic.json").write_text(json.dumps({"tag": tag, "windows_per_gpu": win, "bytes_per_launch": traffic,
                                                  "other_bytes_per_launch": other_traffic,
                                                  "other_launch": {"k_orb_bruteforce": "64 frame pairs of 2000 x 2000", "k_liba<24>": "128 LocalInertialBA windows (config 4, ~2 000 landmarks)"},
                                                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; "
                                                            "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md HBM section)"},
                                                 indent=1))
    print(piv.to_string(float_format=lambda x: "%.4g" % x))
