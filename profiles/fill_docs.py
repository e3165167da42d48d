#!/usr/bin/env python3
"""Fill the {H_*} placeholders of DESIGN.md / README.md from the round's committed headline profile files
(profiles/rNN_bench.json, rNN_kernel_stats_streams1.csv, rNN_pmc_summary.txt, rNN_pmc_cfg2_h_incr0.txt), so that the
prose cannot drift from the files it cites.  usage: python profiles/fill_docs.py r06 FILE..."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(path):
    vals = {}
    for line in open(path):
        parts = line.split()
        if len(parts) >= 4 and parts[1] == "mean/dispatch":
            vals[parts[0]] = float(parts[3])
    return vals


def main(tag, files, extra):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
    r = d["roofline"]
    k_ms = r["kernel_ms"]
    pm = pmc(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.txt"))
    prof_ms = prof_pct = None
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_streams1.csv"), newline="") as f:
        for row in csv.DictReader(f):
            if "k_sweep" in row["Name"]:
                prof_ms, prof_pct = float(row["AverageNs"]) * 1e-6, float(row["Percentage"])
                break
    traffic = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
    model = 67108864 * 3600 / (k_ms * 1e-3)
    rep = {
        "H_VALUE": f"{d['value'] / 1e6:.3f}", "H_MS": f"{d['ms_per_step']:.3f}",
        "H_ONE_VALUE": f"{d['one_sweep_in_flight']['value'] / 1e6:.3f}", "H_ONE_MS": f"{d['one_sweep_in_flight']['ms_per_step']:.3f}",
        "H_SUS_VALUE": f"{d['sustained']['value'] / 1e6:.3f}", "H_KMS": f"{k_ms:.2f}", "H_PROF_MS": f"{prof_ms:.3f}",
        "H_PROF_PCT": f"{prof_pct:.1f}", "H_TF": f"{r['achieved']:.1f}", "H_FRAC": f"{r['frac']:.2f}",
        "H_PS": f"{r['ps_per_point_lag']:.2f}", "H_VALU_N": f"{pm['SQ_INSTS_VALU']:.4g}",
        "H_VALU": f"{pm['SQ_INSTS_VALU'] / (k_ms * 1e-3) / 6.144e11:.2f}", "H_LDS_N": f"{pm['SQ_LDS_IDX_ACTIVE']:.4g}",
        "H_LDS": f"{pm['SQ_LDS_IDX_ACTIVE'] / (k_ms * 1e-3) / (256 * 2.4e9):.2f}",
        "H_CONF": f"{pm['SQ_LDS_BANK_CONFLICT'] / pm['SQ_LDS_IDX_ACTIVE']:.2f}", "H_TRAFFIC_MB": f"{traffic / 1e6:.0f}",
        "H_HBM_GBS": f"{traffic / (k_ms * 1e-3) / 1e9:.0f}", "H_HBM_FRAC": f"{100 * traffic / (k_ms * 1e-3) / 8e12:.2f}",
        "H_MODEL_TBS": f"{model / 1e12:.1f}", "H_MODEL_FRAC": f"{model / 8e12:.1f}",
        "H_AF_KMS": f"{d['all_finite_image']['kernel_ms']:.2f}", "H_AF_VALUE": f"{d['all_finite_image']['value'] / 1e6:.3f}",
        "H_PCIE_MS": f"{d['pcie_inclusive']['ms_per_step']:.2f}", "H_PCIE_VALUE": f"{d['pcie_inclusive']['value'] / 1e6:.3f}",
        "H_CPU": f"{d['cpu_baseline']['value']:.1f}",
    }
    off = os.path.join(ROOT, "profiles", f"{tag}_pmc_cfg2_h_incr0.txt")
    if os.path.exists(off):
        rep["C2_VALU_OFF"] = f"{pmc(off)['SQ_INSTS_VALU'] / (4194304 * 3721 / 64.0):.1f}"
    rep.update(extra)
    for path in files:
        s = open(path).read()
        for k, v in rep.items():
            s = s.replace("{" + k + "}", v)
        left = sorted({w for w in __import__("re").findall(r"\{([A-Z0-9_]+)\}", s)})
        open(path, "w").write(s)
        print(path, "filled;", "left:", left)


if __name__ == "__main__":
    extra = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a and not os.path.exists(a))
    main(sys.argv[1], [a for a in sys.argv[2:] if os.path.exists(a)], extra)
