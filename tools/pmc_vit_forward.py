#!/usr/bin/env python
"""Summarise one rocprofv3 --pmc pass over tools/fwd_only.py (training-mode forward of the video tower) into the aggregate MFMA-busy
fraction bench.py prints as roofline.vit_forward_mfma_busy_frac:

    sum over the forward's kernels of SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs
    --------------------------------------------------------------------------
    sum over the same kernels of GRBM_GUI_ACTIVE / 8 XCDs   (= elapsed shader cycles; under --pmc kernels run one at a time)

    python tools/pmc_vit_forward.py <counter_collection.csv> <out.json> [tag]"""
import collections
import csv
import json
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id") or len(disp[k]))
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in per.values()) / 1024.0
    cyc = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in per.values()) / 8.0
    rows = sorted(((v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, k, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0, len(disp[k])) for k, v in per.items()), reverse=True)
    res = {"tag": tag, "mfma_busy_frac": round(busy / cyc, 4) if cyc else None,
           "note": "sum(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / sum(GRBM_GUI_ACTIVE / 8 XCDs) over every kernel of tools/fwd_only.py (training-mode "
                   "passes of the video tower, cfg #2); kernels are serialised under --pmc, so the two chains do not overlap here",
           "kernels": [{"kernel": k, "launches": n, "elapsed_Mcycles": round(c / 1e6, 2), "mfma_busy_frac": round(b / c, 4) if c else None} for c, k, b, n in rows[:14]]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("tag", "mfma_busy_frac")}))
    for r in res["kernels"]:
        print(f"  {r['kernel']:60s} launches {r['launches']:5d}  elapsed {r['elapsed_Mcycles']:9.2f} Mcyc  mfma busy {r['mfma_busy_frac']}")


if __name__ == "__main__":
    main()
