#!/usr/bin/env python
"""Merge the three PMC summaries of tools/pmc_gemm256.sh into one small JSON stamped with the hash of the GEMM sources the
library was built from; bench.py prints `roofline.traffic` from the newest profiles/r*_pmc_gemm256.json and refuses it when
the stamp does not match the sources in the tree (stale evidence is not printed as current).
Usage: pmc_gemm256_json.py <tag> <dir with <tag>_pmc_{sq,fetch,write}_gemm256.csv> <out.json>"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["xpretrain_amd/csrc/gemm256.hip", "xpretrain_amd/csrc/gemm_common.h", "xpretrain_amd/csrc/common.h",
           "xpretrain_amd/csrc/gemm.hip"]


def source_stamp(root=ROOT):
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


# grid sizes (threads) of the probe's launches: tiles * 512 threads
VARIANT = {"gemm256_kernel<false, false": "NT_fc1_fwd", "gemm256_kernel<false, true": "NS_dpre_dx", "gemm256_kernel<true, true": "SS_dw1"}

if __name__ == "__main__":
    tag, d, out = sys.argv[1:4]
    res = {"tag": tag, "source_stamp": source_stamp(), "probe": "tools/gemm256_probe.py (cfg #2 shapes, 4 launches each)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; SQ_* raw", "kernels": {}}
    for part in ("sq", "fetch", "write", "tcc"):
        f = os.path.join(d, f"{tag}_pmc_{part}_gemm256.csv")
        if not os.path.isfile(f):
            continue
        for r in csv.DictReader(open(f)):
            name = next((v for k, v in VARIANT.items() if k in r["kernel"]), "xp_cast_calibration" if "cast_kernel" in r["kernel"] else None)
            if name:
                res["kernels"].setdefault(name, {})[r["counter"]] = float(r["mean_per_launch"])
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res)[:600])
