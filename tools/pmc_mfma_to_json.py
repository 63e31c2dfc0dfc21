#!/usr/bin/env python
"""tools/pmc_mfma.sh's passes -> profiles/<name>.json: per kernel, the matrix-core busy cycles against the cycles the chip was active.

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's SIMDs, the cycles a SIMD's matrix pipe was busy (32 per 32x32x16 16-bit MFMA -
MI355X_MICROARCH.md, price list; checked here: the H = 128 gate issues 1e6 / 32 x 96 MFMAs = 96 M busy cycles, the counter says 97.5 M);
GRBM_GUI_ACTIVE the cycles the launch kept the GPU active, SUMMED OVER THE 8 XCDs (3.5 M for a 0.218 ms launch = 8 x 440 k cycles at
2.0 GHz).  mfma_util = busy / (active / 8 x 1024 SIMDs): the fraction of the dense 16-bit matrix-core peak reached at the clock the
kernel ran at (x clock / 2.4 GHz for the fraction of the 2.5 PF headline peak).
usage: tools/pmc_mfma_to_json.py <dir of tools/pmc_mfma.sh> <out.json>"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gnnome::(?:\(anonymous namespace\)::)?[A-Za-z0-9_]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name.split("(")[0])[:110]


def main():
    src, out = sys.argv[1], sys.argv[2]
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnnome_amd", "lib", "libgnnome_hip.so")
    res = {"so_sha16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16],
           "collected_with": "tools/pmc_mfma.sh: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass per workload "
                             "(python bench.py --steps 3 --warmup 1)",
           "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), both averaged over the kernel's launches", "workloads": {}}
    for d in sorted(glob.glob(os.path.join(src, "*", ""))):
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        per = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("gnnome::"):
                    continue
                per.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        rows = []
        for k, c in per.items():
            busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
            if not busy or not act or sum(busy) == 0:
                continue
            mb, ma = sum(busy) / len(busy), sum(act) / len(act)
            rows.append({"kernel": k, "launches": len(busy), "mfma_busy_cycles": mb, "gui_active_cycles": ma,
                         "sq_busy_cycles": (sum(c["SQ_BUSY_CYCLES"]) / len(c["SQ_BUSY_CYCLES"])) if c.get("SQ_BUSY_CYCLES") else None,
                         "mfma_util": mb / (ma / 8 * SIMDS), "total_active_cycles": sum(act)})
        rows.sort(key=lambda r: -r["total_active_cycles"])
        res["workloads"][os.path.basename(os.path.dirname(d))] = rows
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for w, rows in res["workloads"].items():
        print(w)
        for r in rows[:12]:
            print(f"   {r['kernel'][:70]:70s} x{r['launches']:4d}  mfma_util {r['mfma_util']:.3f}  active {r['gui_active_cycles']:.0f} cyc")


if __name__ == "__main__":
    main()
