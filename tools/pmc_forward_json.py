#!/usr/bin/env python
"""rocprofv3 --pmc passes of tools/pmc_forward.sh (FETCH_SIZE / WRITE_SIZE over one bench.py forward run) -> profiles/<tag>_forward_pmc_<workload>.json:
HBM bytes per launch of EVERY kernel of the forward, stamped with the .so that was profiled (bench.py quotes `traffic` from it for that build only).
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte reads at 64 B, so it is doubled (MI355X_MICROARCH.md, "HBM");
WRITE_SIZE is taken as is (it equals the gate's exact output size E*H*4 B, which calibrates it).
usage: tools/pmc_forward_json.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out.json> <workload>"""
import collections
import csv
import hashlib
import json
import os
import sys


def main():
    src, out, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        with open(os.path.join(src, c, "p_counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("gnnome::", "")
                tot[name][k] += float(row["Counter_Value"])
                if k == 0:
                    tot[name][2] += 1
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnnome_amd", "lib", "libgnnome_hip.so")
    kernels = {name: {"launches": n, "hbm_read_bytes_per_launch": int(2 * f * 1024 / n), "hbm_write_bytes_per_launch": int(w * 1024 / n)}
               for name, (f, w, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]) if n and (f + w) / n > 1024}
    res = {"so_sha16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16], "workload": workload,
           "collected_with": "tools/pmc_forward.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per pass, over "
                             f"`bench.py --workload {workload} --steps 2 --warmup 1`",
           "corrections": "FETCH_SIZE doubled (gfx950 tallies 128-byte reads at 64 B - MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for name, v in list(kernels.items())[:8]:
        print(f"{name[:70]:70s} x{v['launches']:3d}  read {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
