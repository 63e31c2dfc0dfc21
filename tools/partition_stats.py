#!/usr/bin/env python
"""World-2/4/8 destination-range partition statistics of the benchmark graphs, on the CPU (no GPU, no process group).

    python tools/partition_stats.py [--workloads 10m,c4,c5] [--kinds banded,uniform] [--worlds 2,4,8] > profiles/r04_partition_stats.md

Per (graph, kind, world): cut fraction, edge replication (local edges summed over ranks / E), imbalance (largest rank's
local edges x world / sum), and for the largest rank the owned nodes, local edges, halo rows received and rows sent per
layer, bytes per exchange at the workload's hidden width, and the time of one exchange at the xGMI per-link rate when
every peer pair uses its own link (direct all_to_all: the largest single-link transfer bounds it).
gnnome_amd.dist.partition_census is checked against the plans real ranks build in tests/test_dist_gloo.py."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402
from gnnome_amd.dist import partition_census  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

XGMI_LINK = 153e9   # B/s per direction and link, MI355X_MICROARCH.md (7 links per GPU, one per peer)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="10m,c4,c5")
    ap.add_argument("--kinds", default="banded,uniform")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--json", default=None, help="also write the raw census records here")
    args = ap.parse_args()
    records = []
    print("# destination-range partition statistics (tools/partition_stats.py, gnnome_amd.dist.partition_census; seed 1)\n")
    print("| graph | kind | world | cut edges | replication | imbalance | largest rank: owned nodes | local edges | halo rows | rows sent / layer | "
          "largest link rows | exchange bytes sent (fp32) | ms / exchange at 153 GB/s per link |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name in args.workloads.split(","):
        n, e, h = WORKLOADS[name]
        for kind in args.kinds.split(","):
            t0 = time.time()
            g = make_graph(n, e, seed=1, kind=kind)
            print(f"[{name} {kind}: graph in {time.time() - t0:.0f} s]", file=sys.stderr, flush=True)
            for world in (int(w) for w in args.worlds.split(",")):
                c = partition_census(g["src"], g["dst"], n, world)
                big = max(c["ranks"], key=lambda r: r["local_edges"])
                link_rows = max(r["largest_link_rows"] for r in c["ranks"])
                sent = max(r["rows_sent_per_layer"] for r in c["ranks"])
                c.update(workload=name, kind=kind, hidden=h, exchange_ms_link_bound=link_rows * h * 4 / XGMI_LINK * 1e3)
                records.append(c)
                print(f"| {name} N={n} E={e} H={h} | {kind} | {world} | {100 * c['cut_fraction']:.2f} % | {c['edge_replication']:.3f} | "
                      f"{c['imbalance_local_edges']:.3f} | {big['owned_nodes']} | {big['local_edges']} | {big['halo_rows']} | {sent} | {link_rows} | "
                      f"{sent * h * 4 / 1e6:.1f} MB | {c['exchange_ms_link_bound']:.3f} |", flush=True)
            del g
    if args.json:
        with open(args.json, "w") as f:
            json.dump(records, f)


if __name__ == "__main__":
    main()
