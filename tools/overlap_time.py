#!/usr/bin/env python
"""Time gnnome_overlap_edit_distance on overlaps of assembly shape (GPU box).

    python tools/overlap_time.py [--reads 4000] [--rate 0.003] [--full-too]

Reads of 12-18 kb laid out every 2.5 kb on a random genome, each with `rate` substitutions / indels (HiFi: ~0.1-0.5 %);
every read against its next five, both strands.  Prints overlaps/s for the shipped path (Ukkonen band first) and, with
--full-too, for the full-matrix kernels alone (gnnome_set_tuning(9, 1)); the two must give the same distances."""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops, overlap  # noqa: E402


def mutate(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
            continue
        if r < rate:
            out.append(rng.choice("ACGT"))
        out.append(ch)
    return "".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--rate", type=float, default=0.003)
    ap.add_argument("--full-too", action="store_true")
    args = ap.parse_args()
    rng = random.Random(0)
    R = args.reads
    genome = "".join(rng.choice("ACGT") for _ in range(R * 2500 + 20000))
    reads = [mutate(rng, genome[r * 2500: r * 2500 + rng.randrange(12000, 18000)], args.rate) for r in range(R)]
    src, dst, ol = [], [], []
    for r in range(R - 6):
        for t in range(1, 6):
            o = len(reads[r]) - 2500 * t
            if 500 < o <= len(reads[r + t]):   # (a read contained in another one is no overlap edge: assemblers drop contained reads)
                src.append(2 * r), dst.append(2 * (r + t)), ol.append(o)              # suffix of r against prefix of r + t
                src.append(2 * (r + t) + 1), dst.append(2 * r + 1), ol.append(o)      # its mate on the other strand
    packed = overlap.pack_reads(reads)
    dev = torch.device("cuda", 0)

    def run(tag):
        st = {}
        overlap.edit_distances(packed, src, dst, ol, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, _ = overlap.edit_distances(packed, src, dst, ol, device=dev, stats=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cells = sum(o * o for o in ol)
        print(f"{tag}: {len(ol)} overlaps, mean length {sum(ol) / len(ol):.0f}, rate {args.rate}, {dt * 1e3:.1f} ms, {len(ol) / dt:.0f} overlaps/s, "
              f"{cells / dt / 1e12:.2f} T full-matrix-equivalent cells/s, settled by the band {st['banded']} / {st['edges']}, "
              f"median dist {int(d.median())}, max dist {int(d.max())}", flush=True)
        return d
    d0 = run("band + full")
    if args.full_too:
        ops.set_tuning(9, 1)
        d1 = run("full matrix only")
        ops.set_tuning(9, 0)
        print("same distances:", bool(torch.equal(d0, d1)))


if __name__ == "__main__":
    main()
