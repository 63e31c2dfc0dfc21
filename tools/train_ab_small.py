#!/usr/bin/env python
"""A/B of the training step with and without the one-launch weight packing / BatchNorm-backward terms (same box, same process order)."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, json, io, contextlib
sys.path.insert(0, %r)
import gnnome_amd.ops as o
if %d:
    del o.pack_layer, o.bn_bwd_terms
import bench
sys.argv = ['bench.py', '--mode', 'train', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-extras', '--no-kernel-timers']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
print(('torch operators' if %d else 'one launch each'), round(json.loads(buf.getvalue().strip().splitlines()[-1])['ms_per_step'], 4))
"""
for rnd in range(2):
    for off in (0, 1):
        subprocess.run([sys.executable, "-c", code % (root, off, off)], check=False, stderr=subprocess.DEVNULL)
