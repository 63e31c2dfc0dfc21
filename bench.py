#!/usr/bin/env python
"""bench.py - edges/sec of one full SymGatedGCNModel forward (encoders + 8 layers + scorer).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|10m|parity64|c4shard|c4|c5|c5shard|c5quarter|c4quarter] [--kind banded|uniform]
                    [--mode infer|train]

One "step" = one `model(graph, x, e)` on a synthetic assembly graph already resident in HBM (graph views prebuilt, as a
caller that scores the same graph repeatedly would have them; the cold numbers - view build, first call - are reported
separately).  N = 1 runs BASELINE.json configs[1] (c2: N=1e5, E=1e6, H=128) by default and adds two sub-records to the
line: `target_10m` (the north_star's 10M-edge graph, GPU forward) and `train` (configs[2]'s shape: fwd + BCE + bwd + Adam,
fp32, replayed from a hipGraph).  N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) runs ONE graph
partitioned by destination-node range (gnnome_amd/dist.py): strong scaling; the default graph there is the 10M-edge one
(at 1M edges a rank has < 1 ms of kernels per forward and the launch / collective host work is what gets timed), and
rank 0 first times the same graph on its GPU alone (`scaling_reference`).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nodes, edges, hidden) - BASELINE.json configs[1] is the default
    "c2": (100_000, 1_000_000, 128),
    "10m": (1_000_000, 10_000_000, 128),       # north_star's 10x target graph
    "parity64": (100_000, 1_000_000, 64),
    "c4shard": (250_000, 2_500_000, 256),      # one GPU's eighth of configs[3]
    "c4": (2_000_000, 20_000_000, 256),        # BASELINE.json configs[3] (8 GPUs; 20.5 GB of edge state: fits one GPU as well)
    "c4quarter": (500_000, 5_000_000, 256),    # a quarter of configs[3]: `--gpus 8 --one-gpu-gloo` plumbing runs at H = 256
    "c5": (5_000_000, 50_000_000, 256),        # BASELINE.json configs[4] (8-GPU training step)
    "c5shard": (625_000, 6_250_000, 256),      # one GPU's eighth of configs[4]: what a rank of the 8-GPU training step holds
    "c5quarter": (1_250_000, 12_500_000, 256), # a quarter of configs[4] (two ranks' shares: `--gpus 2 --one-gpu-gloo --mode train`)
    # SURVEY.md 8d's substitute for configs[0]: an E. coli-sized graph with the SHIPPED checkpoint (tests/golden/weights.pt,
    # H = 64) in the default "auto" arithmetic - what a user of inference.py runs: layer 0 (bn_e gain 135) goes through the
    # reference-order fp32 VALU kernels, layers 1-7 through the bf16x6 matrix-core kernels
    "ecoli": (30_000, 300_000, 64),
}
HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12  # flop/s, v_mfma_f32_32x32x2_f32
MFMA_BF16_PEAK = 2.5e15   # flop/s dense bf16 (no sparsity), v_mfma_f32_32x32x16_bf16


def algorithmic_bytes(n, e, h, layers=8):
    """B_fwd of BASELINE.md section 2.2 (s = 4 bytes)."""
    b_enc = (2 * n + 2 * e) * 4 + (n + e) * h * 4
    b_layer = (2 * e * h + 2 * n * h) * 4 + 2 * e * 4
    b_pred = (e * h + n * h) * 4 + 2 * e * 4 + e * 4
    return b_enc + layers * b_layer + b_pred


def algorithmic_flops(n, e, h, layers=8, h_ne=16, hs=64):
    f_enc = 2 * (n + e) * (2 * h_ne + h_ne * h)
    f_layer = 10 * n * h * h + 2 * e * h * h
    f_pred = 2 * e * (3 * h * hs + hs * 32 + 32)
    return f_enc + layers * f_layer + f_pred


def workload_state_dict(workload, hidden):
    if workload == "ecoli":
        return torch.load(os.path.join(ROOT, "tests", "golden", "weights.pt"), map_location="cpu")
    from gnnome_amd.synth import random_state_dict
    return random_state_dict(hidden, seed=1)


def so_sha16():
    """Identity of the shipped kernels: PMC files under profiles/ are keyed by it (a stale one is never quoted)."""
    from gnnome_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


class KernelTimer:
    """HIP-event pairs around launches of chosen gnnome_amd.ops entry points, on the launch stream (the library launches
    on torch's current stream, which is the stream torch.cuda.Event records on)."""

    def __init__(self, ops_mod, names, every=1):
        """every=k: instrument only every k-th launch of each entry point (k = 8 -> one layer's launch per step)."""
        self.ops, self.names, self.events, self.orig, self.on = ops_mod, names, {n: [] for n in names}, {}, False
        self.every, self.calls = every, {n: 0 for n in names}

    def __enter__(self):
        for name in self.names:
            fn = getattr(self.ops, name)
            self.orig[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **k):
                if not self.on:
                    return _fn(*a, **k)
                self.calls[_name] += 1
                if self.calls[_name] % self.every:
                    return _fn(*a, **k)
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = _fn(*a, **k)
                t.record()
                self.events[_name].append((s, t))
                return out

            setattr(self.ops, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.ops, name, fn)

    def mean_ms(self, name):
        ev = self.events[name]
        return sum(s.elapsed_time(t) for s, t in ev) / max(len(ev), 1), len(ev)


class ForwardEventTimer:
    """KernelTimer's interface for a forward that is ONE library call (gnnome_model_forward_f32, the default since round 6): the HIP events go
    around the last layer's edge-gate and aggregation launches INSIDE that call (gnnome_debug_forward_events), on the stream it launches on -
    one pair per kernel and step, as before.  The events are created (and recorded once, which is what makes torch create them) before the
    timed region."""

    def __init__(self, lib, steps, layer, launches_per_step):
        self.lib, self.layer, self.on, self.step = lib, layer, False, 0
        self.pool = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
        for quad in self.pool:
            for ev in quad:
                ev.record()
        torch.cuda.synchronize()
        self.events = {"edge_gate": [], "node_aggregate": []}
        self.calls = {"edge_gate": 0, "node_aggregate": 0}
        self.per_step = launches_per_step

    def before_step(self):
        if not self.on or self.step >= len(self.pool):
            self.lib.gnnome_debug_forward_events(None, None, None, None, -1)
            return
        q = self.pool[self.step]
        self.lib.gnnome_debug_forward_events(*(ev.cuda_event for ev in q), self.layer)
        self.events["edge_gate"].append((q[0], q[1]))
        self.events["node_aggregate"].append((q[2], q[3]))
        for k, v in self.per_step.items():
            self.calls[k] += v
        self.step += 1

    def close(self):
        self.lib.gnnome_debug_forward_events(None, None, None, None, -1)

    def mean_ms(self, name):
        ev = self.events[name]
        return sum(s.elapsed_time(t) for s, t in ev) / max(len(ev), 1), len(ev)


def _placement_record(ops):
    """What gnnome_amd.ops._placed_buffers did for this run's forward: the candidates' forward times, per buffer group ("Placement")."""
    placed = [st for st in getattr(ops, "_PLACED", {}).values() if st.bufs is not None and st.log]
    if not placed:
        return {"tuned": False, "note": "buffers as the allocator handed them out (GNNOME_TUNE_PLACEMENT=0, a workspace block, or shapes whose placement does not vary)"}
    log = placed[-1].log
    return {"tuned": True, "start_ms": round(log[0][1], 4), "kept_ms": round(min(t for _, t in log), 4),
            "candidates_ms": [("+".join(g) if not isinstance(g, str) else g, round(t, 4)) for g, t in log[1:]],
            "note": "forward time with each candidate allocation of a buffer group, all else fixed; the fastest is kept for (device, stream, shapes) - same kernels, same bits"}


def _mfma_util(workload_key):
    """Matrix-core utilisation per kernel from the committed PMC pass over this workload (tools/pmc_mfma.sh -> profiles/<tag>_mfma_busy.json:
    SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE x SIMDs, per the guide's PMC section; north_star: "MFMA utilisation ... against CDNA4 peak").
    Quoted only when the file was collected on THIS build of the library."""
    prof = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof), reverse=True):
        if name.endswith("_mfma_busy.json"):
            with open(os.path.join(prof, name)) as f:
                d = json.load(f)
            if d.get("so_sha16") != so_sha16() or workload_key not in d.get("workloads", {}):
                continue
            return {"source": "profiles/" + name, "definition": d.get("mfma_util"),
                    "kernels": {r["kernel"]: round(r["mfma_util"], 4) for r in d["workloads"][workload_key]}}
    return None


def _pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` (a substring of its name) from the committed PMC passes over one forward of `workload`
    (FETCH_SIZE / WRITE_SIZE cannot be read from inside this process; tools/pmc_forward.sh collects them per the guide - separate
    --pmc passes, FETCH_SIZE doubled on gfx950 - and tools/pmc_forward_json.py stamps the file with the .so it profiled).  Only quoted
    for that exact build and workload: a stale file is never used."""
    prof = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof), reverse=True):
        if name.endswith(f"_forward_pmc_{workload}.json"):
            with open(os.path.join(prof, name)) as f:
                d = json.load(f)
            if d.get("so_sha16") != so_sha16():
                continue
            hits = [v for k, v in d["kernels"].items() if kernel in k]
            if hits:
                v = max(hits, key=lambda r: r["launches"])
                return v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
    return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, kind, mode="infer", full=False):
    """Child-process leg: the CPU path timed on this host's cores.  `try: import dgl` - if DGL 0.8.1 were importable the
    reference's own classes would be timed (kind "reference"); it is not installable offline, so the oracle
    (oracle/symgated_oracle.py: torch-CPU restatement of the reference path, pinned to it by the goldens) is (kind "port").
    Protocol: the thread setting is chosen on a small graph of the same generator and width (E = 200k, one
    warm-up + one run at 8, 32 and all threads), then the workload at E = 1M (configs[1]'s own graph): 1 warm-up + median
    of 3 at that setting - ~70-90 s on the GPU pool's hosts.  full=True adds SURVEY 8d's single E = 10M run (minutes).
    Training steps (--mode train) are timed on E = 20k."""
    try:
        import dgl  # noqa: F401
        have_dgl = True
    except Exception:  # noqa: BLE001
        have_dgl = False
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import bce_loss, degree_features, model_from_state_dict
    cores = os.cpu_count() or 1
    hidden = WORKLOADS[workload][2]
    model = model_from_state_dict(workload_state_dict(workload, hidden))

    def prepare(n, e):
        g = make_graph(n, e, seed=1, kind=kind)
        x = degree_features(g["src"], g["dst"], n)
        graph = (g["src"], g["dst"], n)
        if mode == "train":
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)

            def run():
                loss = bce_loss(model(graph, x, g["e"]), g["y"], g["pos_weight"])
                opt.zero_grad()
                loss.backward()
                opt.step()
        else:
            model.eval()

            def run():
                with torch.no_grad():
                    model(graph, x, g["e"])
        return run

    def timed(run, reps, warm=True):
        if warm:
            run()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    def record(seconds, threads, n, e, protocol):
        rec = {
            "value": e / seconds, "unit": "edges/s", "cores": threads, "kind": "port" if not have_dgl else "port (dgl importable but not used)",
            "sample": f"{mode}: {kind} synthetic graph N={n} E={e} H={hidden} L=8 fp32 through oracle/symgated_oracle.py (torch-CPU "
                      f"restatement of the reference path; `import dgl` {'succeeded' if have_dgl else 'failed: DGL 0.8.1 is not installable offline'}); "
                      f"{protocol}",
            "sample_edges": e, "host": f"{_cpu_model()}, {cores} logical cores", "seconds_per_forward": seconds,
        }
        print(json.dumps(rec), flush=True)   # one line per stage: the parent keeps the last one if a later stage is cut off
        return rec

    # stage 1: thread setting on the small graph
    n_s, e_s = (20_000, 200_000) if mode == "infer" else (2_000, 20_000)
    run = prepare(n_s, e_s)
    best, tried = None, []
    for threads in sorted({min(8, cores), min(32, cores), cores}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        run()
        first = time.perf_counter() - t0
        tried.append(threads)
        if best is not None and first > 2.5 * best[0]:
            break  # oversubscribed pool: larger settings only get slower
        med = timed(run, 1 if mode == "infer" else 3, warm=False)
        if best is None or med < best[0]:
            best = (med, threads)
        rec = record(best[0], best[1], n_s, e_s, f"thread-setting probe: best of {'/'.join(map(str, tried))} threads")
    if mode != "infer":
        return
    # stage 2: SURVEY 8d's 1M-edge protocol at the chosen setting (the ecoli workload: its own 300k-edge graph)
    torch.set_num_threads(best[1])
    n1, e1 = WORKLOADS[workload][:2] if workload == "ecoli" else (100_000, 1_000_000)
    run1 = prepare(n1, e1)
    rec = record(timed(run1, 3), best[1], n1, e1, f"1 warm-up + median of 3 at {best[1]} threads (fastest of {'/'.join(map(str, tried))} on a 200k-edge probe)")
    if full:
        run10 = prepare(1_000_000, 10_000_000)
        t10 = timed(run10, 1)
        rec["e10m"] = {"value": 1e7 / t10, "unit": "edges/s", "cores": best[1], "seconds_per_forward": t10, "sample_edges": 10_000_000,
                       "sample": "N=1e6 E=1e7: 1 warm-up + 1 timed run (SURVEY.md 8d)"}
        print(json.dumps(rec), flush=True)


def _run_cpu_child(args, timeout):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload, "--kind", args.kind,
           "--mode", args.mode] + (["--cpu-baseline-full"] if args.cpu_baseline_full else [])
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        child.kill()
        out, err = child.communicate()
        err = f"stopped after {timeout} s; " + err[-200:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return (json.loads(lines[-1]) if lines else None), err


def _spawn_ranks(gpus):
    """Re-run this command line under torch.distributed.run with one rank per GPU; returns the launcher's exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    return subprocess.call(cmd, env=env)


def _stdout_to_stderr(fn):
    """Run fn with file descriptor 1 pointing at stderr: gloo's C++ side prints its connection banner to stdout, and this
    program's stdout carries exactly ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _time_steps(step, steps, warmup, barrier):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    t_host = time.perf_counter() - t0
    barrier()
    return out, time.perf_counter() - t0, t_host


def _single_gpu_forward(gnnome_amd, ops, make_graph, random_state_dict, workload, kind, dev, steps, warmup):
    """ms per forward of `workload` on ONE GPU (used for the target_10m sub-record and the N>1 scaling reference)."""
    n, e, hidden = WORKLOADS[workload]
    g = make_graph(n, e, seed=1, kind=kind)
    model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
    model.load_state_dict(random_state_dict(hidden, seed=1))
    model.to(dev)
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    x, ef = ops.degree_features(views), g["e"].to(dev)
    _, elapsed, _ = _time_steps(lambda: model(views, x, ef), steps, warmup, torch.cuda.synchronize)
    del model, views, x, ef
    torch.cuda.empty_cache()
    ms = elapsed / steps * 1e3
    return {"workload": f"{workload}: {kind} N={n} E={e} H={hidden}", "n_gpus": 1, "ms_per_step": ms, "value": e / (ms * 1e-3), "unit": "edges/s",
            "steps": steps, "hbm_roofline_frac_whole_fwd": algorithmic_bytes(n, e, hidden) / (ms * 1e-3) / HBM_PEAK}


def _memory_record(dev, edges_local, hidden):
    """Peak device memory of this process since the last reset: what the kernels' tensors took (allocated) and what torch's
    caching allocator held from the driver for them (reserved >= allocated: rounding + fragmentation), and both per local edge."""
    alloc, reserved = torch.cuda.max_memory_allocated(dev), torch.cuda.max_memory_reserved(dev)
    free, total = torch.cuda.mem_get_info(dev)
    return {"peak_memory_GB": alloc / 1e9, "peak_reserved_GB": reserved / 1e9, "device_total_GB": total / 1e9,
            "bytes_per_local_edge_allocated": alloc / max(edges_local, 1), "bytes_per_local_edge_reserved": reserved / max(edges_local, 1),
            "local_edges": edges_local, "hidden": hidden}


def _train_record(gnnome_amd, ops, g, n, e, hidden, dev, steps, warmup, symmetry, dropout, storage="fp32", recompute=False):
    """configs[2]'s shape: fwd + loss + bwd + Adam on the whole graph, fp32, the step replayed from a hipGraph (a training
    loop over one graph repeats the same launch sequence; ~450 library launches + a few hundred small torch ops cost
    30-40 ms of host time per step when issued eagerly)."""
    from gnnome_amd.loss import bce_loss, symmetry_loss
    model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch", dropout=dropout).train()
    from gnnome_amd.synth import random_state_dict
    model.load_state_dict(random_state_dict(hidden, seed=1))
    model.to(dev)
    model.activation_storage = storage   # "bf16": xe / dxe stored as bfloat16 between the kernels (arithmetic stays fp32)
    model.recompute_gate = recompute     # True: xe is not kept for the backward, the raw gate runs again (gnnome_amd/train.py)
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    x, ef, y, pw = ops.degree_features(views), g["e"].to(dev), g["y"].to(dev), g["pos_weight"].to(dev)
    # torch.optim.Adam as train.py:259 builds it, in its fused single-kernel form (fused / capturable are implementation
    # switches of the same update rule; the default foreach form issues ~80 multi-tensor launches per step)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True, fused=True)
    rev, x_rev = views.reversed(), ops.degree_features(views, reverse=True)

    def eager_step():
        logits = model(views, x, ef)
        if symmetry:   # train.py:159-170: second pass over dgl.reverse(g) with the degree columns swapped
            loss = symmetry_loss(logits.squeeze(-1), model(rev, x_rev, ef).squeeze(-1), y, pw, alpha=0.1)
        else:          # train.py:138-145
            loss = bce_loss(logits.squeeze(-1), y, pw)
        opt.zero_grad()          # (set_to_none: the backward ASSIGNS the 142 gradients instead of zero-filling and adding)
        loss.backward()
        opt.step()
        return loss.detach()

    # every eager step on ONE side stream (torch's allocator keeps a pool per stream: a step on a second stream would hold a
    # second set of activations - at the configs[4] shard that is 2 x 170 GB), then the cached blocks go back to the driver
    # before the capture allocates the step's tensors once more in the graph's private pool
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        eager_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) * 1e3
    memory_eager = _memory_record(dev, e, hidden)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = eager_step()

    def step():
        graph.replay()
        return static_loss

    loss, elapsed, _ = _time_steps(step, steps, warmup, torch.cuda.synchronize)
    assert torch.isfinite(loss).all()
    ms = elapsed / steps * 1e3
    passes = 2 if symmetry else 1
    b_fwd, f_fwd = algorithmic_bytes(n, e, hidden), algorithmic_flops(n, e, hidden)
    return {"metric": "edges/sec full-graph training step", "value": e / (ms * 1e-3), "unit": "edges/s", "ms_per_step": ms, "steps": steps,
            "step": ("train.py:159-170 symmetry loss: two train-mode forwards (graph and reversed graph) + BCE both ways + |org - rev|" if symmetry
                     else "train.py:138-145 + :328-330: train-mode forward (batch-statistic BatchNorm) + BCEWithLogits(pos_weight)")
                    + f" + backward + Adam, fp32 arithmetic, activation storage {storage}, dropout {dropout or 0}, whole step replayed from one hipGraph",
            "eager_ms_per_step": eager_ms, "dtype": "f32", "activation_storage": storage, "recompute_gate": recompute, "loss": float(loss),
            "memory": {"eager_step": memory_eager, "hipgraph_step": _memory_record(dev, e, hidden)},
            "hbm_roofline_frac_3xBfwd": passes * 3 * b_fwd / (ms * 1e-3) / HBM_PEAK, "mfma_f32_frac_3xFfwd": passes * 3 * f_fwd / (ms * 1e-3) / MFMA_F32_PEAK}


def _partitioned_train_record(gnnome_amd, gdist, ops, g, n, e, hidden, dev, rank, world, plan, runner, model, args, gloo_transport):
    """BASELINE configs[4]'s step on a destination-range partition (SURVEY.md 8e "Training additions"; train.py:138-145,
    328-330): every rank runs the train-mode forward over its rows (halo exchange per layer, BatchNorm statistics merged over
    ranks), computes BCEWithLogits(pos_weight) on the assembled logits, runs the backward (halo gradients returned to their
    owners per layer, BatchNorm-backward sums all-reduced) and ends with ONE flat all-reduce of the 142 parameter gradients;
    identical Adam instances then stay in step without a wrapper.  Eager (collectives between the kernels: no hipGraph)."""
    import torch.distributed as dist
    from gnnome_amd.loss import bce_loss
    y, pw = g["y"].to(dev), g["pos_weight"].to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    def step():
        loss = bce_loss(runner.train_forward().squeeze(-1), y, pw)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.detach()

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    loss, elapsed, t_host = _time_steps(step, args.steps, args.warmup, barrier)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo_transport else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert torch.isfinite(loss).all()
    # the parameters every rank ends up with must be the same bits (same summed gradients, same optimizer state)
    digest = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
    every = gdist.all_gather_rows(digest.to(dev), world).view(-1)
    in_step = bool((every == every[0]).all())
    ms = elapsed / args.steps * 1e3
    b_fwd, f_fwd = algorithmic_bytes(n, e, hidden), algorithmic_flops(n, e, hidden)
    n_params = sum(p.numel() for p in model.parameters())
    transport = ("ALL RANKS ON ONE GPU over host-staged gloo: plumbing check, not a measurement" if args.one_gpu_gloo
                 else "RCCL FAILED: host-staged gloo transport" if gloo_transport else "RCCL")
    return {
        "metric": "edges/sec full-graph training step", "value": e / (ms * 1e-3), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {args.kind} synthetic assembly graph N={n} E={e}, SymGatedGCNModel hidden={hidden} L=8 hs=64, "
                               f"train.py:138-145 + :328-330: train-mode forward (batch-statistic BatchNorm over the whole graph) + "
                               f"BCEWithLogits(pos_weight) + backward + Adam, fp32, random-init weights seed 1",
                   "parallelism": f"dst-range x{world}, halo all_to_all per layer both ways, BatchNorm statistics merged over ranks, "
                                  f"one flat gradient all-reduce ({n_params * 4} B) per step; {transport}"},
        "loss": float(loss), "ranks_in_step": in_step, "recompute_gate": bool(getattr(model, "recompute_gate", False)),
        "activation_storage": getattr(model, "activation_storage", "fp32"), "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
        "hbm_roofline_frac_3xBfwd": 3 * b_fwd / (ms * 1e-3) / (world * HBM_PEAK), "mfma_f32_frac_3xFfwd": 3 * f_fwd / (ms * 1e-3) / (world * MFMA_F32_PEAK),
        "rank0": {"owned_nodes": plan.n_own, "halo_nodes": plan.n_local - plan.n_own, "local_edges": plan.views.num_edges,
                  "owned_in_edges": plan.n_score, "rows_sent_per_layer": int(sum(plan.send_counts)),
                  "peak_memory_GB": torch.cuda.max_memory_allocated(dev) / 1e9,
                  "memory": _memory_record(dev, plan.views.num_edges, hidden)},
        "so_sha16": so_sha16(),
    }


def _local_degree_features(ops, g, n, plan, dev):
    """x rows of this rank's owned + halo nodes: the z-scored degrees of the WHOLE graph (inference.py:416-420), computed on this
    rank's own GPU from the edge list every rank holds."""
    x_global = ops.degree_features(ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n))
    x_local = plan.local_node_rows(x_global).contiguous()
    del x_global
    return x_local


# Peak device memory of one training step per LOCAL edge at hidden 256, torch's reserved bytes (what the caching allocator holds from
# the driver), measured on the MI355X at 2.5M and 6.25M local edges - the two sizes agree to 0.2 % (profiles/r04_bench_train_c5shard*.json,
# r04_bench_train_c4shard*.json): xe stored / recomputed in the backward (model.recompute_gate).  Nearly everything scales with the width.
TRAIN_BYTES_PER_EDGE_H256 = {"stored": 35_300, "recompute": 29_200}
DEVICE_BUDGET = 0.85 * 309.2e9    # of the 288 GiB torch reports for an MI355X; the rest is left to RCCL's buffers and the views


def train_workload_for(world, kind, one_gpu):
    """Default graph of `--mode train` at N > 1 -> (workload, recompute_gate): BASELINE configs[4] (c5) when one rank's share of
    the step fits its HBM, else the largest workload that does; xe is recomputed in the backward only where storing it does not fit
    (it costs 10 % of the step).  c5 on 8 ranks: 6.31M local edges x 35.3 KB = 223 GB of the 309."""
    cut = 1.0 + ((world - 1) / world if kind == "uniform" else 0.01)
    for name in ("c5", "c4", "10m", "c2"):
        _, e, h = WORKLOADS[name]
        for mode in ("stored", "recompute"):
            per_rank = TRAIN_BYTES_PER_EDGE_H256[mode] * (h / 256) * (e / world) * cut
            if per_rank * (world if one_gpu else 1) < DEVICE_BUDGET:
                return name, mode == "recompute"
    return "c2", False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 250 forwards at c2 (>= 1.2 s timed), fewer for the larger workloads")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: c2 at --gpus 1, 10m at --gpus > 1")
    ap.add_argument("--kind", default="banded", choices=["banded", "uniform", "permuted"],
                    help="permuted: the banded graph with shuffled read ids (locality exists, the numbering hides it)")
    ap.add_argument("--node-order", default="auto", choices=["auto", "input", "locality"],
                    help="auto (the model's default): one device statistic decides whether the node ids follow the layout; if not the reads are renumbered "
                         "once by gnnome_amd.node_order.locality_order, outside the timed region (its cost is in `cold`); locality: always; input: never")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: one forward (BASELINE configs[1]); train: the training step as the headline value (configs[2], fp32)")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"], help="train: model.activation_storage (bf16 = xe / dxe stored as bfloat16)")
    ap.add_argument("--recompute-gate", action="store_true", help="train: model.recompute_gate (xe recomputed in the backward instead of stored: "
                                                                  "-8 of ~34 KB per edge at hidden 256)")
    ap.add_argument("--symmetry", action="store_true", help="train: the reference's default step (symmetry loss: two forwards, dropout 0.2)")
    ap.add_argument("--hipgraph", action="store_true", help="replay the forward from a captured hipGraph (infer; at --gpus > 1: one graph per stretch between two collectives)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="SURVEY.md 8d's protocol: E = 1M median of 3 + one E = 10M run (minutes)")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the target_10m / train sub-records (N = 1) and the scaling reference (N > 1)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--tuning", default="", help="A/B measurement only: gnnome_set_tuning pairs 'key=value,key=value' (include/gnnome_hip.h); recorded in the line")
    ap.add_argument("--plan", default="slices", choices=["slices", "global"],
                    help="N > 1: build the partition from per-rank slices of the edge list (default) or from the whole list on every rank")
    ap.add_argument("--one-gpu-gloo", action="store_true",
                    help="plumbing check of the N>1 path on a 1-GPU box: all ranks share cuda:0, collectives go over gloo "
                         "(host-staged); the numbers it prints are NOT a multi-GPU measurement")
    args = ap.parse_args()
    if args.workload is None:
        if args.gpus > 1 and args.mode == "train":
            args.workload, need_recompute = train_workload_for(args.gpus, args.kind, args.one_gpu_gloo)
            args.recompute_gate = args.recompute_gate or need_recompute
        else:
            args.workload = "c2" if args.gpus == 1 else "10m"
    if args.cpu_baseline_only:  # child process of the cpu_baseline leg: no GPU work, bounded by the parent's timeout
        cpu_baseline(args.workload, args.kind, mode=args.mode, full=args.cpu_baseline_full)
        return
    n, e, hidden = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = max(10, min(250, int(2.5e8 // e))) if args.mode == "infer" else max(5, min(50, int(5e7 // e)))
    if args.warmup is None:
        args.warmup = max(3, args.steps // 20)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run on
        # 127.0.0.1), exactly what the driver's own launch line does; rank 0 of the children prints the JSON line
        raise SystemExit(_spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.one_gpu_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import gnnome_amd
    from gnnome_amd import _lib, ops
    for pair in filter(None, args.tuning.split(",")):
        ops.set_tuning(*(int(v) for v in pair.split("=")))
    from gnnome_amd.synth import make_graph, random_state_dict
    _lib.load()

    gloo_transport = False
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        gloo_transport = args.one_gpu_gloo
        if gloo_transport:
            _stdout_to_stderr(lambda: dist.init_process_group("gloo"))
        else:
            try:   # RCCL over xGMI; a collective is run right away so that a broken transport shows here, on every rank alike
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as ex:  # noqa: BLE001 - keep the scaling run alive on host-staged gloo and SAY so in the line
                print(f"[bench] rank {rank}: RCCL unavailable ({type(ex).__name__}: {ex}); falling back to host-staged gloo", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                _stdout_to_stderr(lambda: dist.init_process_group("gloo"))
                gloo_transport = True

    extras = {}
    g = make_graph(n, e, seed=1, kind=args.kind)
    model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
    model.load_state_dict(workload_state_dict(args.workload, hidden))
    model.to(dev)
    cold = None

    if world == 1:
        src, dst = g["src"].to(dev), g["dst"].to(dev)
        ef = g["e"].to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        node_perm, order_ms = None, None
        if args.node_order == "locality":
            from gnnome_amd import node_order as _order
            _order.locality_order(src, dst, n)   # (first call: kernel code load)
            torch.cuda.synchronize()
            t_o = time.perf_counter()
            node_perm, order_stats = _order.locality_order(src, dst, n, return_stats=True)
            torch.cuda.synchronize()
            order_ms = (time.perf_counter() - t_o) * 1e3
            extras["node_order"] = dict(order_stats, ms=order_ms, kind="locality (gnnome_amd/node_order.py), computed once per graph, not timed")
            t0 = time.perf_counter()
        elif args.node_order == "auto":   # what model.node_order = "auto" does for a cached graph object (gnnome_amd.graph.views_for)
            from gnnome_amd import node_order as _order
            node_perm, info = _order.auto_order(src, dst, n)
            torch.cuda.synchronize()
            order_ms = (time.perf_counter() - t0) * 1e3
            extras["node_order"] = dict(info, decision_and_order_ms=order_ms, kind="auto: mean edge span against N / 16, then locality_order if the ids are shuffled; once per graph, not timed")
            t0 = time.perf_counter()
        views = ops.GraphViews(src, dst, n, node_perm=node_perm)
        x = ops.degree_features(views)   # inference.py:416-420 on the device, off the views' CSR pointers
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        from gnnome_amd import engine as _engine
        _engine.prepared_for(model, dev, _engine.Prepared)   # weight preparation alone (concatenations, BatchNorm fold, uploads)
        torch.cuda.synchronize()
        t1b = time.perf_counter()
        model(views, x, ef)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # a FRESH graph in a warm process (the reference builds new graph objects per call when it masks, train.py:96,336):
        # the edge list reversed, so no cache can apply
        for _ in range(2):   # (the first of the two pays the one-time load of the deferred range check's kernel)
            t3 = time.perf_counter()
            views2 = ops.GraphViews(dst, src, n, validate="lazy")
            ops.degree_features(views2)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            del views2
        cold = {"node_order_ms": order_ms, "graph_views_and_features_ms": (t1 - t0) * 1e3, "first_call_ms": (t2 - t1) * 1e3, "weight_preparation_ms": (t1b - t1) * 1e3,
                "first_forward_after_preparation_ms": (t2 - t1b) * 1e3,
                "fresh_graph_warm_process_ms": (t4 - t3) * 1e3,
                "note": "first call = weight preparation (weight_preparation_ms) + allocator growth + kernel code load + one forward "
                        "(first_forward_after_preparation_ms); the first view build pays the process's "
                        "one-time HIP / rocPRIM initialisation, a fresh graph afterwards costs fresh_graph_warm_process_ms"}

        if args.mode == "train":
            rec = _train_record(gnnome_amd, ops, g, n, e, hidden, dev, args.steps, args.warmup, args.symmetry, 0.2 if args.symmetry else None,
                                storage=args.storage, recompute=args.recompute_gate)
            rec.update({"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                        "data": "synthetic", "config": {"workload": f"{args.workload}: {args.kind} synthetic assembly graph N={n} E={e}, SymGatedGCNModel "
                                                                    f"hidden={hidden} L=8 hs=64, {rec.pop('step')}, random-init weights seed 1",
                                                        "parallelism": "single (hipGraph replay)"}})
            if not args.no_cpu_baseline:
                rec["cpu_baseline"], err = _run_cpu_child(args, 200)
                if rec["cpu_baseline"]:
                    rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
            print(json.dumps(rec))
            return
        if args.hipgraph:
            from gnnome_amd.capture import CapturedForward
            captured = CapturedForward(model, views, x, ef)
            args.no_kernel_timers = True  # events cannot be recorded inside a replayed graph

            def step():
                return captured()
        else:
            def step():
                return model(views, x, ef)

        def barrier():
            torch.cuda.synchronize()
        parallelism = "single" + (" (hipGraph replay)" if args.hipgraph else "")
    else:
        import torch.distributed as dist
        from gnnome_amd import dist as gdist
        if rank == 0 and not args.no_extras and not args.one_gpu_gloo:
            # the same graph on ONE GPU, timed by rank 0 while the other ranks wait: strong-scaling reference for this line
            extras["scaling_reference"] = _single_gpu_forward(gnnome_amd, ops, make_graph, random_state_dict, args.workload, args.kind, dev,
                                                              max(5, args.steps // 2), 3)
        dist.barrier()
        t0 = time.perf_counter()
        if args.mode == "train":
            model.train()
            model.recompute_gate = args.recompute_gate
            if args.storage == "bf16" and not args.recompute_gate:   # (alternatives: nothing is stored when xe is recomputed)
                model.activation_storage = "bf16"   # partitions too since round 4
        node_perm = None
        if args.node_order == "locality":   # the order needs the whole graph once; every rank computes the same permutation on its GPU
            from gnnome_amd import node_order as _order
            node_perm = _order.locality_order(g["src"].to(dev), g["dst"].to(dev), n)
        elif args.node_order == "auto":   # the same statistic on every rank: the same decision and the same permutation
            from gnnome_amd import node_order as _order
            node_perm, _ = _order.auto_order(g["src"].to(dev), g["dst"].to(dev), n)
        if args.plan == "slices":
            # every rank starts from ITS 1/world of the edge list (as a reader splitting the input would): degrees all-reduced, edges
            # and their features shuffled to the owners of their endpoints (PartitionedGraph.from_slices) - no rank plans over E rows
            a, b = e * rank // world, e * (rank + 1) // world
            plan = gdist.PartitionedGraph.from_slices(g["src"][a:b], g["dst"][a:b], n, rank, world, dev, node_perm=node_perm)
            runner = gdist.PartitionedRunner(model, plan, None, None, dev, x_local=plan.local_degree_features(),
                                             e_local=plan.shuffle_edge_rows(g["e"][a:b].to(dev)))
        else:
            # every rank holds the edge list (as inference.py holds the whole graph): degree features of the WHOLE graph
            # on its own GPU, then each rank keeps the rows of its partition
            plan = gdist.PartitionedGraph.from_global(g["src"], g["dst"], n, rank, world, dev, node_perm=node_perm)
            runner = gdist.PartitionedRunner(model, plan, None, g["e"], dev, x_local=_local_degree_features(ops, g, n, plan, dev))
        torch.cuda.synchronize()
        cold = {"partition_plan_views_features_ms": (time.perf_counter() - t0) * 1e3, "plan": args.plan}
        if args.mode == "train":
            rec = _partitioned_train_record(gnnome_amd, gdist, ops, g, n, e, hidden, dev, rank, world, plan, runner, model, args, gloo_transport)
            if rank == 0:
                rec["cold"] = cold
                print(json.dumps(rec))
            dist.destroy_process_group()
            return

        if args.hipgraph:   # the kernels between two collectives replayed from one hipGraph each (dist.CapturedPartitionedForward)
            runner.capture()
            args.no_kernel_timers = True  # events cannot be recorded inside a replayed graph

        def step():
            return runner.forward()

        def barrier():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        parallelism = f"dst-range x{world}" + (" (hipGraph segments between the collectives)" if args.hipgraph else "") + (" (ALL RANKS ON ONE GPU over gloo: plumbing check, not a measurement)" if args.one_gpu_gloo
                                               else " (RCCL FAILED: host-staged gloo transport)" if gloo_transport else " over RCCL")
        # what the halo exchange moves: rows this rank sends / receives per layer, and the time of one exchange alone
        sent, recv = int(sum(plan.send_counts)), int(sum(plan.recv_counts))
        xchg = gdist.HaloExchange(plan, ops)
        h_probe = torch.zeros((plan.n_local, hidden), dtype=torch.float32, device=dev)
        for _ in range(3):
            xchg.start(h_probe)
            xchg.finish()
        barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            xchg.start(h_probe)
            xchg.finish()
        barrier()
        extras["exchange"] = {"transport": "gloo (host-staged)" if gloo_transport else "rccl", "rccl_ranks": 0 if gloo_transport else world,
                              "rank0_rows_sent_per_layer": sent, "rank0_rows_received_per_layer": recv,
                              "rank0_bytes_sent_per_layer": sent * hidden * 4, "exchanges_per_forward": 8, "ms_per_exchange_alone": (time.perf_counter() - t0) * 100,
                              "rank0_owned_nodes": plan.n_own, "rank0_halo_nodes": plan.n_local - plan.n_own,
                              "rank0_local_edges": plan.views.num_edges, "rank0_scored_edges": plan.n_score,
                              "logits": f"all_gather of {plan.score_pad * 4} B per rank + one index_select"}
        del h_probe

    # HIP events in the timed region go around ONE launch of the dominant kernel per step (the 8 layers launch
    # the same shape): a pair around every launch of every kernel cost ~1.6 ms per step here (56 events, ~28 us
    # of pipeline bubble each) and inflated what it measured; a pair per gate launch still cost ~0.4 ms.
    # (N>1: rank 0 times its own launches; its gate kernel covers the edges incident to its node range)
    # Round 5: the gate AND the aggregation carry a pair each (one launch in eight = one per step each), and the line's `roofline` is
    # whichever of the two holds the larger share of the measured step (the aggregation had overtaken the gate).
    dominant = [] if args.no_kernel_timers else ["edge_gate", "node_aggregate"]
    e_gate = e if world == 1 else plan.views.num_edges
    from gnnome_amd import engine
    one_call = world == 1 and engine.ONE_CALL_FORWARD and not args.hipgraph   # the forward is one library call: events inside it
    if one_call and dominant:
        from gnnome_amd import _lib as _glib
        ref_layers = sum(1 for lw in engine.prepared_for(model, dev, engine.Prepared).layers if lw.ref)
        fwd_timer = ForwardEventTimer(_glib.load(), args.steps, 7, {"edge_gate": 8 - ref_layers - (1 if ref_layers == 0 else 0), "node_aggregate": 8})
    else:
        fwd_timer = None
    with KernelTimer(ops, [] if fwd_timer else dominant, every=8) as kt:
        for _ in range(args.warmup):
            step()
        barrier()
        kt.on = True
        if fwd_timer:
            fwd_timer.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if fwd_timer:
                fwd_timer.before_step()
            out = step()
        t_host = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
        kt.on = False
    if fwd_timer:
        fwd_timer.close()
        kt = fwd_timer
    # what a forward costs the HOST when the launch queue is empty (in the timed region above the host runs ahead until the queue is full and
    # then waits for the GPU: host_enqueue_ms_per_step is the GPU's time there, not Python's)
    torch.cuda.synchronize()
    t0h = time.perf_counter()
    for _ in range(10):
        step()
    host_unblocked_ms = (time.perf_counter() - t0h) / 10 * 1e3
    barrier()
    # untimed diagnostic pass: every kernel family instrumented, for the per-kernel table only
    others = [] if args.no_kernel_timers or world > 1 else ["node_aggregate", "linear", "edge_score", "encode", "linear_ref", "edge_gate_ref",
                                                            "edge_gate_encode"] + (["edge_gate"] if args.workload == "ecoli" else [])
    chunks_timed = engine.PIPELINE_CHUNKS if (world == 1 and n >= engine.PIPELINE_MIN_NODES) else 1
    with KernelTimer(ops, others) as kd:
        kd.on = True
        was_one_call = engine.ONE_CALL_FORWARD
        if others:
            engine.PIPELINE_CHUNKS = 1   # the per-kernel table wants every kernel alone on the chip, one launch per layer
            engine.ONE_CALL_FORWARD = False   # ... and launched call by call, so that each can carry its own pair of events
        try:
            for _ in range(min(args.steps, 5)):
                step()
        finally:
            engine.PIPELINE_CHUNKS = chunks_timed if chunks_timed > 1 else engine.PIPELINE_CHUNKS
            engine.ONE_CALL_FORWARD = was_one_call
        barrier()
    timed = dominant
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo_transport else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        b_fwd, f_fwd = algorithmic_bytes(n, e, hidden), algorithmic_flops(n, e, hidden)
        res = {
            "metric": "edges/sec full-graph GatedGCN fwd", "value": e / (ms * 1e-3), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.kind} synthetic assembly graph N={n} E={e}, SymGatedGCNModel hidden={hidden} "
                                   f"L=8 hs=64, BatchNorm(eval), fwd only, " + ("the reference's shipped checkpoint (weights/weights.pt), arithmetic=auto"
                                                                                 if args.workload == "ecoli" else "random-init weights seed 1"),
                       "parallelism": parallelism},
            "hbm_roofline_frac_whole_fwd": (b_fwd / (ms * 1e-3)) / (world * HBM_PEAK),
            "mfma_f32_frac_whole_fwd": (f_fwd / (ms * 1e-3)) / (world * MFMA_F32_PEAK),
            "algorithmic_bytes_fwd": b_fwd, "algorithmic_flops_fwd": f_fwd, "cold": cold, "timed_region_s": elapsed,
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3, "host_ms_per_forward_queue_empty": host_unblocked_ms,
            "placement": _placement_record(ops) if one_call else None,
            "forward_entry": ("gnnome_model_forward_buffers_f32 (one library call per forward, buffers allocated one by one)" if ops.FORWARD_BUFFERS != "block" else "gnnome_model_forward_f32 (one library call per forward, one workspace block)") if one_call else "per-kernel entries, call by call",
            "so_sha16": so_sha16(),
            "streams": (f"2: every node projection after the first runs on a second HIP stream under the aggregation, which is cut into "
                        f"{chunks_timed} node ranges (engine.aggregate_then_project)") if chunks_timed > 1 else "1",
        }
        res.update(extras)
        res["arithmetic"] = ("fp32 in / fp32 out; dense products as an exact-split fp32 emulation on the 16-bit matrix cores with fp32 accumulation: fp16x3 "
                             "(two fp16 planes per operand, three MFMAs; measured no further from an fp64 product than an fp32 GEMM - tests/test_f16x3_model.py) "
                             "for the forward at H >= 128, bf16x6 elsewhere; --tuning 10=1 = bf16x6 everywhere (round 3)")
        res["config"]["node_order"] = args.node_order + (f" -> {extras['node_order'].get('decision')}" if args.node_order == "auto" and "node_order" in extras else "")
        if args.tuning:
            res["tuning"] = args.tuning   # not the shipped defaults
        if "scaling_reference" in extras:
            res["speedup_over_one_gpu_same_graph"] = extras["scaling_reference"]["ms_per_step"] / ms
            if world > 1 and "exchange" in extras:
                # DESIGN.md section 5's prediction next to what was measured, so that one SCALE record confirms or falsifies it by itself: rank 0's
                # share of the edges at the one-GPU rate + the per-layer halo exchanges and the logits gather at 0.1 ms each (what a small RCCL
                # collective over xGMI is ASSUMED to cost; measured alone: exchange.ms_per_exchange_alone)
                t1, share, n_coll = extras["scaling_reference"]["ms_per_step"], plan.views.num_edges / e, 8 + 1
                pred = t1 * share + n_coll * 0.1
                res["prediction"] = {
                    "model": "one-GPU ms x (rank 0's local edges / E) + (8 halo exchanges + 1 logits all_gather) x 0.1 ms, exchanges not overlapped",
                    "one_gpu_ms": t1, "rank0_edge_share": share, "collectives_per_forward": n_coll, "assumed_ms_per_small_collective": 0.1,
                    "predicted_ms": pred, "predicted_speedup": t1 / pred, "measured_ms": ms, "measured_over_predicted": ms / pred,
                    "measured_ms_per_exchange_alone": extras["exchange"]["ms_per_exchange_alone"],
                    "valid": "only over RCCL (exchange.rccl_ranks == n_gpus); a host-staged gloo run is a plumbing check"}
        if timed and kt.events[timed[0]]:
            gate_ms, gate_n = kt.mean_ms(timed[0])
            gate_flops = 2.0 * e_gate * hidden * hidden
            gate_bytes = 2.0 * e_gate * hidden * 4 + 2 * e_gate * 4   # read e, write e', read src/dst (B_layer's edge part)
            # the kernel's own operands touched once: + the two node tables it gathers from (B1h, B2h) and W3 - the same two figures the aggregation's record carries
            gate_operand_bytes = gate_bytes + 2.0 * (n if world == 1 else plan.n_local) * hidden * 4 + hidden * hidden * 4.0
            # round 4: the forward's dense products run as fp16x3 (two fp16 planes per fp32 operand, three f16 MFMAs per K = 16 step - csrc/edge_tile_f16.hip)
            # unless --tuning 10=1 asks for round 3's bf16x6 (six); H = 64 still runs bf16x6
            terms = 6.0 if (hidden == 64 or "10=1" in (args.tuning or "").replace(" ", "")) else 3.0
            arith = "bf16x6" if terms == 6.0 else "fp16x3"
            if hidden in (64, 128):
                # the product costs `terms` 16-bit MFMAs per K=16 (98 GF per launch at configs[1] as fp16x3 = 0.04 ms at the 2.5 PF peak)
                # against 1.03 GB = 0.13 ms at 8 TB/s: HBM is the bound
                res["roofline"] = {
                    "kernel": ("k_edge_gate_pl" if hidden == 128 else "k_edge_gate_bf") + f" (fused B_3 GEMM as {arith} + u_add_v + bn_e + relu + residual)",
                    "bound": "hbm", "achieved": gate_bytes / (gate_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": gate_bytes / (gate_ms * 1e-3) / HBM_PEAK,
                    "traffic": _pmc_traffic(args.workload, "k_edge_gate_pl<false, 0," if hidden == 128 else "k_edge_gate_bf") if (world == 1 and args.kind == "banded") else None,
                    "avg_launch_ms": gate_ms, "launches": gate_n, "algorithmic_bytes_per_launch": gate_bytes, "operand_bytes_per_launch": gate_operand_bytes,
                    "fp32_equivalent_flops_per_launch": gate_flops, "fp32_equivalent_tflops": gate_flops / (gate_ms * 1e-3) / 1e12,
                    "mfma_16bit_frac": terms * gate_flops / (gate_ms * 1e-3) / MFMA_BF16_PEAK, "arithmetic": arith,
                }
            elif terms == 3.0:
                # H = 256, fp16x3 + LDS-DMA (k_edge_tile_f16): 3 f16 MFMAs per K=16 = 0.39 ms per launch at the 2.5 PF peak against 0.64 ms of HBM
                # time for this shard (8 TB/s): HBM is the bound now
                res["roofline"] = {
                    "kernel": "k_edge_tile_f16 (H=256: fp16x3, e tiles by LDS-DMA, planes in place, W3 in registers, two workgroups per row)",
                    "bound": "hbm", "achieved": gate_bytes / (gate_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": gate_bytes / (gate_ms * 1e-3) / HBM_PEAK,
                    "traffic": _pmc_traffic(args.workload, "k_edge_tile_f16<0") if (world == 1 and args.kind == "banded") else None,
                    "avg_launch_ms": gate_ms, "launches": gate_n, "algorithmic_bytes_per_launch": gate_bytes, "operand_bytes_per_launch": gate_operand_bytes,
                    "fp32_equivalent_tflops": gate_flops / (gate_ms * 1e-3) / 1e12,
                    "mfma_16bit_frac": terms * gate_flops / (gate_ms * 1e-3) / MFMA_BF16_PEAK, "arithmetic": arith,
                    "bf16x6_equivalent_mfma_frac": 6.0 * gate_flops / (gate_ms * 1e-3) / MFMA_BF16_PEAK,
                }
            else:
                # H = 256: the wave-specialised plane form (bf16x6, W3 in registers, two workgroups per row); 6 bf16 MFMAs per K=16
                # = 0.79 ms per launch at the 2.5 PF peak against 0.64 ms of HBM time for this shard: the matrix cores are the bound
                res["roofline"] = {
                    "kernel": "k_edge_gate_pl256 (H=256: W3 as bf16 planes in registers, e tiles split once into LDS planes by the load waves, bf16x6)",
                    "bound": "mfma", "achieved": 6.0 * gate_flops / (gate_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": 6.0 * gate_flops / (gate_ms * 1e-3) / MFMA_BF16_PEAK, "traffic": None,
                    "avg_launch_ms": gate_ms, "launches": gate_n, "bf16_flops_per_launch": 6.0 * gate_flops,
                    "fp32_equivalent_tflops": gate_flops / (gate_ms * 1e-3) / 1e12,
                    "algorithmic_bytes_per_launch": gate_bytes, "operand_bytes_per_launch": gate_operand_bytes, "hbm_frac": gate_bytes / (gate_ms * 1e-3) / HBM_PEAK, "arithmetic": arith,
                }
            if world > 1:
                res["roofline"]["note"] = f"rank 0's launches: {e_gate} local edges (its node range's in- and out-edges)"
            # the aggregation, timed in the same region on the same stream (one launch per step): if it holds the larger share of the
            # step, IT is the line's roofline and the gate becomes the second record
            n_agg = n if world == 1 else plan.n_own
            if kt.events.get("node_aggregate"):
                agg_ms_t, agg_n_t = kt.mean_ms("node_aggregate")
                # algorithmic bytes (SURVEY.md 8d counts e' once per layer): e' read once + 3 index arrays + three node tables' worth of
                # rows; the kernel's own operands touched once (4 node-table reads + h' written) are `operand_bytes_per_launch`
                agg_bytes_t = 1.0 * e_gate * hidden * 4 + 3 * e_gate * 4 + 3 * n_agg * hidden * 4
                agg_rec = {
                    "kernel": "k_node_aggregate (sigmoid + both gated sums over in- / out-edges + node update, one wave per node; e' rows read by both passes)",
                    "bound": "hbm", "achieved": agg_bytes_t / (agg_ms_t * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": agg_bytes_t / (agg_ms_t * 1e-3) / HBM_PEAK,
                    "traffic": _pmc_traffic(args.workload, "k_node_aggregate") if (world == 1 and args.kind == "banded") else None,
                    "avg_launch_ms": agg_ms_t, "launches": agg_n_t, "algorithmic_bytes_per_launch": agg_bytes_t,
                    "operand_bytes_per_launch": 1.0 * e_gate * hidden * 4 + 3 * e_gate * 4 + 5 * n_agg * hidden * 4,
                }
                gate_share = kt.calls["edge_gate"] / args.steps * gate_ms / ms   # (launches per step x average launch; layer 0 runs the folded-encoder gate)
                agg_share = kt.calls["node_aggregate"] / args.steps * agg_ms_t / ms
                res["roofline"]["share_of_step"], agg_rec["share_of_step"] = gate_share, agg_share
                if agg_share > gate_share:
                    res["roofline"], res["roofline_second"] = agg_rec, res["roofline"]
                else:
                    res["roofline_second"] = agg_rec
                res["roofline"]["chosen_by"] = "the larger measured share of the step among the two big kernels (HIP events on one launch of each per step, inside the timed region)"
        if timed and others:
            agg_ms, agg_n = kd.mean_ms("node_aggregate")
            # e' is read ONCE algorithmically (SURVEY.md 8d's B_layer has no second read of it); the kernel's in- and out-edge
            # passes each stream it, which shows up as traffic, not as algorithmic bytes
            agg_bytes = 1.0 * e * hidden * 4 + 3 * e * 4 + 3 * n * hidden * 4
            lin_ms, lin_n = kd.mean_ms("linear")
            sc_ms, sc_n = kd.mean_ms("edge_score")
            en_ms, en_n = kd.mean_ms("encode")
            res["kernels_note"] = "measured in a separate untimed pass with HIP events around every launch (inflates each by a few %)"
            res["kernels"] = [
                {"kernel": "k_node_aggregate", "bound": "hbm", "avg_launch_ms": agg_ms, "launches": agg_n,
                 "achieved": agg_bytes / (agg_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "frac": agg_bytes / (agg_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": agg_bytes},
                {"kernel": "linear (all calls: node projections [N,H]x[H,5H] and predictor node halves)", "bound": "hbm",
                 "avg_launch_ms": lin_ms, "launches": lin_n},
                {"kernel": "k_edge_score_ws<H, fp16x3> (weight-stationary streaming scorer: e W1e^T and the 64 -> 32 layer as fp16x3 on the f16 matrix cores, fp32 tail)", "bound": "hbm", "avg_launch_ms": sc_ms, "launches": sc_n,
                 "achieved": (e * hidden * 4.0 + 3 * e * 4.0) / (sc_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "frac": (e * hidden * 4.0 + 3 * e * 4.0) / (sc_ms * 1e-3) / HBM_PEAK},
                {"kernel": "k_encode (node + edge)", "bound": "hbm", "avg_launch_ms": en_ms, "launches": en_n},
            ]
        if world == 1 and args.workload in ("c2", "c4shard"):
            mu = _mfma_util(f"{args.workload}_infer")
            if mu:
                res["mfma_util"] = mu
        if timed and others and kd.events["edge_gate_ref"]:
            # layers that run in the reference's ORDER of evaluation (fp32 VALU, csrc/reference_order.hip) next to the bf16x6
            # matrix-core kernels they stand in for, same shapes, same pass
            ref_gate, n_rg = kd.mean_ms("edge_gate_ref")
            ref_lin, n_rl = kd.mean_ms("linear_ref")
            fast_gate, n_fg = kd.mean_ms("edge_gate")
            per_fwd = max(min(args.steps, 5), 1)
            ref_ms = (ref_gate * n_rg + ref_lin * n_rl) / per_fwd
            res["reference_order"] = {
                "layers": n_rg // per_fwd, "k_edge_gate_ref_ms": ref_gate, "k_linear_ref_ms": ref_lin,
                "bf16x6_gate_ms_same_shape": fast_gate if n_fg else None, "bf16x6_projection_ms_same_shape": lin_ms,
                "ms_per_forward_in_reference_order_kernels": ref_ms, "valu_fp32_frac_of_forward": ref_ms / ms,
                "layer_cost_ratio_ref_over_fast": ((ref_gate + ref_lin) / (fast_gate + lin_ms)) if n_fg else None,
                "note": "arithmetic='auto': a layer whose eval-BatchNorm gain exceeds engine.REFERENCE_ORDER_GAIN evaluates its dense products "
                        "as k-ascending fp32 fma chains (torch-CPU / MKL's order) so that the 1e-4 bar on probabilities holds against the reference",
            }
        if world == 1:
            del views, x, ef
            torch.cuda.empty_cache()
            if not args.no_extras and args.workload == "c2":
                res["target_10m"] = _single_gpu_forward(gnnome_amd, ops, make_graph, random_state_dict, "10m", args.kind, dev, 20, 3)
                # the width of configs[3] / configs[4], driver-timed: one GPU's eighth of configs[3] (2.5M edges, H = 256)
                res["h256_shard"] = _single_gpu_forward(gnnome_amd, ops, make_graph, random_state_dict, "c4shard", args.kind, dev, 20, 3)
                res["train"] = _train_record(gnnome_amd, ops, g, n, e, hidden, dev, 20, 3, False, None)
                t16 = _train_record(gnnome_amd, ops, g, n, e, hidden, dev, 20, 3, False, None, storage="bf16")
                res["train"]["bf16_storage"] = {k: t16[k] for k in ("value", "ms_per_step", "loss", "activation_storage")}
            if not args.no_cpu_baseline:
                # in a child process (own thread pool, hard time limit): the baseline must never stall the bench line
                res["cpu_baseline"], err_cpu = _run_cpu_child(args, 1500 if args.cpu_baseline_full else 400)
                if res["cpu_baseline"]:
                    res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
                    if "target_10m" in res:   # edges/s is size-normalised: the 10M-edge GPU rate over the CPU rate of the sample
                        cpu10 = res["cpu_baseline"].get("e10m") or res["cpu_baseline"]
                        res["target_10m"]["gpu_over_cpu"] = res["target_10m"]["value"] / cpu10["value"]
                        res["target_10m"]["cpu_sample_edges"] = cpu10.get("sample_edges")   # the CPU rate this ratio divides by was measured at this size
                        res["target_10m"]["north_star_target"] = ">= 10x the CPU path's edges/s on the 10M-edge graph at 1 GPU"
                else:
                    res["cpu_baseline_error"] = err_cpu[-300:]
        print(json.dumps(res))

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
