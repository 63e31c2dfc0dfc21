"""World-size-N worker for tests/test_dist_gloo.py: runs gnnome_amd.dist's partition + halo exchange on CPU
ranks (gloo) with the checker backend as the per-rank compute, and returns each rank's assembled logits."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    g = torch.load(case, weights_only=False)
    if g.get("force_collectives"):   # one rank that still issues every collective (gnnome_amd.dist.force_collectives)
        os.environ["GNNOME_FORCE_COLLECTIVES"] = "1"
    if g.get("transport") == "nccl":   # RCCL on device memory: one rank per GPU (the test box has one GPU => world 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import cpu_ops
        import gnnome_amd
        from gnnome_amd import dist as gdist
        from gnnome_amd import engine
        if g.get("wrappers"):   # the collective wrappers alone, on rows that really travel (a rank sends to itself at world 1)
            where = torch.device("cuda", rank) if g.get("device") == "cuda" else torch.device("cpu")
            gen = torch.Generator().manual_seed(5 + rank)
            rows = torch.randn(world * 300, 128, generator=gen).to(where)
            got = torch.empty_like(rows)
            work = gdist.all_to_all_rows(got, rows, [300] * world, [300] * world, None, async_op=True)
            if work is not None:
                work.wait()
            summed = gdist.all_reduce_sum(rows[:7].clone())
            every = gdist.all_gather_rows(rows[:5], world)
            ints = gdist.all_gather_rows(torch.arange(4, dtype=torch.int64, device=where) + rank, world)
            torch.save({"sent": rows.cpu(), "got": got.cpu(), "summed": summed.cpu(), "every": every.cpu(), "ints": ints.cpu(),
                        "backend": dist.get_backend()}, os.path.join(out_dir, f"rank{rank}.pt"))
            return
        m = gnnome_amd.models.SymGatedGCNModel(2, 2, g["hidden"], 16, g["layers"], g.get("hs", 64), g.get("normalization", "batch"), dropout=0.0).eval()
        m.load_state_dict(g["state_dict"])
        if g.get("device") == "cuda":
            # both ranks on the one GPU of the test box: HIP kernels as compute, gloo (host-staged) as transport
            from gnnome_amd import ops as backend
            where = torch.device("cuda", rank if g.get("transport") == "nccl" else 0)
            m.to(where)
            g = {k: (v.to(where) if k in ("x", "e", "y", "pos_weight") else v) for k, v in g.items()}
        else:
            backend, where = cpu_ops, torch.device("cpu")
        part = gdist.PartitionedGraph.from_global(g["src"], g["dst"], g["num_nodes"], rank, world, where, ops=backend,
                                                  node_perm=g.get("node_perm"))
        if g.get("sliced"):
            # the same plan from this rank's SLICE of the edge list (uneven slices, rank order): must equal from_global's, array by array
            E = int(g["src"].numel())
            cuts = [0] + [min(E, (E * (r + 1)) // world + (37 if r + 1 < world else 0)) for r in range(world)]
            a, b = cuts[rank], cuts[rank + 1]
            sl = gdist.PartitionedGraph.from_slices(g["src"][a:b], g["dst"][a:b], g["num_nodes"], rank, world, where, ops=backend,
                                                    node_perm=g.get("node_perm"))
            same = all(torch.equal(getattr(sl, k).cpu(), getattr(part, k).cpu()) for k in ("node_gid", "edge_gid", "srt_geid", "send_idx")) and \
                all(getattr(sl, k) == getattr(part, k) for k in ("bounds", "n_own", "n_local", "n_score", "send_counts", "recv_counts", "score_pad",
                                                               "num_edges_global")) and \
                all(torch.equal(getattr(sl.views, k).cpu(), getattr(part.views, k).cpu()) for k in ("in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos"))
            e_rows = sl.shuffle_edge_rows(g["e"][a:b])
            same = same and torch.equal(e_rows.cpu(), part.local_edge_rows(g["e"]).cpu())
            x_rows = sl.local_degree_features()
            same = same and torch.allclose(x_rows.cpu(), part.local_node_rows(g["x"]).cpu(), atol=2e-6, rtol=1e-6)
            part_sliced_equal = bool(same)
            part = sl      # and the run below uses the sliced plan
        if g.get("train"):
            # train.py:138-145 + :328-330 on the partition: every rank computes the loss on the assembled logits
            import torch.nn.functional as F
            m.train()
            m.recompute_gate = bool(g.get("recompute_gate"))
            m.activation_storage = g.get("activation_storage", "fp32")
            runner = gdist.PartitionedRunner(m, part, g["x"], g["e"], where, ops=backend)
            logits = runner.train_forward()
            loss = F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"])
            loss.backward()
            torch.save({"logits": logits.detach().squeeze(1).cpu(), "loss": loss.detach().cpu(),
                        "grads": {k: p.grad.cpu() for k, p in m.named_parameters()},
                        "buffers": {k: b.detach().cpu().clone() for k, b in m.named_buffers()}, "n_own": part.n_own, "n_local": part.n_local,
                        "n_score": part.n_score, "e_local": int(part.edge_gid.numel()), "backend": dist.get_backend()},
                       os.path.join(out_dir, f"rank{rank}.pt"))
            return
        prep = engine.Prepared(m, where)
        with torch.no_grad():
            logits = gdist.run_partitioned(backend, prep, part, part.local_node_rows(g["x"]).contiguous(),
                                           part.local_edge_rows(g["e"]).contiguous())
        replayed = None
        if g.get("captured"):   # the same forward as hipGraph segments between the collectives, replayed twice
            runner = gdist.PartitionedRunner(m, part, g["x"], g["e"], where, ops=backend).capture()
            runner.forward()
            replayed = runner.forward().squeeze(1).cpu()
        torch.save({"logits": logits.cpu(), "replayed": replayed, "n_own": part.n_own, "n_local": part.n_local, "n_score": part.n_score,
                    "e_local": int(part.edge_gid.numel()), "bounds": part.bounds, "send": part.send_counts,
                    "recv": part.recv_counts, "backend": dist.get_backend(), "score_index": part.score_index is not None,
                    "sliced_equal": part_sliced_equal if g.get("sliced") else None},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()
