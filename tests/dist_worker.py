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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import cpu_ops
        import gnnome_amd
        from gnnome_amd import dist as gdist
        from gnnome_amd import engine
        g = torch.load(case, weights_only=False)
        m = gnnome_amd.models.SymGatedGCNModel(2, 2, g["hidden"], 16, g["layers"], 64, "batch").eval()
        m.load_state_dict(g["state_dict"])
        cpu = torch.device("cpu")
        part = gdist.PartitionedGraph.from_global(g["src"], g["dst"], g["num_nodes"], rank, world, cpu, ops=cpu_ops)
        prep = engine.Prepared(m, cpu)
        with torch.no_grad():
            logits = gdist.run_partitioned(cpu_ops, prep, part, part.local_node_rows(g["x"]), part.local_edge_rows(g["e"]))
        torch.save({"logits": logits, "n_own": part.n_own, "n_local": part.n_local, "n_score": part.n_score,
                    "e_local": int(part.edge_gid.numel()), "bounds": part.bounds, "send": part.send_counts,
                    "recv": part.recv_counts}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()
