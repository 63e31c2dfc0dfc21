"""DGL graphs -> the arrays this package works on ("a converter from DGL `.dgl` files when DGL is importable").

The reference's datasets are `.dgl` files written by `dgl.save_graphs` (create_inference_graphs.py:27) and read back with
`dgl.load_graphs(path)[0][0]` (graph_dataset.py:50); `preprocess_graph` (utils/data_utils.py:31-41) then builds
`edata['e'] = [zscore(overlap_length) | overlap_similarity]`.  DGL is not a dependency of this package and is not in this
image: `from_dgl_graph` takes any DGLGraph-shaped object (edges(), num_nodes(), ndata / edata mappings - a real DGLGraph on a
machine that has DGL, the test double here), `load_dgl_file` imports dgl lazily and says so if it is missing.

    g = dgl_io.load_dgl_file("processed/0.dgl")                  # needs `import dgl`
    views = gnnome_amd.graph.views_for((g["src"], g["dst"], g["num_nodes"]), device)
    x = features.degree_features(views);  e = features.edge_features(g["overlap_length"].to(device), g["overlap_similarity"].to(device))
    logits = model(views, x, e)                                  # inference.py:413-441
The dict has the keys gnnome_amd.gfa.read_gfa returns, so `pipeline.score_graph` / `pipeline.assemble` take it as is.
"""
import torch


def from_dgl_graph(graph):
    """DGLGraph (or anything with edges() / num_nodes() / ndata / edata) -> dict(src, dst int64[E], num_nodes, overlap_length,
    overlap_similarity, prefix_length, read_length; entries the graph does not carry are None).  Edge order = DGL edge ids."""
    src, dst = graph.edges()
    src, dst = torch.as_tensor(src).long().cpu(), torch.as_tensor(dst).long().cpu()
    edata, ndata = getattr(graph, "edata", {}), getattr(graph, "ndata", {})

    def edge(key, dtype):
        return torch.as_tensor(edata[key]).to(dtype).cpu().reshape(-1) if key in edata else None

    def node(key, dtype):
        return torch.as_tensor(ndata[key]).to(dtype).cpu().reshape(-1) if key in ndata else None

    out = {"src": src, "dst": dst, "num_nodes": int(graph.num_nodes()),
           "overlap_length": edge("overlap_length", torch.int64), "overlap_similarity": edge("overlap_similarity", torch.float32),
           "prefix_length": edge("prefix_length", torch.int64), "read_length": node("read_length", torch.int64),
           # training labels and stored degrees where the graph has them (graph_parser.py: 'y'; data_utils.py:50-51: in_deg / out_deg)
           "y": edge("y", torch.float32), "in_deg": node("in_deg", torch.float32), "out_deg": node("out_deg", torch.float32),
           "read_to_node": None, "node_to_read": None, "read_to_node2": None, "read_seqs": None}
    for key, t in (("overlap_length", out["overlap_length"]), ("overlap_similarity", out["overlap_similarity"]), ("prefix_length", out["prefix_length"])):
        if t is not None and t.numel() != src.numel():
            raise ValueError(f"edata[{key!r}] has {t.numel()} entries for {src.numel()} edges")
    return out


def load_dgl_file(path, index=0):
    """graph_dataset.py:50 - `dgl.load_graphs(path)[0][index]` - converted with from_dgl_graph.  Needs DGL (the reference pins
    dgl==0.8.1, requirements_cpu.txt:23); raises ImportError with that pointer where it is not installed."""
    try:
        import dgl
    except ImportError as ex:
        raise ImportError("reading .dgl files needs the DGL package the files were written with (the reference pins dgl==0.8.1); "
                          "without it, start from the GFA: gnnome_amd.gfa.read_gfa") from ex
    return from_dgl_graph(dgl.load_graphs(str(path))[0][index])
