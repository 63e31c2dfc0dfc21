"""GFA -> the arrays the scoring path and the decoder need, without DGL / networkx / Biopython
(graph_parser.py:120-411, only_from_gfa).

    g = read_gfa("asm.bp.raw.r_utg.noseq.gfa")
    g["src"], g["dst"]                       edge list in the order DGL would number the edges (see below)
    g["overlap_length"], g["prefix_length"]  per edge;  g["read_length"] per node
    g["overlap_similarity"]                  per edge, or None (see below)
    views = gnnome_amd.graph.views_for((g["src"], g["dst"], g["num_nodes"]), device)
    x = features.degree_features(views);  e = features.edge_features(g["overlap_length"], g["overlap_similarity"])

What is reproduced from the reference, line by line:
  * every `S` line makes two nodes, 2k (the read) and 2k+1 (its reverse complement), in file order (:167-181);
    `LN:i:` gives both their length (:173, :186-187);
  * `A` lines following a `utg*` segment are consumed and recorded in read_to_node2 / node_to_read (:189-208);
  * `L` lines with 6 (raven / GFA 1), 7 (hifiasm: the `:a-b` suffix of the ids is dropped) or 8 (newer hifiasm) fields
    (:278-290); the overlap length is the integer in front of the CIGAR's letter (:292-296); zero-length overlaps are
    skipped (:299-300); the four orientation cases give the edge and its reverse-complement mate (:302-321);
  * the graph is a networkx DiGraph there: a repeated (u, v) is ONE edge whose attributes are the last ones written
    (:323-340), and prefix_length = read_length[src] - overlap_length (:339-340);
  * edge numbering: dgl.from_networkx relabels the nodes in sorted order and numbers the edges as networkx iterates
    them - by source node, and for one source in the order its out-edges were first inserted (:407).  That order is
    what `src`, `dst` and every per-edge array here follow.  (DGL itself cannot be run here; this is its documented
    behaviour, pinned in tests/golden/g10_gfa.pt through networkx's own iteration order - see make_golden_gfa.py.)

overlap_similarity (:101-117, :372-376) is 1 - editDistance(suffix, prefix) / overlap_length, which the reference gets
from the third-party aligner edlib on the CPU.  Here (`similarity="auto"`, the default) it comes, in this order, from
`SI:f:` tags on the `L` lines where the GFA carries them (our own extension, written by `write_similarity_tags`), else - for
a GFA with sequences - from the device kernel gnnome_overlap_edit_distance (gnnome_amd/overlap.py: exact edit distances,
one wavefront per overlap; needs the MI355X, there is no CPU version in this package); a caller-supplied
`similarity(src_seq, dst_seq, overlap_length)` callable overrides both.  A GFA without sequences and without tags gives
None - never a guess (hyperparameters.py:17 `use_similarities`: a model trained with them needs them)."""
import gzip
import re

import torch

_HIFIASM_ID = re.compile(r"(.*):\d-\d*")
# Bio.Seq.reverse_complement's table (IUPAC ambiguity codes, both cases) - the same one gnnome_amd/overlap.py hands the device
_COMPLEMENT = str.maketrans("ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna")


def read_gfa(path, similarity="auto", keep_sequences=False):
    """-> dict(src, dst int64[E]; num_nodes; overlap_length, prefix_length int64[E]; read_length int64[N];
    overlap_similarity float32[E] | None; read_to_node, node_to_read, read_to_node2; read_seqs | None)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        lines = f.readlines()
    read_to_node, node_to_read, read_to_node2 = {}, {}, {}
    read_lengths, read_seqs, forward = [], {}, []
    adj = []                      # adj[u]: dict v -> (overlap_length, similarity tag or None); insertion-ordered like networkx's adjacency
    no_seqs = False
    node_idx, i = 0, 0
    while i < len(lines):
        line = lines[i].strip().split()
        i += 1
        if not line:
            continue
        if line[0] == "S":
            _, rid, sequence, length = line[:4]
            if sequence == "*":
                no_seqs = True
            real, virt = node_idx, node_idx + 1
            read_to_node[rid] = (real, virt)
            node_to_read[real] = node_to_read[virt] = rid
            if similarity not in (None, False):
                forward.append(sequence)
            if keep_sequences or callable(similarity):
                read_seqs[real] = sequence
                read_seqs[virt] = sequence.translate(_COMPLEMENT)[::-1]
            ln = int(length[5:])
            read_lengths += [ln, ln]
            adj += [{}, {}]
            if rid.startswith("utg"):
                ids = []
                while i < len(lines):
                    nxt = lines[i].strip().split()
                    if not nxt or nxt[0] != "A":
                        break
                    i += 1
                    ids.append((nxt[4], nxt[3]))
                    read_to_node2[nxt[4]] = (real, virt)
                node_to_read[real] = node_to_read[virt] = ids
            node_idx += 2
        elif line[0] == "L":
            tags = []
            if len(line) >= 6 and any(t.startswith("SI:f:") for t in line[6:]):
                tags = [t for t in line[6:] if t.startswith("SI:f:")]
                line = [t for t in line if not t.startswith("SI:f:")]
            if len(line) == 6:
                _, id1, o1, id2, o2, cigar = line
            elif len(line) == 7:
                _, id1, o1, id2, o2, cigar, _ = line
                id1, id2 = _HIFIASM_ID.findall(id1)[0], _HIFIASM_ID.findall(id2)[0]
            elif len(line) == 8:
                _, id1, o1, id2, o2, cigar, _, _ = line
            else:
                raise ValueError("Unknown GFA format!")
            ol = int(cigar[:-1])
            if ol == 0:
                continue
            a, b = read_to_node[id1], read_to_node[id2]
            if o1 == "+" and o2 == "+":
                sr, dr, sv, dv = a[0], b[0], b[1], a[1]
            elif o1 == "+" and o2 == "-":
                sr, dr, sv, dv = a[0], b[1], b[0], a[1]
            elif o1 == "-" and o2 == "+":
                sr, dr, sv, dv = a[1], b[0], b[1], a[0]
            else:
                sr, dr, sv, dv = a[1], b[1], b[0], a[0]
            sim = float(tags[0][5:]) if tags else None
            adj[sr][dr] = (ol, sim)     # a repeated pair keeps its first position and takes the last attributes (networkx)
            adj[sv][dv] = (ol, sim)
    src, dst, ols, sims = [], [], [], []
    for u, nbrs in enumerate(adj):
        for v, (ol, sim) in nbrs.items():
            src.append(u)
            dst.append(v)
            ols.append(ol)
            sims.append(sim)
    read_length = torch.tensor(read_lengths, dtype=torch.int64)
    src_t, dst_t = torch.tensor(src, dtype=torch.int64), torch.tensor(dst, dtype=torch.int64)
    ol_t = torch.tensor(ols, dtype=torch.int64)
    out = {"src": src_t, "dst": dst_t, "num_nodes": node_idx, "overlap_length": ol_t,
           "prefix_length": (read_length[src_t] - ol_t) if src else ol_t.clone(), "read_length": read_length,
           "read_to_node": read_to_node, "node_to_read": node_to_read, "read_to_node2": read_to_node2,
           "read_seqs": read_seqs if keep_sequences else None, "overlap_similarity": None}
    if callable(similarity) and not no_seqs:    # an explicit aligner overrides SI:f: tags (the docstring said so, the code did not)
        out["overlap_similarity"] = torch.tensor([similarity(read_seqs[u], read_seqs[v], ol) if ol > 0 else 0.5 for u, v, ol in zip(src, dst, ols)],
                                                 dtype=torch.float32)
    elif sims and all(s is not None for s in sims):
        out["overlap_similarity"] = torch.tensor(sims, dtype=torch.float32)
    elif similarity not in (None, False) and not no_seqs and (similarity == "device" or torch.cuda.is_available()):
        from .overlap import overlap_similarity   # the MI355X kernel
        if similarity == "device":              # "device" insists: a missing library, > 32 symbols or an overlap beyond 65 536 bases raise
            out["overlap_similarity"] = overlap_similarity(forward, src_t, dst_t, ol_t).cpu()
        else:                                   # "auto": what cannot be aligned here is reported and left to the caller, as before the kernel existed
            try:
                out["overlap_similarity"] = overlap_similarity(forward, src_t, dst_t, ol_t).cpu()
            except (RuntimeError, ValueError, OSError) as ex:
                import warnings
                warnings.warn(f"read_gfa: overlap similarities not computed on the device ({ex}); overlap_similarity is None")
    return out


def write_similarity_tags(gfa_in, gfa_out, similarity):
    """Copy a GFA, appending `SI:f:<similarity>` to every L line: a way to carry similarities computed once (with edlib,
    on a machine that has it) inside the GFA, so that this reader needs no aligner.  `similarity[(src, dst)]` is keyed by
    node ids as read_gfa numbers them."""
    g = read_gfa(gfa_in, similarity=None)
    r2n = g["read_to_node"]
    with open(gfa_in) as f, open(gfa_out, "w") as o:
        for raw in f:
            line = raw.strip().split()
            if line and line[0] == "L":
                id1, o1, id2, o2 = line[1:5]
                if len(line) == 7:
                    id1, id2 = _HIFIASM_ID.findall(id1)[0], _HIFIASM_ID.findall(id2)[0]
                a, b = r2n[id1], r2n[id2]
                key = (a[0] if o1 == "+" else a[1], b[0] if o2 == "+" else b[1])
                if key in similarity:
                    raw = raw.rstrip("\n") + f"\tSI:f:{similarity[key]:.9g}\n"
            o.write(raw)
