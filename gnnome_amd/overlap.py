"""overlap_similarity on the device (graph_parser.py:101-117, calculate_similarities).

    sim = overlap_similarity(reads, src, dst, overlap_length)          # float32[E], on the device

`reads[r]` is read r as its S line gives it; node 2r is that read, node 2r+1 its reverse complement (graph_parser.py:174-181,
:365).  For every edge the reference computes `1 - edlib.align(src_seq[-ol:], dst_seq[:ol])['editDistance'] / ol` (0.5 where
ol == 0) on the CPU with the third-party aligner edlib; here the exact edit distances come from
gnnome_overlap_edit_distance (csrc/overlap_similarity.hip: Myers' bit-vector programme - an Ukkonen band of 256 rows, one
thread per overlap, settles every overlap whose distance is at most ~96; the full matrix, one wavefront per overlap, the rest) - no
aligner dependency, no reverse-complemented copies of the reads, no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _on, _ptr, _stream

# Bio.Seq.reverse_complement's table (Bio.Data.IUPACData.ambiguous_dna_complement, both cases); every other byte maps to itself
_PAIRS = {"A": "T", "C": "G", "G": "C", "T": "A", "M": "K", "R": "Y", "W": "W", "S": "S", "Y": "R", "K": "M", "V": "B", "H": "D",
          "D": "H", "B": "V", "X": "X", "N": "N", "U": "A"}
COMPLEMENT = np.arange(256, dtype=np.uint8)
for _a, _b in _PAIRS.items():
    COMPLEMENT[ord(_a)] = ord(_b)
    COMPLEMENT[ord(_a.lower())] = ord(_b.lower())

MAX_OVERLAP = 65536   # query rows one wavefront covers (64 lanes x 32 blocks x 32 rows)


def pack_reads(reads):
    """list of str / bytes -> (uint8[total], int64[R+1]) host tensors."""
    chunks = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(chunks) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in chunks], out=off[1:])
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if chunks else np.zeros(0, dtype=np.uint8)
    return torch.from_numpy(data), torch.from_numpy(off)


def symbol_table(data):
    """uint8[512] (byte -> symbol, byte -> symbol of its complement) and the alphabet size, from the bytes present in the reads
    and their complements."""
    present = torch.unique(data).cpu().numpy().astype(np.int64) if data.numel() else np.zeros(0, dtype=np.int64)
    alphabet = sorted(set(present.tolist()) | set(COMPLEMENT[present].tolist()))
    if len(alphabet) > 32:
        raise ValueError(f"the reads use {len(alphabet)} distinct symbols (with complements); at most 32 are supported")
    index = np.zeros(256, dtype=np.uint8)
    for k, b in enumerate(alphabet):
        index[b] = k
    return torch.from_numpy(np.concatenate([index, index[COMPLEMENT]])), max(len(alphabet), 1)


def edit_distances(reads, src, dst, overlap_length, device=None, with_similarity=True, stats=None):
    """-> (dist int32[E], similarity float32[E] | None) on the device.  reads: list of sequences, or (uint8 data, int64
    offsets) as pack_reads returns them.  stats: a dict that receives {"banded": overlaps settled by the Ukkonen-band pass,
    "edges": E} (the rest went through the full-matrix kernels; both are exact)."""
    lib = _lib.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    data, off = reads if isinstance(reads, tuple) else pack_reads(reads)
    data, off = data.to(device), off.to(device)
    symtab, nsym = symbol_table(data)
    symtab = symtab.to(device)
    src = torch.as_tensor(src).to(device=device, dtype=torch.int32).contiguous()
    dst = torch.as_tensor(dst).to(device=device, dtype=torch.int32).contiguous()
    ol = torch.as_tensor(overlap_length).to(device=device, dtype=torch.int32).contiguous()
    E = int(src.numel())
    if dst.numel() != E or ol.numel() != E:
        raise ValueError("src, dst and overlap_length differ in length")
    if E and (int(torch.maximum(src.max(), dst.max())) >= 2 * (off.numel() - 1) or int(torch.minimum(src.min(), dst.min())) < 0):
        raise IndexError("edge endpoint outside the reads' node range [0, 2R)")
    dist = torch.full((E,), -1, dtype=torch.int32, device=device)
    sim = torch.empty(E, dtype=torch.float32, device=device) if with_similarity else None
    need = ctypes.c_size_t(0)
    _lib.check(lib.gnnome_overlap_workspace_bytes(ctypes.byref(need)), "overlap_workspace_bytes")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=device)
    with _on(device):
        _lib.check(lib.gnnome_overlap_edit_distance(_ptr(data), _ptr(off), off.numel() - 1, _ptr(symtab), nsym, _ptr(src), _ptr(dst), _ptr(ol),
                                                    E, _ptr(dist), _ptr(sim), _ptr(ws), ws.numel(), _stream(device)), "overlap_edit_distance")
    ws.record_stream(torch.cuda.current_stream(device))
    if stats is not None:
        stats.update(banded=int(ws.view(torch.int32)[11]) if E else 0, edges=E)
    if E and int(dist.min()) < 0:
        bad = int((dist < 0).sum())
        raise ValueError(f"{bad} overlaps could not be aligned on the device: longer than {MAX_OVERLAP} bases, or an alphabet of {nsym} "
                         f"symbols whose match masks exceed LDS at that length")
    return dist, sim


def overlap_similarity(reads, src, dst, overlap_length, device=None):
    """float32[E]: 1 - editDistance(src_seq[-ol:], dst_seq[:ol]) / ol, 0.5 where ol == 0 (graph_parser.py:108-113)."""
    return edit_distances(reads, src, dst, overlap_length, device)[1]
