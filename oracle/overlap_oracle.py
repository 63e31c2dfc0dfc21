"""TEST INFRASTRUCTURE ONLY: CPU restatement of graph_parser.py:101-117 (calculate_similarities) and of the read_seqs it is fed
(:365: `str(seq if node_id % 2 == 0 else seq.reverse_complement())`), with oracle/overlap_oracle.c as the edit distance that
the third-party aligner edlib (1.3.9, not under /root/reference) returns.  Imported by tests/ only."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# Bio.Seq.reverse_complement (biopython 1.79, Bio.Data.IUPACData.ambiguous_dna_complement), both cases
_COMP = str.maketrans("ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna")


def _lib():
    path = os.path.join(_HERE, "_build", "liboverlap_oracle.so")
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-C", _HERE])
    lib = ctypes.CDLL(path)
    lib.gnnome_oracle_edit_distance.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    return lib


def edit_distance(a, b):
    a, b = (s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in (a, b))
    out = ctypes.c_int64(0)
    if _lib().gnnome_oracle_edit_distance(a, len(a), b, len(b), ctypes.byref(out)) != 0:
        raise MemoryError("gnnome_oracle_edit_distance")
    return int(out.value)


def read_seqs(reads):
    """graph_parser.py:174-181, :365: node 2r = read r, node 2r+1 = its reverse complement."""
    seqs = {}
    for r, s in enumerate(reads):
        seqs[2 * r] = s
        seqs[2 * r + 1] = s.translate(_COMP)[::-1]
    return seqs


def calculate_similarities(reads, src, dst, overlap_lengths):
    """graph_parser.py:101-117 -> (edit distances, similarities) per edge, Python floats like the reference."""
    seqs = read_seqs(reads)
    dists, sims = [], []
    for u, v, ol in zip(src, dst, overlap_lengths):
        u, v, ol = int(u), int(v), int(ol)
        if ol > 0:
            d = edit_distance(seqs[u][-ol:], seqs[v][:ol])
            dists.append(d)
            sims.append(1 - d / ol)
        else:
            dists.append(0)
            sims.append(0.5)
    return dists, sims
