"""overlap_similarity (graph_parser.py:101-117): the oracle against the reference's own values (golden
G10), and the device kernel gnnome_overlap_edit_distance against the oracle - bit-exact integer distances."""
import os
import random

import pytest
import torch

from conftest import GOLDEN, load_golden
from gnnome_amd import gfa
from oracle import overlap_oracle


def _py_edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _forward_reads(g):
    return [g["read_seqs"][2 * r] for r in range(g["num_nodes"] // 2)]


def test_oracle_edit_distance_is_the_wagner_fischer_value():
    rng = random.Random(0)
    for _ in range(60):
        a = "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 70)))
        b = "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 70)))
        assert overlap_oracle.edit_distance(a, b) == _py_edit_distance(a, b)
    assert overlap_oracle.edit_distance("", "") == 0 and overlap_oracle.edit_distance("ACGT", "") == 4


def test_oracle_matches_the_reference_parsers_similarities_golden_g10():
    """The values calculate_similarities itself produced (make_golden_gfa.py: the reference's parser text, run as is)."""
    for c in load_golden("g10_gfa.pt")["cases"]:
        g = gfa.read_gfa(os.path.join(GOLDEN, c["gfa"]), similarity=None, keep_sequences=True)
        if not g["read_seqs"] or any(s == "*" for s in g["read_seqs"].values()):
            continue
        _, sims = overlap_oracle.calculate_similarities(_forward_reads(g), g["src"], g["dst"], g["overlap_length"])
        assert torch.allclose(torch.tensor(sims, dtype=torch.float64), c["overlap_similarity"].double(), atol=1e-7), c["name"]
        # the reverse-complement strand is the reference's: read_seqs[2r+1] of this package's reader and of the oracle agree
        assert overlap_oracle.read_seqs(_forward_reads(g)) == g["read_seqs"]


def dev():
    return torch.device("cuda", 0)


def _mutate(rng, s, rate, alphabet="ACGT"):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue                      # deletion
        if r < 2 * rate / 3:
            out.append(rng.choice(alphabet))   # substitution
            continue
        if r < rate:
            out.append(rng.choice(alphabet))   # insertion
        out.append(ch)
    return "".join(out)


def _overlapping_reads(rng, lengths, rate, alphabet="ACGT"):
    """reads laid out on one genome so that consecutive reads truly overlap (distance ~ rate * overlap), plus noise."""
    genome = "".join(rng.choice(alphabet) for _ in range(sum(lengths)))
    reads, pos = [], 0
    for ln in lengths:
        reads.append(_mutate(rng, genome[pos:pos + ln], rate, alphabet))
        pos += max(1, ln // 3)
    return reads


@pytest.mark.gpu
def test_device_edit_distance_equals_the_oracle_on_every_orientation_and_block_boundary():
    from gnnome_amd import overlap
    rng = random.Random(1)
    lengths = [31, 32, 33, 64, 65, 100, 500, 2047, 2048, 2049, 2100, 4095, 4097, 6200, 7000, 9000, 12_289, 40, 1, 3000]
    reads = _overlapping_reads(rng, lengths, 0.05)
    R = len(reads)
    src, dst, ol = [], [], []
    for r in range(R - 1):
        for su in (0, 1):
            for sv in (0, 1):
                src.append(2 * r + su)
                dst.append(2 * (r + 1) + sv)
                ol.append(rng.randrange(1, min(len(reads[r]), len(reads[r + 1])) + 1))
    # overlaps longer than one or both reads (read_src[-ol:] / read_dst[:ol] are then the whole reads), zero-length, exact
    # block boundaries, a read against itself and against its own reverse complement
    extra = [(0, 2, 10_000), (5, 12, 100_000), (26, 28, 2048), (27, 29, 2047), (18, 21, 4096), (6, 6, 64), (6, 7, 64), (36, 2, 0), (36, 37, 1)]
    for u, v, L in extra:
        src.append(u), dst.append(v), ol.append(L)
    want_d, want_s = overlap_oracle.calculate_similarities(reads, src, dst, ol)
    d, s = overlap.edit_distances(reads, src, dst, ol, device=dev())
    assert d.cpu().tolist() == want_d
    assert torch.allclose(s.cpu().double(), torch.tensor(want_s, dtype=torch.float64), atol=1e-7)
    assert (s.cpu()[torch.tensor(ol) == 0] == 0.5).all()


@pytest.mark.gpu
def test_device_edit_distance_long_overlaps_and_wide_alphabets():
    """Every blocks-per-lane class up to 32 (65 536 query rows), ambiguity codes and lower case through the complement table."""
    from gnnome_amd import overlap
    rng = random.Random(2)
    alphabet = "ACGTNacgtnRYKMSW"
    big = "".join(rng.choice("ACGT") for _ in range(66_000))
    reads = [big[:20_000], _mutate(rng, big[:20_000], 0.02), big[:40_000], _mutate(rng, big[30_000:40_000], 0.1),
             big[:60_000], _mutate(rng, big[58_000:60_000], 0.03), big,
             "".join(rng.choice(alphabet) for _ in range(3000)), "".join(rng.choice(alphabet) for _ in range(2500))]
    cases = [(0, 2, 20_000), (3, 1, 20_000),          # 20k x 20k, both strands                         (class 12)
             (4, 6, 10_000), (7, 5, 10_000),           # 40k-base read's suffix against a 10k read        (class 6)
             (4, 6, 40_000),                           # query 40 000 rows against the whole 10k read      (class 24)
             (8, 10, 60_000), (11, 9, 2000),           # query 60 000 rows against a 2k read               (class 32)
             (14, 16, 3000), (17, 15, 2500), (15, 17, 2999), (14, 15, 3000)]   # 16-symbol alphabet, rc through the IUPAC table
    src, dst, ol = (list(t) for t in zip(*cases))
    want_d, want_s = overlap_oracle.calculate_similarities(reads, src, dst, ol)
    d, s = overlap.edit_distances(reads, src, dst, ol, device=dev())
    assert d.cpu().tolist() == want_d
    assert torch.allclose(s.cpu().double(), torch.tensor(want_s, dtype=torch.float64), atol=1e-7)
    with pytest.raises(ValueError):                    # 66 000 query rows: refused, not guessed
        overlap.edit_distances(reads, [12], [0], [66_000], device=dev())


@pytest.mark.gpu
def test_gfa_reader_computes_similarities_on_the_device_golden_g10():
    """read_gfa(path) with no tags, no callable and no edlib returns the reference parser's similarities."""
    seen = 0
    for c in load_golden("g10_gfa.pt")["cases"]:
        path = os.path.join(GOLDEN, c["gfa"])
        with open(path) as f:
            if any(line.startswith("S") and line.split()[2] == "*" for line in f):
                continue
        g = gfa.read_gfa(path)
        assert g["overlap_similarity"] is not None and g["overlap_similarity"].dtype == torch.float32
        assert torch.allclose(g["overlap_similarity"].double(), c["overlap_similarity"].double(), atol=1e-7), c["name"]
        seen += 1
    assert seen >= 1


@pytest.mark.gpu
def test_many_overlaps_of_assembly_shape_against_the_oracle():
    """A few thousand overlaps of HiFi-like shape (0.5 % divergence, lengths 1-6 kb) - all classes' ticket loops interleaved."""
    from gnnome_amd import overlap
    rng = random.Random(3)
    lengths = [rng.randrange(1500, 6500) for _ in range(150)]
    reads = _overlapping_reads(rng, lengths, 0.005)
    src, dst, ol = [], [], []
    for r in range(len(reads) - 3):
        for t in (1, 2, 3):
            su, sv = rng.randrange(2), rng.randrange(2)
            src.append(2 * r + su), dst.append(2 * (r + t) + sv)
            ol.append(rng.randrange(500, min(len(reads[r]), len(reads[r + t]))))
    want_d, _ = overlap_oracle.calculate_similarities(reads, src, dst, ol)
    d, _ = overlap.edit_distances(reads, src, dst, ol, device=dev())
    assert d.cpu().tolist() == want_d


def _with_edits(rng, s, subs, dels, ins):
    """s with exactly `subs` substitutions, `dels` deletions and `ins` insertions at distinct, spread-out positions."""
    pos = sorted(rng.sample(range(5, len(s) - 5), subs + dels + ins))
    rng.shuffle(kinds := ["s"] * subs + ["d"] * dels + ["i"] * ins)
    out, at = [], 0
    for p, kd in zip(pos, kinds):
        out.append(s[at:p])
        if kd == "s":
            out.append(rng.choice([c for c in "ACGT" if c != s[p]]))
        elif kd == "i":
            out.append(rng.choice("ACGT") + s[p])
        at = p + 1
    out.append(s[at:])
    return "".join(out)


@pytest.mark.gpu
def test_banded_pass_is_exact_inside_its_band_and_hands_over_outside():
    """Round 4: the Ukkonen-band kernel (one thread per overlap, a window of 256 query rows) in front of the full-matrix wave
    kernel.  Around the band's k = 96: distances 80 .. 110 at equal lengths, length differences up to +-250 (k shrinks, then the
    band is not used at all), all four orientations, queries shorter than a block, alphabets of 4 / 5 / 16 / 22 symbols -
    default path, full-matrix-only path (tuning key 9) and the Wagner-Fischer oracle give the same integers; the statistics
    say which kernel settled how many."""
    from gnnome_amd import ops, overlap
    rng = random.Random(11)
    base = "".join(rng.choice("ACGT") for _ in range(4000))
    reads, src, dst, ol = [], [], [], []

    def pair(a, b, L=None):   # read 2i ends with a, read 2i+1 starts with b
        reads.extend([base[:100] + a, b + base[200:300]])
        r = len(reads) // 2 - 1
        su, sv = rng.randrange(2), rng.randrange(2)
        # a strand flip is expressed on the node id: node 2r+1 reads read r reverse-complemented, so store the reverse complement
        if su:
            reads[2 * r] = overlap_oracle.read_seqs([reads[2 * r]])[1]
        if sv:
            reads[2 * r + 1] = overlap_oracle.read_seqs([reads[2 * r + 1]])[1]
        src.append(2 * (2 * r) + su), dst.append(2 * (2 * r + 1) + sv), ol.append(len(a) if L is None else L)
    for d in list(range(80, 111, 3)) + [0, 1, 95, 96, 97]:
        pair(base[:3000], _with_edits(rng, base[:3000], d, 0, 0))                          # equal lengths, d substitutions
    for dl, ins in [(10, 0), (0, 10), (40, 5), (5, 40), (90, 0), (0, 90), (150, 0), (0, 150), (250, 3), (3, 250)]:
        b = _with_edits(rng, base[:3000], 6, dl, ins)                                      # lengths differ by ins - dl
        reads.extend([base[:3000], b])                                                     # ol beyond both reads: the whole reads, m != n
        r = len(reads) // 2 - 1
        src.append(2 * (2 * r)), dst.append(2 * (2 * r + 1)), ol.append(5000)
    for ln in (1, 5, 31, 32, 33, 63, 64, 65, 200, 255, 256, 257):
        pair(base[:ln], _with_edits(rng, base[:ln], min(2, max(ln - 12, 0)), 0, 0) if ln > 12 else base[:ln])
    want_d, _ = overlap_oracle.calculate_similarities(reads, src, dst, ol)
    st = {}
    d, _ = overlap.edit_distances(reads, src, dst, ol, device=dev(), stats=st)
    assert d.cpu().tolist() == want_d
    inside = sum(1 for w, s_, t_, L in zip(want_d, src, dst, ol) if w <= 16 and L > 0)
    assert st["edges"] == len(src) and inside <= st["banded"] < len(src)     # the far-off pairs were NOT settled by the band
    try:
        ops.set_tuning(9, 1)
        st1 = {}
        d1, _ = overlap.edit_distances(reads, src, dst, ol, device=dev(), stats=st1)
    finally:
        ops.set_tuning(9, 0)
    assert st1["banded"] == 0 and d1.cpu().tolist() == want_d
    # alphabets: 5 symbols (3 planes), 16 (4 planes), 22 (no band: full matrix only)
    for alphabet, banded in (("ACGTN", True), ("ACGTNacgtnRYKMSW", True), ("ACGTNacgtnRYKMSWBDHVbd", False)):
        g = "".join(rng.choice(alphabet) for _ in range(2500))
        rd = [g, _mutate(rng, g, 0.01, alphabet), g[:700], _mutate(rng, g[:700], 0.3, alphabet)]
        s2, d2, o2 = [0, 1, 3, 4, 2], [2, 3, 0, 6, 1], [2400, 2500, 2000, 700, 1500]
        w2, _ = overlap_oracle.calculate_similarities(rd, s2, d2, o2)
        st2 = {}
        got, _ = overlap.edit_distances(rd, s2, d2, o2, device=dev(), stats=st2)
        assert got.cpu().tolist() == w2, alphabet
        assert (st2["banded"] > 0) == banded, (alphabet, st2)


@pytest.mark.gpu
def test_endpoints_out_of_range_are_reported_by_the_kernel_itself():
    """a caller of the C ABI (not of the Python wrapper, which checks) may hand over node ids outside [0, 2R): the
    kernels must not index the offsets with them - the edge comes back as -1, its neighbours in the list are aligned as usual."""
    import ctypes

    from gnnome_amd import _lib, overlap
    from gnnome_amd.ops import _ptr, _stream
    rng = random.Random(5)
    reads = _overlapping_reads(rng, [300, 400, 500, 350], 0.02)
    src, dst, ol = [0, 2, 9, 4, -1, 1], [2, 4, 2, 800, 3, 3], [100, 150, 50, 60, 70, 80]
    data, off = overlap.pack_reads(reads)
    symtab, nsym = overlap.symbol_table(data)
    d = dev()
    t = lambda x, dt: torch.as_tensor(x, dtype=dt).to(d)  # noqa: E731
    data, off, symtab = data.to(d), off.to(d), symtab.to(d)
    s_, d_, o_ = t(src, torch.int32), t(dst, torch.int32), t(ol, torch.int32)
    dist = torch.full((len(src),), -7, dtype=torch.int32, device=d)
    lib = _lib.load()
    need = ctypes.c_size_t(0)
    _lib.check(lib.gnnome_overlap_workspace_bytes(ctypes.byref(need)), "ws")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=d)
    _lib.check(lib.gnnome_overlap_edit_distance(_ptr(data), _ptr(off), len(reads), _ptr(symtab), nsym, _ptr(s_), _ptr(d_), _ptr(o_), len(src),
                                                _ptr(dist), None, _ptr(ws), ws.numel(), _stream(d)), "overlap_edit_distance")
    good = [0, 1, 5]
    want, _ = overlap_oracle.calculate_similarities(reads, [src[i] for i in good], [dst[i] for i in good], [ol[i] for i in good])
    got = dist.cpu().tolist()
    assert [got[i] for i in good] == want and [got[i] for i in (2, 3, 4)] == [-1, -1, -1]


@pytest.mark.gpu
def test_symbol_table_entries_beyond_the_alphabet_read_as_its_last_symbol_in_every_kernel():
    """include/gnnome_hip.h: entries >= num_symbols are read as num_symbols - 1.  The banded pass packs symbols into 4-bit nibbles and used to
    copy the table raw: an entry > 15 corrupted the neighbouring columns.  Here T (the last of A, C, G, T) is entered as 77."""
    import ctypes

    from gnnome_amd import _lib, overlap
    from gnnome_amd.ops import _ptr, _stream, set_tuning
    rng = random.Random(8)
    reads = _overlapping_reads(rng, [600, 700, 650, 800], 0.01)
    src, dst, ol = [0, 2, 4, 1, 3], [2, 4, 6, 3, 5], [300, 350, 320, 280, 310]
    data, off = overlap.pack_reads(reads)
    symtab, nsym = overlap.symbol_table(data)
    assert nsym == 4 and int(symtab[ord("T")]) == 3
    odd = symtab.clone()
    odd[ord("T")] = 77
    odd[256 + ord("A")] = 77       # (the complement of A)
    d = dev()
    t = lambda x, dt: torch.as_tensor(x, dtype=dt).to(d)  # noqa: E731
    data, off = data.to(d), off.to(d)
    s_, d_, o_ = t(src, torch.int32), t(dst, torch.int32), t(ol, torch.int32)
    lib = _lib.load()
    need = ctypes.c_size_t(0)
    _lib.check(lib.gnnome_overlap_workspace_bytes(ctypes.byref(need)), "ws")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=d)
    got = {}
    for band in (0, 1):       # 0: banded pass first, 1: full-matrix kernels only
        set_tuning(9, band)
        for name, tab in (("plain", symtab), ("odd", odd)):
            dist = torch.full((len(src),), -7, dtype=torch.int32, device=d)
            _lib.check(lib.gnnome_overlap_edit_distance(_ptr(data), _ptr(off), len(reads), _ptr(tab.to(d)), nsym, _ptr(s_), _ptr(d_), _ptr(o_), len(src),
                                                        _ptr(dist), None, _ptr(ws), ws.numel(), _stream(d)), "overlap_edit_distance")
            got[(band, name)] = dist.cpu().tolist()
    set_tuning(9, 0)
    want, _ = overlap_oracle.calculate_similarities(reads, src, dst, ol)
    assert all(v == want for v in got.values()), got
