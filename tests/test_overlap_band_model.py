"""The Ukkonen-band recurrence of csrc/overlap_similarity.hip (k_overlap_banded) restated in Python, window logic line for line -
a window of 8 blocks sliding one block per 32 columns, blocks entering as upper bounds, the first block of a column taking
hin = +1, the answer accepted only when it is <= the band's k - and checked against the Wagner-Fischer oracle on the CPU:
whatever the band settles is exact, and it settles everything whose distance is within k.  (The kernel itself is checked
against the oracle on the GPU in tests/test_overlap_similarity.py.)"""
import random

from oracle import overlap_oracle

M32 = 0xFFFFFFFF
W = 8
def myers_block(Pv, Mv, Eq, hin):
    hneg = 1 if hin < 0 else 0
    Xv = Eq | Mv
    Eq |= hneg
    Xh = ((((Eq & Pv) + Pv) & M32) ^ Pv) | Eq
    Ph = (Mv | (~(Xh | Pv) & M32)) & M32
    Mh = Pv & Xh
    hout = (Ph >> 31) - (Mh >> 31)
    Ph = ((Ph << 1) & M32) | (1 if hin > 0 else 0)
    Mh = ((Mh << 1) & M32) | hneg
    Pv = (Mh | (~(Xv | Ph) & M32)) & M32
    Mv = Ph & Xv
    return Pv, Mv, hout

def banded(q, t):
    m, n = len(q), len(t)
    delta = n - m; a = max(delta, 0); bneg = -min(delta, 0)
    k = -1
    for kk in range(96, 15, -16):
        uu = (a + kk + 31) // 32; dn = (31 + kk + bneg) // 32
        if uu + dn <= W - 1:
            k, U = kk, uu; break
    if k < 0: return None
    bmax = (m - 1) // 32
    syms = sorted(set(q + t))
    def eqmask(b, ch):
        x = 0
        for bit in range(32):
            r = 32 * b + bit
            if r < m and q[r] == ch: x |= 1 << bit
        return x
    Pv = [M32] * W; Mv = [0] * W; blk = [None] * W
    score = 0
    for i in range(W):
        b = i - U
        if 0 <= b <= bmax:
            blk[i] = b; score += 32
    chunks = (n + 31) // 32
    for qq in range(chunks):
        fb = qq - U
        if qq > 0:
            Pv = Pv[1:] + [M32]; Mv = Mv[1:] + [0]; blk = blk[1:] + [None]
            b = fb + W - 1
            if 0 <= b <= bmax:
                blk[W - 1] = b; score += 32
        il = min(W - 1, bmax - fb)
        for cc in range(min(32, n - 32 * qq)):
            ch = t[32 * qq + cc]
            h = 1
            for i in range(W):
                if fb + i >= 0 and i <= il:
                    assert blk[i] == fb + i, (blk, fb, i)
                    Pv[i], Mv[i], h = myers_block(Pv[i], Mv[i], eqmask(fb + i, ch), h)
                    if i == il: score += h
    ib = bmax - (chunks - 1 - U)
    if not (0 <= ib < W): return None
    used = m - 32 * bmax
    mask = 0 if used >= 32 else (M32 << used) & M32
    d = score - (bin(Pv[ib] & mask).count("1") - bin(Mv[ib] & mask).count("1"))
    return d if d <= k else None



def band_k(m, n):
    delta = n - m
    a, bneg = max(delta, 0), -min(delta, 0)
    for kk in range(96, 15, -16):
        if (a + kk + 31) // 32 + (31 + kk + bneg) // 32 <= W - 1:
            return kk
    return -1


def test_band_model_is_exact_and_complete_against_the_oracle():
    rng = random.Random(5)
    settled = 0
    for _ in range(160):
        m = rng.choice([1, 2, 5, 31, 32, 33, 64, 65, 100, 257, 300, 900])
        q = "".join(rng.choice("ACGT") for _ in range(m))
        t = list(q)
        for p in rng.sample(range(m), min(rng.randrange(0, 130), m)):
            t[p] = rng.choice([c for c in "ACGT" if c != q[p]])
        for _ in range(rng.choice([0, 0, 1, 4, 30])):        # indels
            p = rng.randrange(len(t) + 1)
            if rng.random() < 0.5 and len(t) > 1:
                del t[min(p, len(t) - 1)]
            else:
                t.insert(p, rng.choice("ACGT"))
        shift = rng.choice([0, 0, 0, 3, -3, 40, -40, 100, -200, 260])
        t = "".join(t) + "".join(rng.choice("ACGT") for _ in range(max(shift, 0)))
        if shift < 0:
            t = t[:shift] or "A"
        want, got, k = overlap_oracle.edit_distance(q, t), banded(q, t), band_k(len(q), len(t))
        if got is not None:
            settled += 1
            assert got == want and got <= k, (m, len(t), want, got, k)
        else:
            assert k < 0 or want > k, (m, len(t), want, k)       # gave up only outside the band
    assert settled > 60
