"""Adversarial inputs of the chaining stage (test infrastructure): SMEM + hit sets made to stress what real reads rarely do --
many chains at EQUAL positions (the reference's B-tree order of equal keys), hundreds of chains per read (multi-level tree, node
splits), long chains, repeat SMEMs beyond max_occ (hit sub-sampling, frac_rep), seeds across contig and strand boundaries."""
import numpy as np

from oracle_py import MEM_TL_DTYPE


def make_read(rng, read_len=150, l_pac=200_000, n_smems=None, n_sites=None, dup_bias=0.5, max_hits=40, big=False):
    """Returns (smems, hits).  Hits are drawn from a small set of `sites` shifted by the SMEM's query offset (collinear -> merges)
    or not shifted / shifted wrongly (same position, different diagonal -> a new chain on an equal key)."""
    n_smems = n_smems or int(rng.integers(1, 24))
    n_sites = n_sites or int(rng.integers(1, 30))
    sites = rng.integers(1000, 2 * l_pac - 1000, size=n_sites)
    smems = np.zeros(n_smems, MEM_TL_DTYPE)
    hits = []
    seen = set()
    for i in range(n_smems):
        while True:                                        # (start, end) unique: records with equal keys describe the same substring in real
            s = int(rng.integers(0, read_len - 19))        # data, hence carry the same hits -- their order (an unstable sort's) cannot matter
            e = int(min(read_len, s + rng.integers(19, 80)))
            if (s, e) not in seen:
                seen.add((s, e))
                break
        k = int(rng.integers(1, max_hits + 1))
        if big and rng.random() < 0.2:
            k = int(rng.integers(501, 1400))              # beyond max_occ: sub-sampled, counts into frac_rep
        pos = []
        for _ in range(k):
            site = int(sites[rng.integers(0, n_sites)])
            mode = rng.random()
            if mode < dup_bias:
                p = site                                   # the bare site: equal positions across SMEMs with different query offsets
            elif mode < dup_bias + 0.3:
                p = site + s                               # collinear with the other SMEMs of this site
            else:
                p = site + s + int(rng.integers(-120, 120))   # near the diagonal: inside or outside the band
            pos.append(min(max(p, 0), 2 * l_pac - 1))
        smems[i]["start"], smems[i]["end"], smems[i]["hitbeg"], smems[i]["hitcount"] = s, e, len(hits), k
        hits += pos
    return smems, np.array(hits, np.uint64)


def workload(seed, n_reads, **kw):
    rng = np.random.default_rng(seed)
    out = []
    for r in range(n_reads):
        big = r % 7 == 3
        kind = r % 5
        if kind == 0:
            out.append(make_read(rng, n_sites=int(rng.integers(1, 4)), dup_bias=0.8, big=big, **kw))          # few sites, mostly equal keys
        elif kind == 1:
            out.append(make_read(rng, n_smems=int(rng.integers(10, 40)), n_sites=int(rng.integers(40, 200)), dup_bias=0.3, max_hits=60, big=big, **kw))
        elif kind == 2:
            out.append(make_read(rng, read_len=250, dup_bias=0.5, big=big, **kw))
        else:
            out.append(make_read(rng, big=big, **kw))
    return out
