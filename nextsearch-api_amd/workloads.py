"""Query workloads of BASELINE.json's configs over the synthetic CORD-19-shaped index.

Queries are produced as TEXT (the reference's entry point is Engine::search(const std::string&, int),
src/api_engine.cpp:369), so tokenisation, stop-word filtering and lexicon probing are exercised too.
Deterministic: Python's Mersenne-Twister `random.Random(seed)` only.
"""
import bisect
import math
import random

WORDS = ["covid", "virus", "vaccine", "infection", "patients", "respiratory", "coronavirus", "pandemic"]


def term_name(rank):
    """Same naming as the index generator (host/gen_index.hpp: term_name)."""
    if 1 <= rank <= 8:
        return WORDS[rank - 1]
    return "t%06d" % rank


class _Zipf:
    def __init__(self, lo, hi):
        self.lo = lo
        acc, c = [], 0.0
        for r in range(lo, hi + 1):
            c += 1.0 / r
            acc.append(c)
        self.cum, self.total = acc, c

    def draw(self, rng):
        return self.lo + bisect.bisect_left(self.cum, rng.random() * self.total)


def _poisson2(rng):
    # inverse CDF of Poisson(lambda = 2)
    u, k, p = rng.random(), 0, math.exp(-2.0)
    c = p
    while u > c and k < 64:
        k += 1
        p *= 2.0 / k
        c += p
    return k


def cfg2_queries(n=1024, seed=2002, vocab=65536):
    """2 distinct terms per query, ranks uniform in [10, 1000] (conjunctive extension, K=10)."""
    rng = random.Random(seed)
    hi = min(1000, vocab)
    out = []
    for _ in range(n):
        a = rng.randint(10, hi)
        b = rng.randint(10, hi)
        while b == a:
            b = rng.randint(10, hi)
        out.append(f"{term_name(a)} {term_name(b)}")
    return out


def cfg3_queries(n=4096, seed=2003, vocab=65536, nterms=5, max_rank=5000):
    """5 distinct terms per query, rank ~ 1/r over [1, 5000] (disjunctive, K=100)."""
    rng = random.Random(seed)
    z = _Zipf(1, min(max_rank, vocab))
    out = []
    for _ in range(n):
        ranks = []
        while len(ranks) < nterms:
            r = z.draw(rng)
            if r not in ranks:
                ranks.append(r)
        out.append(" ".join(term_name(r) for r in ranks))
    return out


def cfg4_queries(n=4096, seed=2004, vocab=65536):
    """cfg3's law, used over 8 x 125k-doc segments with K=10."""
    return cfg3_queries(n, seed, vocab)


def cfg5_queries(n=16384, seed=2005, vocab=65536):
    """1..8 terms (1 + Poisson(2), clamped); each term hot (rank U[1,32]) with p = 0.3, else
    log-uniform over [33, vocab] (Zipf-skewed hot + long-tail mix, K=10)."""
    rng = random.Random(seed)
    out = []
    lo_tail = 33
    ratio = vocab / lo_tail
    for _ in range(n):
        nt = max(1, min(8, 1 + _poisson2(rng)))
        ranks = []
        for _ in range(nt):
            if rng.random() < 0.3 or vocab <= lo_tail:
                ranks.append(rng.randint(1, min(32, vocab)))
            else:
                r = int(lo_tail * (ratio ** rng.random()))
                ranks.append(max(lo_tail, min(vocab, r)))
        out.append(" ".join(term_name(r) for r in ranks))
    return out


WORKLOADS = {
    # name: (generator, default Q, K, flags(0=OR,1=AND), index shape (n_segments, docs_per_segment))
    "cfg2": (cfg2_queries, 1024, 10, 1, (1, 100_000)),
    "cfg3": (cfg3_queries, 4096, 100, 0, (1, 1_000_000)),
    "cfg4": (cfg4_queries, 4096, 10, 0, (8, 125_000)),
    "cfg5": (cfg5_queries, 16384, 10, 0, (1, 1_000_000)),
}
