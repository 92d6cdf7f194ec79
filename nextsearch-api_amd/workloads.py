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


# ---- synthetic metadata.csv (CORD-19 column layout) for the result-decoration tests -----------------
_META_HEADER = ("cord_uid,sha,source_x,title,doi,pmcid,pubmed_id,license,abstract,publish_time,authors,journal,"
                "mag_id,who_covidence_id,arxiv_id,pdf_json_files,pmc_json_files,url,s2_id")
_TITLES = ["Clinical features of patients", "A study, with commas, of viral load", 'The "quoted" outbreak', "",
           "Café au lait spots — a review", "Tab\there and back\\slash", "Short", "  padded title  "]
_AUTHORS = ["Smith, John; Doe, Jane", "John Smith", "(Zhang Wei) 张伟; Li, Na", " ; Doe, Jane", "Smith,", "",
            "Madonna", "van der Berg, Anna;", "  García López, María  ; X, Y", "O'Neil, Pat"]
_URLS = ["https://doi.org/10.1000/{i}; https://www.ncbi.nlm.nih.gov/pubmed/{i}/", "https://example.org/paper/{i}", "",
         ";https://only-after-semicolon/{i}"]
_DATES = ["2020-03-{d:02d}", "2020", "", "2019-12-31"]


def _csv_field(s):
    """Quote the way CORD-19's writer does: only when needed; embedded quotes are doubled."""
    if any(c in s for c in ',"\n') or s != s.strip():
        return '"' + s.replace('"', '""') + '"'
    return s


def metadata_csv(n_docs_total, seed=11):
    """Deterministic CSV text (bytes) for documents u00000000 .. u{n_docs_total-1:08d}: ~70 % of the docs get a
    row; plus duplicate cord_uids (first row wins), rows for unknown uids, a short (malformed) row, a row with an
    empty cord_uid, a quoted field with an embedded newline (split by the line-based reader) and CRLF endings."""
    rng = random.Random(seed)
    lines = [_META_HEADER]
    for i in range(n_docs_total):
        if rng.random() < 0.3:
            continue
        t = _TITLES[rng.randrange(len(_TITLES))]
        if t and rng.random() < 0.5:
            t = f"{t} {i}"
        row = [f"u{i:08d}", f"{rng.getrandbits(64):016x}", "PMC", t, f"10.1000/{i}", f"PMC{i}", str(i), "cc-by",
               "Abstract text, with a comma." if rng.random() < 0.5 else "", _DATES[rng.randrange(len(_DATES))].format(d=1 + i % 28),
               _AUTHORS[rng.randrange(len(_AUTHORS))], "Journal of Tests", "", "", "", f"document_parses/pdf_json/{i}.json", "",
               _URLS[rng.randrange(len(_URLS))].format(i=i), str(1000 + i)]
        line = ",".join(_csv_field(x) for x in row)
        if rng.random() < 0.05:
            line += "\r"   # CRLF ending: the \r stays in the last column
        lines.append(line)
        if rng.random() < 0.03:   # a later duplicate with other values: must be ignored
            dup = list(row)
            dup[3] = "DUPLICATE ROW " + str(i)
            dup[10] = "Nobody, N"
            lines.append(",".join(_csv_field(x) for x in dup))
        if rng.random() < 0.02:
            lines.append(f"zz{i:08d},,,Unknown document {i},,,,,,2021,Ghost G,,,,,,,http://nowhere/{i},")
        if rng.random() < 0.01:
            lines.append("")                       # short row (no cord_uid column content)
            lines.append(",deadbeef,PMC,Row with an empty cord_uid,,,,,,2020,A B,,,,,,,http://x,")
        if rng.random() < 0.01:
            lines.append(f'yy{i:08d},,,"A title with an embedded\nnewline, inside quotes",,,,,,2020,"C, D",,,,,,,http://y/{i},')
    return ("\n".join(lines) + "\n").encode("utf-8")


def embeddings_text(vocab, dim=24, seed=5, clusters=0):
    """A deterministic word-embedding text file for the synthetic vocabulary (term_name(1..vocab)), in the format
    src/semantic_embedding.cpp:36-100 reads: optional "<count> <dim>" header, then "word v1 .. vD" lines.
    Terms come in clusters (centre + noise) so that neighbours above the reference's min_sim 0.55 exist; the file
    also holds words no lexicon has, a line with fewer than 10 values and a line of another dimension (all skipped
    by the loader), and a repeated word (its second row is unreachable by name but still a neighbour)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    clusters = clusters or max(8, vocab // 6)
    centres = rng.standard_normal((clusters, dim))
    lines = ["%d %d" % (vocab + 40, dim)]
    order = rng.permutation(vocab) + 1
    for n, r in enumerate(order):
        v = centres[int(r) % clusters] + rng.standard_normal(dim) * (0.25 + 0.5 * ((int(r) // clusters) % 3))
        lines.append(term_name(int(r)) + " " + " ".join("%.6f" % x for x in v))
        if n % 97 == 5:
            lines.append("zz_notindexed_%d " % n + " ".join("%.6f" % x for x in rng.standard_normal(dim)))
        if n == 11:
            lines.append(term_name(int(order[3])) + " " + " ".join("%.6f" % x for x in rng.standard_normal(dim)))   # repeated word
            lines.append(term_name(int(order[4])) + "x 0.1 0.2 0.3")                                                    # < 10 values
            lines.append(term_name(int(order[5])) + "y " + " ".join("%.6f" % x for x in rng.standard_normal(dim + 3)))  # other dim
            lines.append("")
    return ("\n".join(lines) + "\n").encode("utf-8")
