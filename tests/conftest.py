import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nextsearch-api_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# The forced kernel variants (ns_set_tuning variant != 0) live in libnextsearch_hip_variants.so, not in the product library.
# A process that set NS_HIP_LIB to that build runs the variant cases; every other process skips them, and ONE test of the
# main suite (test_gpu_parity.py::test_forced_kernel_variants_in_the_variants_build) runs them in such a process.
VARIANTS_LIB = os.path.join(PKG, "libnextsearch_hip_variants.so")
VARIANTS_BUILD = "variants" in os.path.basename(os.environ.get("NS_HIP_LIB", ""))


def need_variants(variant):
    if variant != 0 and not VARIANTS_BUILD:
        pytest.skip("forced kernel variants run in the variants build (see test_forced_kernel_variants_in_the_variants_build)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(PKG, "libnextsearch_hip.so"), os.path.join(PKG, "libnextsearch_host.so"), VARIANTS_LIB]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-s", "-C", PKG, "all"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libbm25_oracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_ensure_built()

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# hit-list fixtures (found + per hit: segment, docId, score bits); meta1.json holds JSON TEXT of the reference instead
GOLDEN_NAMES = sorted(f[:-5] for f in os.listdir(GOLDEN_DIR) if f.endswith(".json") and f not in ("meta1.json", "invert1.json", "sem1.json", "segwriter1.json", "cache1.json", "semload1.json", "fullsize.json"))


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        return json.load(f)


def sha256_tree(root):
    out = {}
    for d, _, files in sorted(os.walk(root)):
        for fn in sorted(files):
            p = os.path.join(d, fn)
            with open(p, "rb") as f:
                out[os.path.relpath(p, root)] = hashlib.sha256(f.read()).hexdigest()
    return out


@pytest.fixture(scope="session")
def index_factory(tmp_path_factory):
    """Generates (once per session) an index with the repo's deterministic generator."""
    import nsbind

    cache = {}

    def make(n_segments, docs_per_segment, vocab=65536, seed=1337, legacy=False):
        key = (n_segments, docs_per_segment, vocab, seed, legacy)
        if key not in cache:
            d = str(tmp_path_factory.mktemp("idx") / "index")
            total = nsbind.gen_index(d, n_segments, docs_per_segment, vocab, seed, legacy)
            cache[key] = (d, total)
        return cache[key]

    return make


@pytest.fixture(scope="session")
def golden_index(index_factory):
    def make(name):
        g = load_golden(name)
        p = g["params"]
        d, total = index_factory(p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
        return g, d, total

    return make


@pytest.fixture(scope="session")
def gpu_available():
    import ctypes as C

    import nsbind

    h = C.c_void_p()
    rc = nsbind.hip_lib().ns_ctx_create(0, C.byref(h))
    if rc == 0:
        nsbind.hip_lib().ns_ctx_destroy(h)
        return True
    return False
