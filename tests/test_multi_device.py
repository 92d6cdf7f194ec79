"""The in-process multi-device facade ("one host thread + ns_ctx per GPU").

nextsearch::Engine built over a LIST of devices replicates the index on each and cuts a batch into contiguous shards of
ceil(Q / N) queries, one per device, each driven by its own host thread through its own context; the results land in
the caller's one set of host arrays.  The reference has nothing to compare with here (it serialises every search behind
Engine::mtx, src/api_engine.cpp:372): the check is that the sharded answer equals the single-context answer and the
oracle's, byte for byte."""
import ctypes as C

import numpy as np
import pytest

import nsbind
import orc
import workloads


def _bounds(Q, r, n):
    a, b = C.c_uint64(), C.c_uint64()
    nsbind.host_lib().nsh_shard_bounds(Q, r, n, C.byref(a), C.byref(b))
    return a.value, b.value


def test_shard_bounds_are_contiguous_ceil_shards():
    for Q in (0, 1, 2, 7, 100, 101, 16384, 16385):
        for n in (1, 2, 3, 8):
            per = -(-Q // n) if n else Q
            prev = 0
            for r in range(n):
                a, b = _bounds(Q, r, n)
                assert a == min(Q, r * per) and b == min(Q, (r + 1) * per)
                assert a == prev and a <= b
                prev = b
            assert prev == Q


def test_multi_device_engine_without_devices_fails_loudly(index_factory):
    """Host-only (device < 0): the index loads, query preparation works, every search fails with a message — there is no
    CPU scoring path behind the sharded call either."""
    d, _ = index_factory(2, 3000, 2048, 1337, False)
    eng = nsbind.Engine(d, [-1, -1])
    try:
        assert eng.num_devices == 2
        qd, refs, usable = eng.build_refs(["covid virus", "zzzz", "the of"])
        assert list(usable) == [1, 1, 0] and len(refs) > 0
        with pytest.raises(RuntimeError, match="no device context"):
            eng.search_batch(["covid"] * 64, 10)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ctx", [2, 3])
def test_n_contexts_on_one_device_equal_the_single_context_engine(n_ctx, index_factory):
    d, _ = index_factory(3, 20_000, 4096, 77, False)
    one, many, ora = nsbind.Engine(d, 0), nsbind.Engine(d, [0] * n_ctx), orc.Oracle(d)
    try:
        assert many.num_devices == n_ctx
        for Q in (1, n_ctx, 2 * n_ctx, 101, 9000):     # below 2 N queries the batch stays on the primary context; 9000: pipelined shards
            queries = (workloads.cfg5_queries(Q, 11, 4096) + ["zzzz", "the of", "covid"])[:Q] if Q > 3 else workloads.cfg5_queries(Q, 11, 4096)
            for k, flags in ((10, 0), (100, 0), (10, nsbind.NS_FLAG_AND)):
                a, b = one.search_batch(queries, k, flags), many.search_batch(queries, k, flags)
                for x, y in zip(a, b):
                    assert x.tobytes() == y.tobytes(), (Q, k, flags)
            if Q == 101:
                oh, on, of, ou = ora.search_batch(queries, 10)
                gh, gn, gf, gu = many.search_batch(queries, 10)
                assert np.array_equal(gu.astype(bool), ou.astype(bool))
                for q in range(Q):
                    if ou[q]:
                        n = int(on[q])
                        assert int(gn[q]) == n and int(gf[q]) == int(of[q])
                        assert gh[q, :n].tobytes() == oh[q, :n].tobytes()
        # the optional streams reach every replica
        many.build_impacts(); one.build_impacts()
        queries = workloads.cfg5_queries(300, 12, 4096)
        for x, y in zip(one.search_batch(queries, 10), many.search_batch(queries, 10)):
            assert x.tobytes() == y.tobytes()
        # reload keeps the replica set
        many.reload()
        assert many.num_devices == n_ctx
        for x, y in zip(one.search_batch(queries, 10), many.search_batch(queries, 10)):
            assert x.tobytes() == y.tobytes()
    finally:
        one.close(); many.close(); ora.close()
