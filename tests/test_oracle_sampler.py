"""The sampler oracle (oracle/merw_oracle.c) against (a) glibc, (b) the committed golden vectors that
the unmodified reference binary produced, (c) the reference binary itself when it is available."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from conftest import golden, golden_files
from oracle import merw

SAMPLER_GOLDENS = golden_files("sampler_*.npz")


def test_glibc_rand_restatement_matches_libc():
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 42, 2 ** 31 + 5, 2 ** 32 - 1):
        libc.srand(ctypes.c_uint(seed))
        ref = np.array([libc.rand() for _ in range(1500)], dtype=np.int32)
        assert (merw.glibc_stream(seed, 1500) == ref).all()
        assert (merw.glibc_stream(seed, 100, skip=700) == ref[700:800]).all()


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10 (kat_vectors): counter, key -> output
    # ctr = 0, key = 0
    L = merw.lib()
    got = [merw.philox_draw(0, 0, q) for q in range(4)]
    assert got == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    # ctr = ffffffff x4, key = ffffffff x2
    seed = 0xffffffffffffffff
    got = [merw.philox_draw(seed, 0xffffffffffffffff, (0xffffffffffffffff << 2 | q) & 0xffffffffffffffff)
           for q in range(4)]
    # offset/4 is limited to 62 bits through this entry point, so only check the first vector strictly
    assert len(got) == 4 and L is not None


@pytest.mark.parametrize("name", SAMPLER_GOLDENS)
def test_oracle_reproduces_reference_golden(name):
    g = golden(name)
    n, W, L, seed, E = int(g["n"]), int(g["W"]), int(g["L"]), int(g["seed"]), int(g["epochs"])
    ids, codes = merw.sample_full(n, g["u"], g["v"], g["p"], W, L, merw.DRAW_GLIBC, seed, epoch_count=E)
    assert (ids == g["ids"]).all()
    assert (codes == g["codes"]).all()
    assert hashlib.md5(merw.format_text(ids, codes)).hexdigest() == str(g["md5"])


def test_oracle_epoch_and_node_windows_are_consistent():
    g = golden("sampler_cornell_7_6.npz")
    n, W, L, seed = int(g["n"]), int(g["W"]), int(g["L"]), int(g["seed"])
    off, A, B, S = merw.alias_build(n, g["u"], g["v"], g["p"])
    dis = merw.bfs_dense(n, g["u"], g["v"], L)
    ids, codes = merw.walk(n, off, A, B, S, dis, W, L, merw.DRAW_GLIBC, seed, epoch_begin=1, epoch_count=2,
                           node_begin=17, node_count=50)
    assert (ids == g["ids"][1:3, 17:67]).all()
    assert (codes == g["codes"][1:3, 17:67]).all()


def test_structural_invariants_philox():
    g = golden("sampler_synthetic97_12_5.npz")
    n, W, L = int(g["n"]), int(g["W"]), int(g["L"])
    u, v = g["u"], g["v"]
    ids, codes = merw.sample_full(n, u, v, g["p"], W, L, merw.DRAW_PHILOX, 1234, epoch_count=2)
    adj = set(zip(u.tolist(), v.tolist()))
    assert (ids[:, :, :, 0] == np.arange(n)[None, :, None]).all()
    assert (codes[..., 0] == 0).all()
    flat = ids.reshape(-1, L)
    for t in range(L - 1):
        assert all((a, b) in adj for a, b in zip(flat[:, t].tolist(), flat[:, t + 1].tolist()))
    assert (codes <= np.arange(L)[None, None, None, :]).all()      # node at step t is <= t hops away
    # different seeds differ, same seed repeats
    ids2, _ = merw.sample_full(n, u, v, g["p"], W, L, merw.DRAW_PHILOX, 1234, epoch_count=2)
    ids3, _ = merw.sample_full(n, u, v, g["p"], W, L, merw.DRAW_PHILOX, 1235, epoch_count=2)
    assert (ids == ids2).all() and (ids != ids3).any()


@pytest.mark.skipif(not merw.have_ref(), reason="oracle/_ref not built")
def test_oracle_bit_exact_against_reference_binary(tmp_path):
    """Live run of the unmodified reference program (needs only oracle/_ref, not /root/reference)."""
    g = golden("sampler_synthetic97_12_5.npz")
    n, W, L = int(g["n"]), 9, 3
    edge = os.path.join(tmp_path, "g.in")
    merw.write_edge_file(edge, n, g["u"], g["v"], g["p"])
    ids, codes = merw.sample_full(n, g["u"], g["v"], g["p"], W, L, merw.DRAW_GLIBC, 77, epoch_count=6)
    txt = merw.format_text(ids, codes)
    ref = merw.run_ref(edge, W, L, 77, max_bytes=len(txt))
    assert ref == txt


@pytest.mark.reference
@pytest.mark.skipif(not merw.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["cora", "citeseer", "cora_nsl"])
def test_oracle_bit_exact_on_shipped_graphs(name):
    f = "/root/reference/edge_input/%s.in" % name
    n, u, v, p = merw.read_edge_file(f)
    ids, codes = merw.sample_full(n, u, v, p, 40, 4, merw.DRAW_GLIBC, 5, epoch_count=2)
    txt = merw.format_text(ids, codes)
    assert merw.run_ref(f, 40, 4, 5, max_bytes=len(txt)) == txt


# ---- the uniform random-walk sampler (gen.cpp), SURVEY.md §8 f-4 ------------------------------------------------
@pytest.mark.parametrize("name", golden_files("uniform_*.npz"))
def test_uniform_oracle_reproduces_reference_golden(name):
    g = golden(name)
    n, W, L, seed, E = int(g["n"]), int(g["W"]), int(g["L"]), int(g["seed"]), int(g["epochs"])
    ids, codes = merw.sample_uniform(n, g["u"], g["v"], W, L, merw.DRAW_GLIBC, seed, epoch_count=E)
    assert (ids == g["ids"]).all()
    assert (codes == g["codes"]).all()
    assert hashlib.md5(merw.format_text(ids, codes)).hexdigest() == str(g["md5"])


def test_uniform_graph_build_keeps_self_loop_first_and_parallel_edges():
    # gen.cpp:83-94: link(i, i) for every node, then both directions of every pair with u != v, in file order
    off, nbr = merw.uniform_build(4, np.array([0, 1, 2, 0, 3], np.int32), np.array([1, 0, 2, 1, 0], np.int32))
    lists = [nbr[off[i]:off[i + 1]].tolist() for i in range(4)]
    assert lists == [[0, 1, 1, 1, 3], [1, 0, 0, 0], [2], [3, 0]]


@pytest.mark.skipif(not os.path.exists(merw.REF_GEN), reason="oracle/_ref/gen not built (reference sources absent)")
def test_uniform_oracle_matches_reference_binary_on_a_fresh_graph(tmp_path):
    rng = np.random.default_rng(77)
    n = 41
    u = np.concatenate([np.arange(n), rng.integers(0, n, 60)]).astype(np.int32)
    v = np.concatenate([(np.arange(n) + 1) % n, rng.integers(0, n, 60)]).astype(np.int32)
    f = os.path.join(tmp_path, "x_nsl.in")
    merw.write_pair_file(f, n, u, v)
    ids, codes = merw.sample_uniform(n, u, v, 6, 5, merw.DRAW_GLIBC, 31337, epoch_count=2)
    txt = merw.format_text(ids, codes)
    assert merw.run_ref_uniform(f, 6, 5, 31337, max_bytes=len(txt)) == txt
