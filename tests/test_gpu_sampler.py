"""GPU sampler (pn_sample_paths) against the reference-binary goldens and the C oracle.  Bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, golden_files
from oracle import merw

pytestmark = pytest.mark.gpu


def make_sampler(g, L=None):
    from pathnet_amd import MerwSampler
    return MerwSampler(int(g["n"]), g["u"], g["v"], g["p"], int(g["L"]) if L is None else L)


@pytest.mark.parametrize("name", golden_files("sampler_*.npz"))
def test_glibc_replay_bit_exact_vs_reference_golden(name):
    from pathnet_amd import DRAW_GLIBC_REPLAY
    g = golden(name)
    smp = make_sampler(g)
    ids, codes = smp.sample(int(g["W"]), int(g["seed"]), epoch_count=int(g["epochs"]), draw_source=DRAW_GLIBC_REPLAY)
    ids, codes = ids.cpu().numpy(), codes.cpu().numpy()
    assert ids.shape == g["ids"].shape
    bad = np.argwhere(ids != g["ids"])
    assert bad.size == 0, "first mismatch at %s: got %s want %s" % (bad[0], ids[tuple(bad[0])], g["ids"][tuple(bad[0])])
    assert (codes == g["codes"]).all()


def test_glibc_replay_windows_match_full_stream():
    from pathnet_amd import DRAW_GLIBC_REPLAY
    g = golden("sampler_cornell_7_6.npz")
    smp = make_sampler(g)
    ids, codes = smp.sample(int(g["W"]), int(g["seed"]), epoch_begin=1, epoch_count=2, node_begin=17, node_count=50,
                            draw_source=DRAW_GLIBC_REPLAY)
    assert (ids.cpu().numpy() == g["ids"][1:3, 17:67]).all()
    assert (codes.cpu().numpy() == g["codes"][1:3, 17:67]).all()


def test_glibc_replay_deep_in_the_stream_matches_oracle():
    from pathnet_amd import DRAW_GLIBC_REPLAY
    g = golden("sampler_cornell_40_4.npz")
    n, W, L = int(g["n"]), 40, 4
    smp = make_sampler(g)
    ids, codes = smp.sample(W, 20220722, epoch_begin=997, epoch_count=3, draw_source=DRAW_GLIBC_REPLAY)
    oi, oc = merw.sample_full(n, g["u"], g["v"], g["p"], W, L, merw.DRAW_GLIBC, 20220722, epoch_begin=997,
                              epoch_count=3)
    assert (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()


@pytest.mark.parametrize("name", ["sampler_synthetic97_12_5.npz", "sampler_nba_5_4.npz"])
def test_philox_bit_exact_vs_oracle(name):
    from pathnet_amd import DRAW_PHILOX
    g = golden(name)
    n, W, L = int(g["n"]), int(g["W"]), int(g["L"])
    smp = make_sampler(g)
    ids, codes = smp.sample(W, 0xC0FFEE1234, epoch_begin=3, epoch_count=2, draw_source=DRAW_PHILOX)
    oi, oc = merw.sample_full(n, g["u"], g["v"], g["p"], W, L, merw.DRAW_PHILOX, 0xC0FFEE1234, epoch_begin=3,
                              epoch_count=2)
    assert (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()


def synthetic_graph(n, deg, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, n, n * deg // 2)
    b = rng.integers(0, n, n * deg // 2)
    und = np.unique(np.stack([np.minimum(a, b), np.maximum(a, b)], 1)[a != b], axis=0)
    src = np.concatenate([und[:, 0], und[:, 1], np.arange(n)])
    dst = np.concatenate([und[:, 1], und[:, 0], np.arange(n)])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    w = rng.random(len(src)) + 0.1
    tot = np.zeros(n)
    np.add.at(tot, src, w)
    p = w / tot[src]
    u = np.repeat(src, 2).astype(np.int32)        # every row twice, like the shipped files
    v = np.repeat(dst, 2).astype(np.int32)
    return n, u, v, np.repeat(p, 2)


def test_structural_invariants_at_cora_scale_philox():
    from pathnet_amd import DRAW_PHILOX, MerwSampler
    n, u, v, p = synthetic_graph(2708, 4, 0)
    W, L = 40, 4
    smp = MerwSampler(n, u, v, p, L)
    ids, codes = smp.sample(W, 11, epoch_count=4, draw_source=DRAW_PHILOX)
    ids, codes = ids.cpu().numpy(), codes.cpu().numpy()
    assert (ids[:, :, :, 0] == np.arange(n)[None, :, None]).all() and (codes[..., 0] == 0).all()
    key = set((u.astype(np.int64) * n + v).tolist())
    flat = ids.reshape(-1, L).astype(np.int64)
    for t in range(L - 1):
        assert set((flat[:, t] * n + flat[:, t + 1]).tolist()) <= key          # every hop is an edge row
    dis = merw.bfs_dense(n, u, v, L)
    want = dis[np.repeat(np.arange(n), W)[None, :].repeat(4, 0).reshape(-1)[:, None], flat] - 1
    assert (codes.reshape(-1, L) == want).all()                                 # code = BFS hops
    oi, oc = merw.sample_full(n, u, v, p, W, L, merw.DRAW_PHILOX, 11, epoch_count=4)
    assert (ids == oi).all() and (codes == oc).all()


def test_empty_table_is_reported():
    from pathnet_amd import DRAW_PHILOX, MerwSampler, _lib
    # node 2 has no outgoing rows; walks from 0 reach it
    u = np.array([0, 0, 1, 1], np.int32)
    v = np.array([2, 2, 0, 0], np.int32)
    p = np.array([0.5, 0.5, 0.5, 0.5])
    smp = MerwSampler(3, u, v, p, 3)
    with pytest.raises(_lib.PnError) as e:
        smp.sample(4, 1, draw_source=DRAW_PHILOX)
    assert e.value.code == _lib.PN_ERR_EMPTY_TABLE


@pytest.mark.skipif(not merw.have_ref(), reason="oracle/_ref not built")
def test_cli_output_is_byte_identical_to_reference_program(tmp_path):
    """python -m pathnet_amd.sampler <name> <W> <L> vs ./gen_merw <name> <W> <L> (same srand seed)."""
    g = golden("sampler_synthetic97_12_5.npz")
    W, L, seed, epochs = 6, 4, 31337, 5
    os.makedirs(os.path.join(tmp_path, "preprocess"))
    os.makedirs(os.path.join(tmp_path, "edge_input"))
    edge = os.path.join(tmp_path, "edge_input", "syn.in")
    merw.write_edge_file(edge, int(g["n"]), g["u"], g["v"], g["p"])
    cwd = os.path.join(tmp_path, "preprocess")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, "-m", "pathnet_amd.sampler", "syn", str(W), str(L), "--seed", str(seed),
                    "--epochs", str(epochs)], cwd=cwd, env=env, check=True)
    mine = open(os.path.join(cwd, "syn_%d_%d_merw.txt" % (W, L)), "rb").read()
    ref = merw.run_ref(edge, W, L, seed, max_bytes=len(mine))
    assert mine.count(b"\n") == epochs * int(g["n"]) * W and mine == ref
    # per-epoch variant (gen_epoch_merw.cpp): same stream, one file per epoch
    subprocess.run([sys.executable, "-m", "pathnet_amd.sampler", "syn", str(W), str(L), "--seed", str(seed),
                    "--epochs", "3", "--per-epoch"], cwd=cwd, env=env, check=True)
    cat = b"".join(open(os.path.join(cwd, "syn_%d_%d_%d_merw.txt" % (W, L, e)), "rb").read() for e in range(3))
    assert cat == ref[:len(cat)]


def test_pubmed_scale_philox_bit_exact_vs_oracle():
    """configs[2]: Pubmed-size graph (19 717 nodes, dense hop table 389 MB in HBM), one epoch = 788 680 paths."""
    from pathnet_amd import DRAW_PHILOX, MerwSampler
    n, u, v, p = synthetic_graph(19717, 5, 3)
    W, L = 40, 4
    smp = MerwSampler(n, u, v, p, L)
    ids, codes = smp.sample(W, 2024, epoch_begin=7, epoch_count=1, draw_source=DRAW_PHILOX)
    oi, oc = merw.sample_full(n, u, v, p, W, L, merw.DRAW_PHILOX, 2024, epoch_begin=7, epoch_count=1)
    assert (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()
    # node window = what one of 8 ranks would sample
    lo, cnt = 3 * (n // 8), n // 8
    ids_w, codes_w = smp.sample(W, 2024, epoch_begin=7, epoch_count=1, node_begin=lo, node_count=cnt,
                                draw_source=DRAW_PHILOX)
    assert (ids_w.cpu().numpy() == oi[:, lo:lo + cnt]).all() and (codes_w.cpu().numpy() == oc[:, lo:lo + cnt]).all()


def test_bgp_scale_dense_table_window_bit_exact_vs_oracle():
    """configs[3]: a BGP-sized graph (63 977 nodes): the dense hop table dis[n][n] of gen_merw.cpp:101-123 is 4.1 GB in
    HBM; a window of source nodes (what one of 8 ranks samples) against the C restatement, which walks the same
    window over its own dense table; node lists (what a training step samples) give the same rows."""
    from pathnet_amd import DRAW_PHILOX, MerwSampler
    n, u, v, p = synthetic_graph(63977, 5, 7)
    W, L = 40, 4
    smp = MerwSampler(n, u, v, p, L)                 # dense table (n^2 = 4.1 GB <= the 8 GB switch to on-the-fly codes)
    assert smp.hops == "dense"
    lo, cnt = 5 * (n // 8), n // 8
    ids_w, codes_w = smp.sample(W, 99, epoch_begin=3, epoch_count=1, node_begin=lo, node_count=cnt, draw_source=DRAW_PHILOX)
    off, A, B, S = merw.alias_build(n, u, v, p)
    dis = merw.bfs_dense(n, u, v, L)
    oi, oc = merw.walk(n, off, A, B, S, dis, W, L, merw.DRAW_PHILOX, 99, epoch_begin=3, epoch_count=1, node_begin=lo,
                       node_count=cnt)
    assert (ids_w.cpu().numpy() == oi).all() and (codes_w.cpu().numpy() == oc).all()
    nodes = np.sort(np.random.default_rng(1).permutation(cnt)[:2000]) + lo
    ids_l, codes_l = smp.sample(W, 99, epoch_begin=3, epoch_count=1, nodes=torch.as_tensor(nodes.astype(np.int32)).cuda(),
                                draw_source=DRAW_PHILOX)
    assert (ids_l.cpu().numpy() == oi[:, nodes - lo]).all() and (codes_l.cpu().numpy() == oc[:, nodes - lo]).all()


# ------------------------------------------------------------------------------------------------
# on-the-fly hop codes (no dense n*n table): must give the same codes as the reference's bfs()
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_files("sampler_*.npz"))
def test_otf_hop_codes_glibc_replay_match_reference_golden(name):
    from pathnet_amd import DRAW_GLIBC_REPLAY, MerwSampler
    g = golden(name)
    smp = MerwSampler(int(g["n"]), g["u"], g["v"], g["p"], int(g["L"]), hops="otf")
    ids, codes = smp.sample(int(g["W"]), int(g["seed"]), epoch_count=int(g["epochs"]), draw_source=DRAW_GLIBC_REPLAY)
    assert (ids.cpu().numpy() == g["ids"]).all()
    bad = np.argwhere(codes.cpu().numpy() != g["codes"])
    assert bad.size == 0, "first code mismatch at %s" % (bad[0],)


def hub_and_directed_graph(n, seed):
    """A graph that stresses the on-the-fly search: two hubs with degree > 1100 (their 1- and 2-balls overflow the
    LDS table, forcing the radius-1 / radius-0 fallbacks), one-way edges (in-lists != out-lists), a ring."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for a in range(n):
        src += [a, a]
        dst += [a, (a + 1) % n]                    # self loop + one-way ring
    for hub in (5, 77):
        leaves = rng.permutation(n)[:1200]
        src += [hub] * len(leaves) + leaves.tolist()
        dst += leaves.tolist() + [hub] * len(leaves)
    extra = rng.integers(0, n, (3 * n, 2))
    src += extra[:, 0].tolist()
    dst += extra[:, 1].tolist()                    # one-way random edges
    src, dst = np.array(src, np.int32), np.array(dst, np.int32)
    key = np.unique(src.astype(np.int64) * n + dst)
    src, dst = (key // n).astype(np.int32), (key % n).astype(np.int32)
    deg = np.bincount(src, minlength=n)
    p = 1.0 / deg[src]
    return n, src, dst, p


@pytest.mark.parametrize("L,W", [(4, 40), (6, 12), (3, 5), (5, 9), (8, 6)])
def test_otf_hop_codes_match_dense_on_hubs_and_directed_edges(L, W):
    from pathnet_amd import DRAW_PHILOX, MerwSampler
    n, u, v, p = hub_and_directed_graph(4000, 9)
    dense = MerwSampler(n, u, v, p, L, hops="dense")
    otf = MerwSampler(n, u, v, p, L, hops="otf")
    a_ids, a_codes = dense.sample(W, 77, epoch_count=2, draw_source=DRAW_PHILOX)
    b_ids, b_codes = otf.sample(W, 77, epoch_count=2, draw_source=DRAW_PHILOX)
    assert torch.equal(a_ids, b_ids)
    bad = torch.nonzero(a_codes != b_codes)
    assert bad.numel() == 0, "first mismatch %s dense %d otf %d" % (
        bad[0].tolist(), int(a_codes[tuple(bad[0])]), int(b_codes[tuple(bad[0])]))
    oi, oc = merw.sample_full(n, u, v, p, W, L, merw.DRAW_PHILOX, 77, epoch_count=2)
    assert (b_ids.cpu().numpy() == oi).all() and (b_codes.cpu().numpy() == oc).all()


def test_otf_large_graph_beyond_the_reference_cap():
    """n = 400 000 > the reference's compiled-in 100 050 (a dense table would be 160 GB): structural invariants plus
    exact BFS hop counts from scipy for a sample of source nodes."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import dijkstra
    from pathnet_amd import DRAW_PHILOX, MerwSampler
    n, u, v, p = synthetic_graph(400000, 8, 12)
    W, L = 40, 6
    smp = MerwSampler(n, u, v, p, L, hops="auto")
    assert smp.hops == "otf"
    nodes = 2000
    ids, codes = smp.sample(W, 5, node_begin=1234, node_count=nodes, draw_source=DRAW_PHILOX)
    ids, codes = ids.cpu().numpy()[0], codes.cpu().numpy()[0]
    assert (ids[:, :, 0] == (1234 + np.arange(nodes))[:, None]).all() and (codes[..., 0] == 0).all()
    assert (codes <= np.arange(L)[None, None, :]).all()
    A = sp.csr_matrix((np.ones(len(u), np.int8), (u, v)), shape=(n, n))
    for s in (0, 17, 999, 1999):
        dist = dijkstra(A, unweighted=True, indices=1234 + s, limit=L)
        want = dist[ids[s]]
        assert (codes[s] == want).all(), (s, codes[s][:3], want[:3])


def test_cli_empty_table_behaves_like_the_reference(tmp_path):
    """gen_merw.cpp:84-87: a walk that reaches a node without outgoing rows prints the message and exits 0."""
    os.makedirs(os.path.join(tmp_path, "preprocess"))
    os.makedirs(os.path.join(tmp_path, "edge_input"))
    with open(os.path.join(tmp_path, "edge_input", "dead.in"), "w") as f:
        f.write("3 4\n0 2 0.5\n0 2 0.5\n1 0 0.5\n1 0 0.5\n")
    r = subprocess.run([sys.executable, "-m", "pathnet_amd.sampler", "dead", "4", "3", "--seed", "1", "--epochs", "2"],
                       cwd=os.path.join(tmp_path, "preprocess"), env=dict(os.environ, PYTHONPATH=ROOT),
                       capture_output=True, text=True)
    assert r.returncode == 0 and "A.size() == 0 in Alias Table" in r.stderr


# ---- the uniform random-walk sampler (gen.cpp / gen_epoch.cpp), SURVEY.md §8 f-4 ---------------------------------
def make_uniform(g, L=None, hops="auto"):
    from pathnet_amd import UniformSampler
    return UniformSampler(int(g["n"]), g["u"], g["v"], int(g["L"]) if L is None else L, hops=hops)


@pytest.mark.parametrize("name", golden_files("uniform_*.npz"))
@pytest.mark.parametrize("hops", ["dense", "otf"])
def test_uniform_glibc_replay_bit_exact_vs_reference_golden(name, hops):
    from pathnet_amd import DRAW_GLIBC_REPLAY
    g = golden(name)
    smp = make_uniform(g, hops=hops)
    ids, codes = smp.sample(int(g["W"]), int(g["seed"]), epoch_count=int(g["epochs"]), draw_source=DRAW_GLIBC_REPLAY)
    ids, codes = ids.cpu().numpy(), codes.cpu().numpy()
    bad = np.argwhere(ids != g["ids"])
    assert bad.size == 0, "first mismatch at %s: got %s want %s" % (bad[0], ids[tuple(bad[0])], g["ids"][tuple(bad[0])])
    assert (codes == g["codes"]).all()


def test_uniform_windows_and_deep_stream_match_oracle():
    from pathnet_amd import DRAW_GLIBC_REPLAY
    g = golden("uniform_g300_40_4.npz")
    n, W, L, seed = int(g["n"]), int(g["W"]), int(g["L"]), int(g["seed"])
    smp = make_uniform(g)
    ids, codes = smp.sample(W, seed, epoch_begin=1, epoch_count=1, node_begin=40, node_count=111,
                            draw_source=DRAW_GLIBC_REPLAY)
    assert (ids.cpu().numpy() == g["ids"][1:2, 40:151]).all() and (codes.cpu().numpy() == g["codes"][1:2, 40:151]).all()
    ids, codes = smp.sample(W, 7, epoch_begin=998, epoch_count=2, draw_source=DRAW_GLIBC_REPLAY)
    oi, oc = merw.sample_uniform(n, g["u"], g["v"], W, L, merw.DRAW_GLIBC, 7, epoch_begin=998, epoch_count=2)
    assert (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()


def test_uniform_philox_bit_exact_vs_oracle():
    from pathnet_amd import DRAW_PHILOX
    g = golden("uniform_g120_7_6.npz")
    n, W, L = int(g["n"]), int(g["W"]), int(g["L"])
    smp = make_uniform(g)
    ids, codes = smp.sample(W, 0xC0FFEE, epoch_begin=3, epoch_count=2, draw_source=DRAW_PHILOX)
    oi, oc = merw.sample_uniform(n, g["u"], g["v"], W, L, merw.DRAW_PHILOX, 0xC0FFEE, epoch_begin=3, epoch_count=2)
    assert (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()
    # one draw per step: every step lands on a neighbour (or the self loop) of the previous node
    off, nbr = merw.uniform_build(n, g["u"], g["v"])
    a = ids.cpu().numpy().reshape(-1, L)
    for row in a[:500]:
        for t in range(L - 1):
            assert row[t + 1] in nbr[off[row[t]]:off[row[t] + 1]]


def test_uniform_cli_output_is_byte_identical_to_reference_program(tmp_path):
    """python -m pathnet_amd.sampler <name> <W> <L> --uniform  vs  ./gen <name> <W> <L> (same srand seed)."""
    g = golden("uniform_ring37_5_4.npz")
    n, W, L, seed, epochs = int(g["n"]), 6, 5, 4242, 4
    os.makedirs(os.path.join(tmp_path, "preprocess"))
    os.makedirs(os.path.join(tmp_path, "edge_input"))
    pair = os.path.join(tmp_path, "edge_input", "syn_nsl.in")
    merw.write_pair_file(pair, n, g["u"], g["v"])
    cwd = os.path.join(tmp_path, "preprocess")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "pathnet_amd.sampler", "syn", str(W), str(L), "--uniform", "--seed",
                        str(seed), "--epochs", str(epochs)], cwd=cwd, env=env, check=True, capture_output=True, text=True)
    assert "File input: ../edge_input/syn_nsl.in" in r.stdout and "File output: ./syn_%d_%d_nsl.txt" % (W, L) in r.stdout
    mine = open(os.path.join(cwd, "syn_%d_%d_nsl.txt" % (W, L)), "rb").read()
    want = merw.format_text(*merw.sample_uniform(n, g["u"], g["v"], W, L, merw.DRAW_GLIBC, seed, epoch_count=epochs))
    assert mine == want
    if os.path.exists(merw.REF_GEN):
        assert merw.run_ref_uniform(pair, W, L, seed, max_bytes=len(mine)) == mine
    # gen_epoch.cpp: reads <name>.in, one file per epoch without a marker, prints "n m"
    merw.write_pair_file(os.path.join(tmp_path, "edge_input", "syn.in"), n, g["u"], g["v"])
    r = subprocess.run([sys.executable, "-m", "pathnet_amd.sampler", "syn", str(W), str(L), "--uniform", "--per-epoch",
                        "--seed", str(seed), "--epochs", "3"], cwd=cwd, env=env, check=True, capture_output=True,
                       text=True)
    assert r.stdout.split() == [str(n), str(len(g["u"]))]
    cat = b"".join(open(os.path.join(cwd, "syn_%d_%d_%d.txt" % (W, L, e)), "rb").read() for e in range(3))
    assert cat == want[:len(cat)]


def test_node_list_window_equals_rows_of_the_full_epoch():
    import pathnet_amd
    from pathnet_amd import _lib
    """pn_sample_paths(node_list): a training step samples the paths of its masked nodes only; a walk's Philox draws are
    a function of (epoch, source node, walk index), so these are the rows a full-epoch sample holds for those nodes --
    with the dense hop table and with on-the-fly hop codes, L = 4 (vector stores) and L = 5."""
    import bench
    n, u, v, p = bench.synthetic_graph(700, 9)
    rng = np.random.default_rng(9)
    for L in (4, 5):
        for hops in ("dense", "otf"):
            smp = pathnet_amd.MerwSampler(n, u, v, p, L, device="cuda", hops=hops)
            full_i, full_c = smp.sample(13, 77, epoch_begin=3, epoch_count=2)
            nodes = torch.as_tensor(rng.permutation(n)[:123].astype(np.int32)).cuda()       # any order, any subset
            for stage in ("0", "1"):        # first-hop tables from L2 / staged in LDS (the default only stages large launches)
                old = _lib.set_knob("PN_SAMPLER_STAGE", int(stage))
                try:
                    got_i, got_c = smp.sample(13, 77, epoch_begin=3, epoch_count=2, nodes=nodes)
                finally:
                    _lib.set_knob("PN_SAMPLER_STAGE", old)
                assert got_i.shape == (2, 123, 13, L)
                assert torch.equal(got_i, full_i[:, nodes.long()]) and torch.equal(got_c, full_c[:, nodes.long()])
            oi, oc = merw.sample_full(n, u, v, p, 13, L, merw.DRAW_PHILOX, 77, epoch_begin=3, epoch_count=2)
            assert (full_i.cpu().numpy() == oi).all() and (full_c.cpu().numpy() == oc).all()
    with pytest.raises(ValueError):
        smp.sample(13, 77, nodes=nodes.long())
    with pytest.raises(_lib.PnError):
        smp.sample(13, 77, nodes=nodes, draw_source=pathnet_amd.DRAW_GLIBC_REPLAY)


def test_walker_without_the_packed_node_refs():
    import pathnet_amd
    """the 8-byte {first triple, count} word per node is an optimisation of the two-word off[] lookup: same paths"""
    g = golden("sampler_synthetic97_12_5.npz")
    n, W, L = int(g["n"]), int(g["W"]), int(g["L"])
    smp = pathnet_amd.MerwSampler(n, g["u"], g["v"], g["p"], L, device="cuda")
    assert smp.d_node_ref is not None
    a = smp.sample(W, int(g["seed"]), epoch_count=int(g["epochs"]), draw_source=pathnet_amd.DRAW_GLIBC_REPLAY)
    smp.d_node_ref = None
    b = smp.sample(W, int(g["seed"]), epoch_count=int(g["epochs"]), draw_source=pathnet_amd.DRAW_GLIBC_REPLAY)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and (a[0].cpu().numpy() == g["ids"]).all()
