"""Generate tests/golden/sampler_*.npz by running the UNMODIFIED reference sampler
(oracle/_ref/gen_merw, compiled from /root/reference/preprocess/gen_merw.cpp) under the fixed-seed
time() shim.  Run in the build container only:  python tests/golden/make_golden_sampler.py

Each fixture holds the parsed edge list (n, u, v, p -- test input data, not source code), the seed,
W, L, the first `epochs` epochs of reference output parsed to ids/codes, and the md5 of those bytes.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import merw  # noqa: E402

REF_EDGE = "/root/reference/edge_input"
OUT = os.path.dirname(os.path.abspath(__file__))


def parse_text(txt, L):
    rows = [list(map(int, ln[1:-1].split(","))) for ln in txt.decode().strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    return a[:, :L].astype(np.int32), a[:, L:].astype(np.uint8)


def one(tag, edge_file, W, L, seed, epochs, full_md5=False):
    n, u, v, p = merw.read_edge_file(edge_file)
    # size of the first `epochs` epochs: get it from the restatement's formatter (already pinned)
    ids_o, codes_o = merw.sample_full(n, u, v, p, W, L, merw.DRAW_GLIBC, seed, epoch_count=epochs)
    nbytes = len(merw.format_text(ids_o, codes_o))
    txt = merw.run_ref(edge_file, W, L, seed, max_bytes=nbytes)
    assert len(txt) == nbytes
    ids, codes = parse_text(txt, L)
    d = dict(n=n, u=u, v=v, p=p, W=W, L=L, seed=seed, epochs=epochs,
             ids=ids.reshape(epochs, n, W, L), codes=codes.reshape(epochs, n, W, L),
             md5=hashlib.md5(txt).hexdigest())
    if full_md5:
        full = merw.run_ref(edge_file, W, L, seed)
        d["full_md5"] = hashlib.md5(full).hexdigest()
        d["full_bytes"] = len(full)
    np.savez_compressed(os.path.join(OUT, "sampler_%s.npz" % tag), **d)
    print(tag, "n", n, "m", len(u), "paths", ids.shape[0], d["md5"], d.get("full_md5"))


def synthetic_edge_file(path, n, seed):
    """A small graph in the shipped files' style: symmetric, self loops, every row duplicated
    (init_rw.py:83-86), some negative and >1 'probabilities' like cora/citeseer."""
    rng = np.random.default_rng(seed)
    und = set()
    for a in range(n):
        und.add((a, (a + 1) % n) if a < (a + 1) % n else ((a + 1) % n, a))
    while len(und) < 3 * n:
        a, b = rng.integers(0, n, 2)
        if a != b:
            und.add((min(a, b), max(a, b)))
    nbrs = [[] for _ in range(n)]
    for a, b in sorted(und):
        nbrs[a].append(b)
        nbrs[b].append(a)
    rows = []
    for a in range(n):
        lst = [a] + nbrs[a]
        w = rng.random(len(lst)) + 0.05
        if a % 7 == 3:                      # pathological rows
            w[0] = -w[0] * 3
            w[-1] = w[-1] * 9
        w = w / np.abs(w).sum()
        for b, pw in zip(lst, w):
            rows.append((a, b, pw))
            rows.append((a, b, pw))
    u = np.array([r[0] for r in rows], np.int32)
    v = np.array([r[1] for r in rows], np.int32)
    p = np.array([r[2] for r in rows], np.float64)
    merw.write_edge_file(path, n, u, v, p)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    one("cornell_40_4", os.path.join(REF_EDGE, "cornell.in"), 40, 4, 1, 2, full_md5=True)
    one("cornell_7_6", os.path.join(REF_EDGE, "cornell.in"), 7, 6, 20220722, 3)
    one("nba_5_4", os.path.join(REF_EDGE, "Nba.in"), 5, 4, 3, 2)
    # the shipped graphs with pathological rows (disconnected: negative and > 1 "probabilities", row sums -92 .. +35,
    # SURVEY.md 8a-1): the alias builder's re-queue paths and the walker see them through the reference binary's output
    one("cora_40_4", os.path.join(REF_EDGE, "cora.in"), 40, 4, 7, 1)
    one("citeseer_10_4", os.path.join(REF_EDGE, "citeseer.in"), 10, 4, 11, 2)
    syn = "/tmp/pn_syn_edges.in"
    synthetic_edge_file(syn, 97, 5)
    one("synthetic97_12_5", syn, 12, 5, 42, 4)
