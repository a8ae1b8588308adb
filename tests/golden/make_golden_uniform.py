"""Generate tests/golden/uniform_*.npz by running the UNMODIFIED uniform random-walk sampler
(oracle/_ref/gen, compiled from /root/reference/preprocess/gen.cpp) under the fixed-seed time() shim.
Run in the build container only:  python tests/golden/make_golden_uniform.py

The reference ships no 2-column `<name>_nsl.in` input (only the 3-column cora_nsl.in), so the
inputs are synthetic pair lists -- with self pairs (skipped by the program) and repeated lines (kept as
parallel edges), the two things its graph build treats specially (gen.cpp:83-94).
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import merw  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def parse_text(txt, L):
    rows = [list(map(int, ln[1:-1].split(","))) for ln in txt.decode().strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    return a[:, :L].astype(np.int32), a[:, L:].astype(np.uint8)


def pair_list(n, m, seed):
    rng = np.random.default_rng(seed)
    pairs = [(a, (a + 3) % n) for a in range(n)]                     # connected ring-ish backbone
    while len(pairs) < m:
        a, b = rng.integers(0, n, 2)
        pairs.append((int(a), int(b)))                               # may be a self pair or a repeat
    pairs += pairs[: max(1, m // 12)]                                # repeated lines
    return np.array([a for a, _ in pairs], np.int32), np.array([b for _, b in pairs], np.int32)


def one(tag, n, m, W, L, seed, epochs):
    u, v = pair_list(n, m, seed)
    f = tempfile.mktemp(suffix=".in")
    merw.write_pair_file(f, n, u, v)
    nbytes = len(merw.format_text(*merw.sample_uniform(n, u, v, W, L, merw.DRAW_GLIBC, seed, epoch_count=epochs)))
    txt = merw.run_ref_uniform(f, W, L, seed, max_bytes=nbytes)
    os.remove(f)
    assert len(txt) == nbytes
    ids, codes = parse_text(txt, L)
    np.savez_compressed(os.path.join(OUT, "uniform_%s.npz" % tag), n=n, u=u, v=v, W=W, L=L, seed=seed,
                        epochs=epochs, ids=ids.reshape(epochs, n, W, L), codes=codes.reshape(epochs, n, W, L),
                        md5=hashlib.md5(txt).hexdigest())
    print(tag, "n", n, "pairs", len(u), "paths", ids.shape[0], hashlib.md5(txt).hexdigest())


if __name__ == "__main__":
    assert os.path.exists(merw.REF_GEN), "build oracle/_ref first (make -C oracle)"
    one("ring37_5_4", 37, 90, 5, 4, 1234, 3)
    one("g300_40_4", 300, 900, 40, 4, 20240917, 2)
    one("g120_7_6", 120, 260, 7, 6, 99, 2)
