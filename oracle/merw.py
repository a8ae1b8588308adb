"""ctypes front-end of oracle/merw_oracle.c + a runner for the unmodified reference binary.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference behaviour restated:
/root/reference/preprocess/gen_merw.cpp (line ranges are listed in merw_oracle.c).
"""
import ctypes
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libmerw_oracle.so")
SHIM_PATH = os.path.join(HERE, "_build", "libtimeshim.so")
REF_GEN_MERW = os.path.join(HERE, "_ref", "gen_merw")
REF_GEN_EPOCH_MERW = os.path.join(HERE, "_ref", "gen_epoch_merw")
REF_GEN = os.path.join(HERE, "_ref", "gen")                    # uniform random walks (gen.cpp)
REF_GEN_EPOCH = os.path.join(HERE, "_ref", "gen_epoch")

DRAW_GLIBC = 0
DRAW_PHILOX = 1

_lib = None


def build():
    """(Re)build the oracle libraries (and oracle/_ref when the reference sources are mounted)."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
        f64p, u8p = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint8)
        L.mo_glibc_stream.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, i32p]
        L.mo_glibc_stream.restype = None
        L.mo_philox_draw.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        L.mo_philox_draw.restype = ctypes.c_uint32
        L.mo_alias_build.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, f64p, i64p, i32p, i32p, f64p,
                                     ctypes.c_int64]
        L.mo_alias_build.restype = ctypes.c_int64
        L.mo_bfs_dense.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, ctypes.c_int32, u8p]
        L.mo_bfs_dense.restype = ctypes.c_int
        L.mo_walk.argtypes = [ctypes.c_int32, i64p, i32p, i32p, f64p, u8p, ctypes.c_int32, ctypes.c_int32,
                              ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                              ctypes.c_int32, i32p, u8p]
        L.mo_walk.restype = ctypes.c_int
        L.mo_uniform_build.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, i64p, i32p, ctypes.c_int64]
        L.mo_uniform_build.restype = ctypes.c_int64
        L.mo_walk_uniform.argtypes = [ctypes.c_int32, i64p, i32p, u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int,
                                      ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                      i32p, u8p]
        L.mo_walk_uniform.restype = ctypes.c_int
        L.mo_format_text.argtypes = [i32p, u8p, ctypes.c_int64, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64]
        L.mo_format_text.restype = ctypes.c_int64
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def read_edge_file(path):
    """Parse '<n> <m>' + m rows 'u v p' (gen_merw.cpp:162-172) -> (n, u[int32], v[int32], p[float64])."""
    with open(path) as f:
        tok = f.read().split()
    n, m = int(tok[0]), int(tok[1])
    body = tok[2:2 + 3 * m]
    u = np.array(body[0::3], dtype=np.int64).astype(np.int32)
    v = np.array(body[1::3], dtype=np.int64).astype(np.int32)
    p = np.array([float(x) for x in body[2::3]], dtype=np.float64)
    return n, u, v, p


def write_edge_file(path, n, u, v, p):
    """Write an edge file whose doubles round-trip exactly through scanf("%lf")."""
    with open(path, "w") as f:
        f.write("%d %d\n" % (n, len(u)))
        for a, b, c in zip(u.tolist(), v.tolist(), p.tolist()):
            f.write("%d %d %s\n" % (a, b, repr(float(c))))


def glibc_stream(seed, count, skip=0):
    out = np.empty(count, dtype=np.int32)
    lib().mo_glibc_stream(seed, skip, count, _p(out, ctypes.c_int32))
    return out


def philox_draw(seed, sub, q):
    return int(lib().mo_philox_draw(seed, sub, q))


def alias_build(n, u, v, p):
    """-> off[int64 n+1], A[int32], B[int32], S[float64]   (AliasTable::init, gen_merw.cpp:23-79)"""
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    off = np.zeros(n + 1, dtype=np.int64)
    dummy_i, dummy_f = np.zeros(1, np.int32), np.zeros(1, np.float64)
    total = lib().mo_alias_build(n, len(u), _p(u, ctypes.c_int32), _p(v, ctypes.c_int32), _p(p, ctypes.c_double),
                                 _p(off, ctypes.c_int64), _p(dummy_i, ctypes.c_int32), _p(dummy_i, ctypes.c_int32),
                                 _p(dummy_f, ctypes.c_double), 0)
    if total < 0:
        raise ValueError("bad edge list")
    A = np.empty(max(total, 1), np.int32)
    B = np.empty(max(total, 1), np.int32)
    S = np.empty(max(total, 1), np.float64)
    lib().mo_alias_build(n, len(u), _p(u, ctypes.c_int32), _p(v, ctypes.c_int32), _p(p, ctypes.c_double),
                         _p(off, ctypes.c_int64), _p(A, ctypes.c_int32), _p(B, ctypes.c_int32),
                         _p(S, ctypes.c_double), total)
    return off, A[:total], B[:total], S[:total]


def bfs_dense(n, u, v, seq_len):
    """-> dis[n, n] uint8 = 1 + hops, 0 = not labelled   (bfs, gen_merw.cpp:101-123)"""
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    dis = np.zeros((n, n), dtype=np.uint8)
    rc = lib().mo_bfs_dense(n, len(u), _p(u, ctypes.c_int32), _p(v, ctypes.c_int32), seq_len, _p(dis, ctypes.c_uint8))
    if rc != 0:
        raise MemoryError
    return dis


def walk(n, off, A, B, S, dis, W, L, draw_source, seed, epoch_begin=0, epoch_count=1, node_begin=0,
         node_count=None):
    """-> ids[int32 E, nodes, W, L], codes[uint8 ...]   (walk loop, gen_merw.cpp:182-209)"""
    if node_count is None:
        node_count = n - node_begin
    ids = np.empty((epoch_count, node_count, W, L), dtype=np.int32)
    codes = np.empty((epoch_count, node_count, W, L), dtype=np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    A = np.ascontiguousarray(A, np.int32)
    B = np.ascontiguousarray(B, np.int32)
    S = np.ascontiguousarray(S, np.float64)
    dis = np.ascontiguousarray(dis, np.uint8)
    rc = lib().mo_walk(n, _p(off, ctypes.c_int64), _p(A, ctypes.c_int32), _p(B, ctypes.c_int32),
                       _p(S, ctypes.c_double), _p(dis, ctypes.c_uint8), W, L, draw_source, seed, epoch_begin,
                       epoch_count, node_begin, node_count, _p(ids, ctypes.c_int32), _p(codes, ctypes.c_uint8))
    if rc != 0:
        raise RuntimeError("mo_walk rc=%d (empty alias table reached)" % rc)
    return ids, codes


def format_text(ids, codes):
    """ids/codes [..., L] -> bytes in the reference line format (gen_merw.cpp:189-206)."""
    L = ids.shape[-1]
    ids = np.ascontiguousarray(ids.reshape(-1, L), np.int32)
    codes = np.ascontiguousarray(codes.reshape(-1, L), np.uint8)
    cap = ids.shape[0] * (2 * L * 13 + 8) + 64
    buf = ctypes.create_string_buffer(cap)
    w = lib().mo_format_text(_p(ids, ctypes.c_int32), _p(codes, ctypes.c_uint8), ids.shape[0], L, buf, cap)
    if w < 0:
        raise MemoryError
    return buf.raw[:w]


def sample_full(n, u, v, p, W, L, draw_source, seed, **kw):
    """Edge list -> (ids, codes): the whole reference pipeline through the restatement."""
    off, A, B, S = alias_build(n, u, v, p)
    dis = bfs_dense(n, u, v, L)
    return walk(n, off, A, B, S, dis, W, L, draw_source, seed, **kw)


# ------------------------------------------------------------------------------------------------
# the uniform random-walk sampler (gen.cpp / gen_epoch.cpp), restated in merw_oracle.c
# ------------------------------------------------------------------------------------------------
def read_pair_file(path):
    """-> n, u[int32 m], v[int32 m]: "n m" then m pairs, read like gen.cpp:80-92 reads them (scanf("%d%d"))."""
    tok = open(path).read().split()
    n, m = int(tok[0]), int(tok[1])
    a = np.array(tok[2:2 + 2 * m], dtype=np.int64).reshape(m, 2)
    return n, a[:, 0].astype(np.int32), a[:, 1].astype(np.int32)


def write_pair_file(path, n, u, v):
    with open(path, "w") as f:
        f.write("%d %d\n" % (n, len(u)))
        for a, b in zip(u, v):
            f.write("%d %d\n" % (a, b))


def uniform_build(n, u, v):
    """-> off[int64 n+1], nbr[int32]: self loop first, then both directions of every pair in file order."""
    u = np.ascontiguousarray(u, np.int32)
    v = np.ascontiguousarray(v, np.int32)
    total = lib().mo_uniform_build(n, len(u), _p(u, ctypes.c_int32), _p(v, ctypes.c_int32), None, None, 0)
    if total < 0:
        raise ValueError("node id out of range")
    off = np.zeros(n + 1, np.int64)
    nbr = np.zeros(total, np.int32)
    lib().mo_uniform_build(n, len(u), _p(u, ctypes.c_int32), _p(v, ctypes.c_int32), _p(off, ctypes.c_int64),
                           _p(nbr, ctypes.c_int32), total)
    return off, nbr


def sample_uniform(n, u, v, W, L, draw_source, seed, epoch_begin=0, epoch_count=1, node_begin=0, node_count=None):
    """Pair list -> (ids, codes): the whole gen.cpp pipeline through the restatement."""
    if node_count is None:
        node_count = n - node_begin
    off, nbr = uniform_build(n, u, v)
    src = np.repeat(np.arange(n, dtype=np.int32), np.diff(off))
    dis = bfs_dense(n, src, nbr, L)                                  # the same BFS over the expanded lists
    ids = np.empty((epoch_count, node_count, W, L), dtype=np.int32)
    codes = np.empty((epoch_count, node_count, W, L), dtype=np.uint8)
    rc = lib().mo_walk_uniform(n, _p(off, ctypes.c_int64), _p(nbr, ctypes.c_int32), _p(dis, ctypes.c_uint8), W, L,
                               draw_source, seed, epoch_begin, epoch_count, node_begin, node_count,
                               _p(ids, ctypes.c_int32), _p(codes, ctypes.c_uint8))
    if rc != 0:
        raise RuntimeError("mo_walk_uniform rc=%d" % rc)
    return ids, codes


def run_ref_uniform(pair_file, W, L, seed, name="g", max_bytes=None):
    """Run oracle/_ref/gen (unmodified gen.cpp) with srand pinned to `seed`; returns the bytes of
    `<name>_<W>_<L>_nsl.txt`, truncated to max_bytes (the program always walks 1000 epochs)."""
    tmp = tempfile.mkdtemp(prefix="pn_ref_")
    try:
        os.makedirs(os.path.join(tmp, "preprocess"))
        os.makedirs(os.path.join(tmp, "edge_input"))
        shutil.copy(pair_file, os.path.join(tmp, "edge_input", name + "_nsl.in"))      # gen.cpp:52-56
        cwd = os.path.join(tmp, "preprocess")
        out_name = "%s_%d_%d_nsl.txt" % (name, W, L)                                  # gen.cpp:58-68
        env = dict(os.environ, LD_PRELOAD=SHIM_PATH, PN_FAKE_TIME=str(seed))
        cmd = [REF_GEN, name, str(W), str(L)]
        if max_bytes is None:
            subprocess.run(cmd, cwd=cwd, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(os.path.join(cwd, out_name), "rb") as f:
                return f.read()
        os.mkfifo(os.path.join(cwd, out_name))
        proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        chunks, got = [], 0
        try:
            with open(os.path.join(cwd, out_name), "rb") as f:
                while got < max_bytes:
                    b = f.read(min(1 << 20, max_bytes - got))
                    if not b:
                        break
                    chunks.append(b)
                    got += len(b)
        finally:
            proc.kill()
            proc.wait()
        return b"".join(chunks)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------
# the unmodified reference program (oracle/_ref), run in a scratch dir laid out the way it expects:
#   <tmp>/preprocess/  (cwd)   and   <tmp>/edge_input/<name>.in       (gen_merw.cpp:134-151)
# ------------------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_GEN_MERW) and os.path.exists(SHIM_PATH)


def run_ref(edge_file, W, L, seed, name="g", max_bytes=None, per_epoch=False, to_devnull=False, timeout=None):
    """Run oracle/_ref/gen_merw (or gen_epoch_merw) with srand pinned to `seed`.

    Returns the output bytes of `<name>_<W>_<L>_merw.txt` (truncated to max_bytes), or for
    per_epoch=True a function epoch -> bytes.  With to_devnull=True the output file is a symlink
    to /dev/null (timing runs) and b"" is returned.
    """
    exe = REF_GEN_EPOCH_MERW if per_epoch else REF_GEN_MERW
    tmp = tempfile.mkdtemp(prefix="pn_ref_")
    try:
        os.makedirs(os.path.join(tmp, "preprocess"))
        os.makedirs(os.path.join(tmp, "edge_input"))
        shutil.copy(edge_file, os.path.join(tmp, "edge_input", name + ".in"))
        cwd = os.path.join(tmp, "preprocess")
        out_name = "%s_%d_%d_merw.txt" % (name, W, L)
        env = dict(os.environ, LD_PRELOAD=SHIM_PATH, PN_FAKE_TIME=str(seed))
        cmd = [exe, name, str(W), str(L)]
        if per_epoch:
            subprocess.run(cmd, cwd=cwd, env=env, check=True, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = {}
            for e in range(1000 if max_bytes is None else max_bytes):
                with open(os.path.join(cwd, "%s_%d_%d_%d_merw.txt" % (name, W, L, e)), "rb") as f:
                    data[e] = f.read()
            return data
        if to_devnull:
            os.symlink("/dev/null", os.path.join(cwd, out_name))
            subprocess.run(cmd, cwd=cwd, env=env, check=True, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return b""
        if max_bytes is None:
            subprocess.run(cmd, cwd=cwd, env=env, check=True, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(os.path.join(cwd, out_name), "rb") as f:
                return f.read()
        # partial read: the output file is a FIFO; we stop reading after max_bytes and the
        # program (which always walks 1000 epochs, gen_merw.cpp:182) dies on SIGPIPE.
        os.mkfifo(os.path.join(cwd, out_name))
        proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        chunks, got = [], 0
        try:
            with open(os.path.join(cwd, out_name), "rb") as f:
                while got < max_bytes:
                    b = f.read(min(1 << 20, max_bytes - got))
                    if not b:
                        break
                    chunks.append(b)
                    got += len(b)
        finally:
            proc.kill()
            proc.wait()
        return b"".join(chunks)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
