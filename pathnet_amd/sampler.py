"""MERW path sampler: host set-up + GPU walker, and the gen_merw / gen_epoch_merw command line.

Mirrors /root/reference/preprocess/gen_merw.cpp: ``main`` (:125-213) becomes
``MerwSampler.from_edge_file(...)`` (parse :162-172, alias tables :174-176, hop table :178-179) and
``MerwSampler.sample(...)`` (walk loop :182-209) which runs on the GPU through libpathnet_hip.so.
``python -m pathnet_amd.sampler <data_name> <path_num> <path_length>`` keeps the reference CLI: reads
``../edge_input/<name>.in`` and writes ``./<name>_<W>_<L>_merw.txt`` (or one file per epoch with
``--per-epoch``, gen_epoch_merw.cpp:166-178), 1000 epochs.
"""
import ctypes
import sys
import time

import numpy as np
import torch

from . import _lib, pathfile

DRAW_GLIBC_REPLAY = _lib.DRAW_GLIBC_REPLAY
DRAW_PHILOX = _lib.DRAW_PHILOX


def read_edge_file(path):
    """-> (n, u int32[m], v int32[m], p float64[m]) in file order (gen_merw.cpp:162-172)."""
    lib = _lib.load()
    n, m = ctypes.c_int32(0), ctypes.c_int64(0)
    _lib.check(lib.pn_edges_read_text(str(path).encode(), ctypes.byref(n), ctypes.byref(m), None, None, None, 0))
    u = np.empty(m.value, np.int32)
    v = np.empty(m.value, np.int32)
    p = np.empty(m.value, np.float64)
    if m.value:
        _lib.check(lib.pn_edges_read_text(str(path).encode(), ctypes.byref(n), ctypes.byref(m),
                                          _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32),
                                          _lib.np_ptr(p, ctypes.c_double), m.value))
    return n.value, u, v, p


def read_pair_file(path):
    """-> (n, u int32[m], v int32[m]) in file order: the uniform sampler's 2-column input (gen.cpp:80-92)."""
    lib = _lib.load()
    n, m = ctypes.c_int32(0), ctypes.c_int64(0)
    _lib.check(lib.pn_pairs_read_text(str(path).encode(), ctypes.byref(n), ctypes.byref(m), None, None, 0))
    u = np.empty(m.value, np.int32)
    v = np.empty(m.value, np.int32)
    if m.value:
        _lib.check(lib.pn_pairs_read_text(str(path).encode(), ctypes.byref(n), ctypes.byref(m),
                                          _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32), m.value))
    return n.value, u, v


def build_uniform(n, u, v):
    """The graph gen.cpp:83-94 walks on -> off[int64 n+1], packed int32 [total*4] walker table, and the same graph as
    a directed edge list (src, nbr) for the hop table / CSR builders."""
    lib = _lib.load()
    u = np.ascontiguousarray(u, np.int32)
    v = np.ascontiguousarray(v, np.int32)
    total = ctypes.c_int64(0)
    args = (n, len(u), _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32))
    _lib.check(lib.pn_uniform_build(*args, None, None, None, None, 0, ctypes.byref(total)))
    t = max(total.value, 1)
    off = np.zeros(n + 1, np.int64)
    packed = np.zeros(t * 4, np.int32)
    src, nbr = np.zeros(t, np.int32), np.zeros(t, np.int32)
    _lib.check(lib.pn_uniform_build(*args, _lib.np_ptr(off, ctypes.c_int64), _lib.np_ptr(packed, ctypes.c_int32),
                                    _lib.np_ptr(src, ctypes.c_int32), _lib.np_ptr(nbr, ctypes.c_int32), total.value,
                                    ctypes.byref(total)))
    k = total.value
    return off, packed, src[:k], nbr[:k]


def build_alias(n, u, v, p):
    """AliasTable::init for every node (gen_merw.cpp:23-79) -> off[int64 n+1], A, B, S(float64), thr(uint32)."""
    lib = _lib.load()
    u = np.ascontiguousarray(u, np.int32)
    v = np.ascontiguousarray(v, np.int32)
    p = np.ascontiguousarray(p, np.float64)
    off = np.zeros(n + 1, np.int64)
    total = ctypes.c_int64(0)
    args = (n, len(u), _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32), _lib.np_ptr(p, ctypes.c_double),
            _lib.np_ptr(off, ctypes.c_int64))
    _lib.check(lib.pn_alias_build(*args, None, None, None, None, 0, ctypes.byref(total)))
    t = max(total.value, 1)
    A, B = np.empty(t, np.int32), np.empty(t, np.int32)
    S, thr = np.empty(t, np.float64), np.empty(t, np.uint32)
    _lib.check(lib.pn_alias_build(*args, _lib.np_ptr(A, ctypes.c_int32), _lib.np_ptr(B, ctypes.c_int32),
                                  _lib.np_ptr(S, ctypes.c_double), _lib.np_ptr(thr, ctypes.c_uint32), total.value,
                                  ctypes.byref(total)))
    k = total.value
    return off, A[:k], B[:k], S[:k], thr[:k]


def hops_dense(n, u, v, seq_len):
    """dis[n, n] uint8 = 1 + hops for nodes within seq_len-1 hops (bfs, gen_merw.cpp:101-123)."""
    u = np.ascontiguousarray(u, np.int32)
    v = np.ascontiguousarray(v, np.int32)
    dis = np.empty((n, n), np.uint8)
    _lib.check(_lib.load().pn_hops_dense(n, len(u), _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32),
                                         seq_len, _lib.np_ptr(dis, ctypes.c_uint8)))
    return dis


def csr_build(n, u, v, reverse=False):
    """Sorted duplicate-free neighbour lists: off int64 [n+1], adj int32 (out-neighbours, or in-neighbours)."""
    lib = _lib.load()
    u = np.ascontiguousarray(u, np.int32)
    v = np.ascontiguousarray(v, np.int32)
    off = np.zeros(n + 1, np.int64)
    cnt = ctypes.c_int64(0)
    args = (n, len(u), _lib.np_ptr(u, ctypes.c_int32), _lib.np_ptr(v, ctypes.c_int32), 1 if reverse else 0,
            _lib.np_ptr(off, ctypes.c_int64))
    _lib.check(lib.pn_csr_build(*args, None, 0, ctypes.byref(cnt)))
    adj = np.empty(max(cnt.value, 1), np.int32)
    _lib.check(lib.pn_csr_build(*args, _lib.np_ptr(adj, ctypes.c_int32), cnt.value, ctypes.byref(cnt)))
    return off, adj[:cnt.value]


def glibc_draws(seed, first, count):
    out = np.empty(count, np.int32)
    _lib.check(_lib.load().pn_glibc_draws(seed & 0xFFFFFFFF, first, count, _lib.np_ptr(out, ctypes.c_int32)))
    return out


class MerwSampler:
    """Device-resident sampler tables for one graph and one path length."""

    DENSE_LIMIT_BYTES = 8 << 30     # hops="auto": dense n*n table up to 8 GB (n <= 92681), else on the fly

    draws_per_step = 2              # the alias roll: slot draw + probability draw (gen_merw.cpp:81-91)

    def __init__(self, n, u, v, p, seq_len, device="cuda", hops="auto"):
        """hops: "dense" = the reference's dis[n][n] byte table in HBM; "otf" = exact hop codes derived on the
        fly from CSR lists (any n); "auto" picks dense while n*n <= DENSE_LIMIT_BYTES."""
        off, A, B, S, thr = build_alias(n, u, v, p)
        self.host = dict(off=off, A=A, B=B, S=S, thr=thr)
        packed = np.empty(max(len(A), 1) * 4, np.int32)
        _lib.check(_lib.load().pn_alias_pack(len(A), _lib.np_ptr(A, ctypes.c_int32), _lib.np_ptr(B, ctypes.c_int32),
                                             _lib.np_ptr(thr, ctypes.c_uint32), _lib.np_ptr(packed, ctypes.c_int32)))
        self._upload(n, seq_len, device, hops, off, packed, len(A), u, v)

    def _upload(self, n, seq_len, device, hops, off, packed, total, eu, ev):
        """Device copies of the walker table (off, packed) and of the hop-code source built from the directed edge
        list (eu -> ev): the dense table or the two CSR lists."""
        self.n, self.L = int(n), int(seq_len)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if hops == "auto":
            hops = "dense" if self.n * self.n <= self.DENSE_LIMIT_BYTES else "otf"
        if hops not in ("dense", "otf"):
            raise ValueError("hops must be 'dense', 'otf' or 'auto'")
        self.hops = hops
        self.d_off = torch.from_numpy(off).to(self.device)
        self.d_triples = torch.from_numpy(packed).to(self.device)
        self.d_node_ref = None
        if int(off[-1]) < 2 ** 32:          # {first triple, count} per node: one 8-byte load per roll
            ref = np.empty(2 * max(int(n), 1), np.uint32)
            _lib.check(_lib.load().pn_node_ref_pack(int(n), _lib.np_ptr(np.ascontiguousarray(off), ctypes.c_int64),
                                                    _lib.np_ptr(ref, ctypes.c_uint32)))
            self.d_node_ref = torch.from_numpy(ref.view(np.int32)).to(self.device)
        self.d_dis = self.d_adj_off = self.d_adj = self.d_radj_off = self.d_radj = None
        if hops == "dense":
            self.d_dis = torch.from_numpy(hops_dense(n, eu, ev, seq_len)).to(self.device)
        else:
            if seq_len > 8:
                raise ValueError("on-the-fly hop codes support path lengths up to 8")
            ao, aa = csr_build(n, eu, ev, reverse=False)
            ro, ra = csr_build(n, eu, ev, reverse=True)
            self.d_adj_off, self.d_adj = torch.from_numpy(ao).to(self.device), torch.from_numpy(aa).to(self.device)
            self.d_radj_off, self.d_radj = torch.from_numpy(ro).to(self.device), torch.from_numpy(ra).to(self.device)
        self.total = int(total)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ws = None

    @classmethod
    def from_edge_file(cls, path, seq_len, device="cuda", hops="auto"):
        n, u, v, p = read_edge_file(path)
        return cls(n, u, v, p, seq_len, device=device, hops=hops)

    def sample(self, W, seed, epoch_begin=0, epoch_count=1, node_begin=0, node_count=None,
               draw_source=DRAW_PHILOX, check=True, out=None, step_state=None, nodes=None):
        """-> ids int32 [epoch_count, node_count, W, L], codes uint8 [...] on the GPU.
        nodes (int32 device tensor, Philox only): sample the paths of these source nodes (in this order) instead of the
        range [node_begin, node_begin + node_count) -- e.g. the step's masked nodes; a walk's draws depend on (epoch,
        source node, walk index) only, so the paths are the ones a full-epoch sample holds for those nodes.
        step_state (pathnet_amd.StepState, Philox only): seed and epoch_begin are read from device memory when the
        kernel runs, so the call can be captured in a hipGraph and replayed."""
        lib = _lib.load()
        if nodes is not None:
            if nodes.dtype != torch.int32 or not nodes.is_cuda or not nodes.is_contiguous() or nodes.dim() != 1:
                raise ValueError("sample(nodes=...): contiguous 1-D int32 device tensor of node ids expected")
            node_begin, node_count = 0, int(nodes.numel())
        elif node_count is None:
            node_count = self.n - node_begin
        L = self.L
        shape = (epoch_count, node_count, W, L)
        if out is None:
            ids = torch.empty(shape, dtype=torch.int32, device=self.device)
            codes = torch.empty(shape, dtype=torch.uint8, device=self.device)
        else:
            ids, codes = out
            for t, dt in ((ids, torch.int32), (codes, torch.uint8)):
                if (tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device.type != "cuda" or
                        (self.device.index is not None and t.device != self.device)):
                    raise ValueError("sample(out=...): contiguous %s tensor of shape %s on %s expected, got %s %s on %s"
                                     % (dt, shape, self.device, t.dtype, tuple(t.shape), t.device))
        need = ctypes.c_int64(0)
        _lib.check(lib.pn_sample_workspace_bytes(W, L, draw_source, epoch_count, node_count, ctypes.byref(need)))
        dp = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
        tb = _lib.SamplerTables(self.n, self.total, self.d_off.data_ptr(), self.d_triples.data_ptr(), dp(self.d_dis),
                                dp(self.d_adj_off), dp(self.d_adj), dp(self.d_radj_off), dp(self.d_radj),
                                self.draws_per_step, dp(self.d_node_ref))
        dev = ids.device
        with torch.cuda.device(dev):        # the library launches on the current device
            if need.value and (self._ws is None or self._ws.numel() < need.value):
                self._ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
            if check:
                self._status.zero_()
            _lib.check(lib.pn_sample_paths(_lib.context(dev), ctypes.byref(tb), W, L, draw_source,
                                           seed & 0xFFFFFFFFFFFFFFFF, epoch_begin, epoch_count, node_begin, node_count,
                                           _lib.ptr(ids), _lib.ptr(codes), _lib.ptr(self._ws) if need.value else None,
                                           need.value, _lib.ptr(self._status),
                                           step_state.ptr() if step_state is not None else None,
                                           _lib.ptr(nodes) if nodes is not None else None, _lib.stream_ptr(dev)))
        if check and int(self._status.item()) != 0:
            # the reference prints this and exits (gen_merw.cpp:84-87)
            raise _lib.PnError(int(self._status.item()), "ERROR:: A.size() == 0 in Alias Table")
        return ids, codes


class UniformSampler(MerwSampler):
    """The uniform random-walk sampler of the "RW-PathNet" ablation (preprocess/gen.cpp, gen_epoch.cpp): same walker
    and hop codes, but the graph is the symmetrised pair list with self loops (gen.cpp:83-94) and a step is ONE draw,
    ``E[u][rand() % deg]`` (gen.cpp:113-114).  ``sample()`` is MerwSampler's; glibc-replay draws are bit-exact to the
    reference binary."""

    draws_per_step = 1

    def __init__(self, n, u, v, seq_len, device="cuda", hops="auto"):
        off, packed, src, nbr = build_uniform(n, u, v)
        self.host = dict(off=off, nbr=nbr)
        self._upload(n, seq_len, device, hops, off, packed, len(nbr), src, nbr)

    @classmethod
    def from_pair_file(cls, path, seq_len, device="cuda", hops="auto"):
        n, u, v = read_pair_file(path)
        return cls(n, u, v, seq_len, device=device, hops=hops)


def main(argv=None):
    """Drop-in for ./gen_merw and ./gen_epoch_merw (argv: <data_name> <path_num> <path_length>); with --uniform for
    ./gen and ./gen_epoch (the uniform random-walk sampler)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    per_epoch = "--per-epoch" in argv
    uniform = "--uniform" in argv        # gen.cpp / gen_epoch.cpp instead of gen_merw.cpp / gen_epoch_merw.cpp
    binary = "--binary" in argv          # also write the PNPATHS1 sidecar(s) next to the text file(s)
    opts = {"--seed": None, "--epochs": "1000", "--draw": "glibc", "--in": None, "--out-root": "./"}
    pos = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ("--per-epoch", "--binary", "--uniform"):
            i += 1
        elif a in opts:
            opts[a] = argv[i + 1]
            i += 2
        else:
            pos.append(a)
            i += 1
    if len(pos) != 3:
        print("ERROR: Incorrect number of parameters. ", file=sys.stderr)   # gen_merw.cpp:128-132
        return 0
    name, W, L = pos[0], int(pos[1]), int(pos[2])
    seed = int(opts["--seed"]) if opts["--seed"] is not None else int(time.time())  # srand(time(0)), :161
    epochs = int(opts["--epochs"])
    draw = DRAW_GLIBC_REPLAY if opts["--draw"] == "glibc" else DRAW_PHILOX
    root = opts["--out-root"]
    if uniform:
        # gen.cpp:50-68 reads <name>_nsl.in and writes <name>_<W>_<L>_nsl.txt; gen_epoch.cpp:50-56,:84-95 reads <name>.in
        # and writes <name>_<W>_<L>_<epoch>.txt
        edge = opts["--in"] or ("../edge_input/%s.in" % name if per_epoch else "../edge_input/%s_nsl.in" % name)
        whole = pathfile.whole_run_name(root, name, W, L, marker="nsl")
        if not per_epoch:
            print("File input: " + edge)                                     # gen.cpp:73-74
            print("File output: " + whole)
        n_, u_, v_ = read_pair_file(edge)
        smp = UniformSampler(n_, u_, v_, L)
        if per_epoch:
            print("%d %d" % (smp.n, len(u_)))                                # gen_epoch.cpp:64
        else:
            print(smp.n, file=sys.stderr)                                    # gen.cpp:81
    else:
        edge = opts["--in"] or "../edge_input/%s.in" % name
        smp = MerwSampler.from_edge_file(edge, L)
        print(name + ": " + str(smp.n), file=sys.stderr)                    # :164
        whole = pathfile.whole_run_name(root, name, W, L)
    chunk = max(1, min(epochs, (64 << 20) // max(1, smp.n * W * L * 5)))
    for e0 in range(0, epochs, chunk):
        ec = min(chunk, epochs - e0)
        try:
            ids, codes = smp.sample(W, seed, epoch_begin=e0, epoch_count=ec, draw_source=draw)
        except _lib.PnError as ex:
            if ex.code != _lib.PN_ERR_EMPTY_TABLE:
                raise
            print("ERROR:: A.size() == 0 in Alias Table", file=sys.stderr)      # gen_merw.cpp:84-87: message + exit(0)
            return 0
        ids, codes = ids.cpu().numpy(), codes.cpu().numpy()
        for k in range(ec):
            if per_epoch:
                fn = ("%s%s_%d_%d_%d.txt" % (root, name, W, L, e0 + k) if uniform       # gen_epoch.cpp:84-95: no marker
                      else pathfile.per_epoch_name(root, name, W, L, e0 + k))
                pathfile.write_paths(fn, ids[k], codes[k])
                if binary:
                    pathfile.write_paths_binary(fn[:-4] + ".bin", ids[k], codes[k])
            else:
                pathfile.write_paths(whole, ids[k], codes[k], append=(e0 + k) > 0)
        if binary and not per_epoch:
            if epochs > chunk:
                raise SystemExit("--binary without --per-epoch needs the whole run in one chunk; use --per-epoch")
            pathfile.write_paths_binary(whole[:-4] + ".bin", ids, codes)
    return 0


if __name__ == "__main__":
    sys.exit(main())
