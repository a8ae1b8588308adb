"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pathnet_hip.h declares,
and its host-only entry points (no GPU needed) agree with the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, golden_files
from oracle import merw
from pathnet_amd import _lib, pathfile, sampler


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "pathnet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    txt = open(os.path.join(ROOT, "include", "pathnet_hip.h")).read()
    declared = int(re.search(r"#define PN_ABI_VERSION (\d+)", txt).group(1))
    assert lib.pn_abi_version() == declared == _lib.ABI_VERSION


@pytest.mark.parametrize("name", golden_files("sampler_*.npz"))
def test_host_alias_tables_and_thresholds_match_oracle(name):
    g = golden(name)
    n = int(g["n"])
    off, A, B, S, thr = sampler.build_alias(n, g["u"], g["v"], g["p"])
    o2, A2, B2, S2 = merw.alias_build(n, g["u"], g["v"], g["p"])
    assert (off == o2).all() and (A == A2).all() and (B == B2).all()
    assert (S == S2).all()          # fp64 bit-identical
    # integer threshold == the reference's fp64 compare, on random draws and at the boundaries
    rng = np.random.default_rng(0)
    for r in (rng.integers(0, 2 ** 31 - 1, len(S)), np.minimum(thr.astype(np.int64), 2 ** 31 - 1),
              np.maximum(thr.astype(np.int64) - 1, 0), np.full(len(S), 2 ** 31 - 1), np.zeros(len(S), np.int64)):
        assert ((1.0 * r / 2147483647 > S) == (r >= thr.astype(np.int64))).all()


def test_host_hop_table_matches_reference_bfs_on_reachable_nodes():
    g = golden("sampler_synthetic97_12_5.npz")
    n, L = int(g["n"]), int(g["L"])
    d = sampler.hops_dense(n, g["u"], g["v"], L)
    d2 = merw.bfs_dense(n, g["u"], g["v"], L)
    within = (d2 >= 1) & (d2 <= L)          # what a walk of L nodes can reach
    assert (d[within] == d2[within]).all() and (d[~within] == 0).all()


def test_host_glibc_jump_algebra():
    for seed, first in ((1, 0), (7, 1), (123, 30), (99, 31), (5, 1 << 20), (2 ** 32 - 1, 987654321)):
        assert (sampler.glibc_draws(seed, first, 200) == merw.glibc_stream(seed, 200, skip=first)).all()


def test_path_file_round_trip_and_format(tmp_path):
    g = golden("sampler_cornell_7_6.npz")
    ids, codes = g["ids"][0], g["codes"][0]
    f = os.path.join(tmp_path, "p.txt")
    pathfile.write_paths(f, ids, codes)
    assert open(f, "rb").read() == merw.format_text(ids, codes)
    i2, c2 = pathfile.read_paths(f, int(g["L"]))
    assert (i2 == ids.reshape(-1, ids.shape[-1])).all() and (c2 == codes.reshape(-1, ids.shape[-1])).all()
    # the reference reader: list(map(int, line[1:-2].split(",")))  (PathNet_run.py:327)
    line = open(f).readline()
    info = list(map(int, line[1:-2].split(",")))
    assert info[:6] == ids.reshape(-1, 6)[0].tolist() and info[6:] == codes.reshape(-1, 6)[0].tolist()
    pathfile.write_paths(f, ids, codes, append=True)
    assert pathfile.read_paths(f, 6)[0].shape[0] == 2 * ids.size // 6


def test_path_file_errors(tmp_path):
    f = os.path.join(tmp_path, "bad.txt")
    open(f, "w").write("[1, 2, 3, 0, 1\n")
    with pytest.raises(_lib.PnError) as e:
        pathfile.read_paths(f, 2)
    assert e.value.code == _lib.PN_ERR_FORMAT
    with pytest.raises(_lib.PnError) as e:
        pathfile.read_paths(os.path.join(tmp_path, "missing.txt"), 2)
    assert e.value.code == _lib.PN_ERR_IO
    open(f, "w").write("")
    assert pathfile.read_paths(f, 4)[0].shape == (0, 4)


def test_path_file_threaded_equals_single_thread(tmp_path, monkeypatch):
    """The text writer / reader split the work over host threads (ranges of paths / of file bytes): same bytes,
    same arrays, same error line as one thread, including negative ids, ragged digit counts and append mode."""
    rng = np.random.default_rng(5)
    n, L = 150_001, 5
    ids = rng.integers(-3, 10 ** rng.integers(1, 10, (n, 1)), (n, L)).astype(np.int32)
    codes = rng.integers(0, 256, (n, L)).astype(np.uint8)
    files = {}
    for threads in ("1", "7"):
        monkeypatch.setenv("PN_HOST_THREADS", threads)
        f = os.path.join(tmp_path, "p%s.txt" % threads)
        pathfile.write_paths(f, ids[:1000], codes[:1000])
        pathfile.write_paths(f, ids[1000:], codes[1000:], append=True)
        files[threads] = f
    blob = open(files["1"], "rb").read()
    assert blob == open(files["7"], "rb").read()
    assert blob[:2000] == merw.format_text(ids[None, :1000], codes[None, :1000])[:2000]
    for threads in ("1", "7"):
        monkeypatch.setenv("PN_HOST_THREADS", threads)
        i2, c2 = pathfile.read_paths(files["7"], L)
        assert (i2 == ids).all() and (c2 == codes).all()
    # a malformed line deep inside the file: first error in file order, numbered like a sequential scan
    lines = blob.split(b"\n")
    bad_at = 123_456
    lines[bad_at] = lines[bad_at].replace(b", ", b"; ", 1)
    lines[bad_at + 5000] = b"oops"
    fbad = os.path.join(tmp_path, "bad.txt")
    open(fbad, "wb").write(b"\n".join(lines))
    msgs = []
    for threads in ("1", "7"):
        monkeypatch.setenv("PN_HOST_THREADS", threads)
        with pytest.raises(_lib.PnError) as e:
            pathfile.read_paths(fbad, L)
        assert e.value.code == _lib.PN_ERR_FORMAT
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "line %d " % bad_at in msgs[0]
    # no final line end: the reference reader's line[1:-2] slice would eat a digit; refused, with the line number
    open(fbad, "wb").write(blob[:-1])
    with pytest.raises(_lib.PnError, match="line %d does not end" % (n - 1)):
        pathfile.read_paths(fbad, L)


def test_host_setup_threaded_equals_single_thread(tmp_path, monkeypatch):
    """Edge-file reader, alias tables, CSR lists and the dense hop table run on all host threads; every output is
    identical to the single-thread run (and the reader to the values written)."""
    rng = np.random.default_rng(11)
    n, m_und = 6000, 60000
    a, b = rng.integers(0, n, m_und), rng.integers(0, n, m_und)
    u = np.concatenate([a, b, np.arange(n)]).astype(np.int32)
    v = np.concatenate([b, a, np.arange(n)]).astype(np.int32)
    order = rng.permutation(len(u))                      # file order is not sorted by source
    u, v = u[order], v[order]
    deg = np.bincount(u, minlength=n)
    p = rng.random(len(u)) * 2.0 / deg[u]                # unnormalised masses: heavy and light entries
    f = os.path.join(tmp_path, "g.in")
    merw.write_edge_file(f, n, u, v, p)
    assert os.path.getsize(f) > 2 << 20                  # several byte ranges
    out = {}
    for threads in ("1", "6"):
        monkeypatch.setenv("PN_HOST_THREADS", threads)
        n2, u2, v2, p2 = sampler.read_edge_file(f)
        assert n2 == n and (u2 == u).all() and (v2 == v).all() and (p2 == p).all()
        out[threads] = (sampler.build_alias(n, u, v, p), sampler.csr_build(n, u, v),
                        sampler.csr_build(n, u, v, reverse=True), sampler.hops_dense(n, u, v, 4))
    flat = lambda t: [np.asarray(x) for part in t for x in (part if isinstance(part, tuple) else (part,))]
    for x, y in zip(flat(out["1"]), flat(out["6"])):
        assert x.shape == y.shape and (x == y).all()
    # scanf-style oddities fall back to the sequential reader: rows spread over lines, "+" signs, trailing tokens
    g = os.path.join(tmp_path, "odd.in")
    open(g, "w").write("3 2\n0\n1 0.25 1\t+2\n5e-1 77 extra\n")
    n3, u3, v3, p3 = sampler.read_edge_file(g)
    assert n3 == 3 and u3.tolist() == [0, 1] and v3.tolist() == [1, 2] and p3.tolist() == [0.25, 0.5]
    open(g, "w").write("3 2\n0 1 0.25\n1 2\n")
    with pytest.raises(_lib.PnError, match="row 1 truncated"):
        sampler.read_edge_file(g)


def test_host_out_of_memory_is_an_error_code_not_an_abort():
    """No exception crosses the C boundary: with the address space capped, a table build that cannot allocate returns
    PN_ERR_NOMEM and the process lives on (worker threads hand their exceptions to the calling thread the same way)."""
    import subprocess
    import sys
    code = r"""
import resource, sys
sys.path.insert(0, %r)
import numpy as np
from pathnet_amd import _lib, sampler
_lib.load()
n, m = 2000, 40_000_000
u = np.zeros(m, np.int32); v = np.zeros(m, np.int32)
vm = [int(l.split()[1]) * 1024 for l in open("/proc/self/status") if l.startswith("VmSize")][0]
resource.setrlimit(resource.RLIMIT_AS, (vm + (64 << 20), vm + (64 << 20)))     # the 160 MB bucket array cannot be had
try:
    sampler.csr_build(n, u, v)
    print("NO ERROR")
except _lib.PnError as e:
    print("CODE", e.code, str(e))
except MemoryError:
    print("PYTHON MemoryError")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    out = r.stdout.strip().splitlines()[-1]
    assert out.startswith("CODE %d" % _lib.PN_ERR_NOMEM) and "out of host memory" in out, out


def test_edge_file_reader(tmp_path):
    g = golden("sampler_synthetic97_12_5.npz")
    f = os.path.join(tmp_path, "g.in")
    merw.write_edge_file(f, int(g["n"]), g["u"], g["v"], g["p"])
    n, u, v, p = sampler.read_edge_file(f)
    assert n == int(g["n"]) and (u == g["u"]).all() and (v == g["v"]).all() and (p == g["p"]).all()
    with pytest.raises(_lib.PnError):
        sampler.read_edge_file(os.path.join(tmp_path, "nope.in"))


def test_workspace_query_and_shape_validation():
    from pathnet_amd import modules
    assert modules.workspace_bytes("homo", 100, 16, 128, 3, 10, 40, 4) > 0
    with pytest.raises(_lib.PnError):
        modules.workspace_bytes("homo", 100, 16, 100, 3, 10, 40, 4)      # H not supported
    with pytest.raises(_lib.PnError):
        modules.workspace_bytes("pagg", 100, 16, 64, 3, 10, 40, 5)        # PAGG has 4 distance layers
    assert modules.workspace_bytes("homo", 100, 16, 256, 3, 10, 832, 4) > 0
    with pytest.raises(_lib.PnError, match="LDS budget"):
        modules.workspace_bytes("homo", 100, 16, 256, 3, 10, 833, 4)      # pooling kernels' LDS


def test_sampler_cli_wrong_argc(capsys):
    assert sampler.main(["only", "two"]) == 0      # reference prints to stderr and returns 0 (gen_merw.cpp:128-132)
    assert "Incorrect number of parameters" in capsys.readouterr().err


def test_csr_build_sorted_unique_both_directions():
    g = golden("sampler_synthetic97_12_5.npz")
    n, u, v = int(g["n"]), g["u"], g["v"]
    for rev in (False, True):
        off, adj = sampler.csr_build(n, u, v, reverse=rev)
        src, dst = (v, u) if rev else (u, v)
        for node in (0, 3, 50, n - 1):
            want = np.unique(dst[src == node])
            assert (adj[off[node]:off[node + 1]] == want).all()
        assert off[-1] == len(adj)


def test_no_kernel_spills_registers():
    """The recurrent kernels prefetch weight fragments with asm loads whose destination registers the compiler must
    not spill between the load and its wait (cdna_hip_programming.md §5.7): require zero scratch in every kernel."""
    import subprocess
    src = os.path.join(ROOT, "pathnet_amd", "csrc")
    for f in ("pn_pagg.hip", "pn_sampler.hip", "pn_train.hip", "pn_merw.hip"):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(src, f), "-o", "/dev/null"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        sizes = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)
        assert sizes and all(int(x) == 0 for x in sizes), (f, sizes)


def test_binary_path_file_is_lossless_and_rejects_garbage(tmp_path):
    g = golden("sampler_cornell_40_4.npz")
    ids, codes = g["ids"][0], g["codes"][0]
    fb, ft = os.path.join(tmp_path, "p.bin"), os.path.join(tmp_path, "p.txt")
    pathfile.write_paths_binary(fb, ids, codes)
    i2, c2 = pathfile.read_paths_binary(fb)
    assert (i2 == ids.reshape(-1, 4)).all() and (c2 == codes.reshape(-1, 4)).all()
    pathfile.write_paths(ft, i2, c2)                       # binary -> text == reference text
    assert open(ft, "rb").read() == merw.format_text(ids, codes)
    assert os.path.getsize(fb) == 32 + ids.size * 5
    open(fb, "wb").write(b"not a path file at all, definitely not")
    with pytest.raises(_lib.PnError) as e:
        pathfile.read_paths_binary(fb)
    assert e.value.code == _lib.PN_ERR_FORMAT


def test_node_ref_pack():
    import ctypes
    lib = _lib.load()
    off = np.array([0, 3, 3, 10, 4000000000], np.int64)
    ref = np.zeros(8, np.uint32)
    _lib.check(lib.pn_node_ref_pack(4, _lib.np_ptr(off, ctypes.c_int64), _lib.np_ptr(ref, ctypes.c_uint32)))
    assert ref.reshape(4, 2).tolist() == [[0, 3], [3, 0], [3, 7], [10, 3999999990]]
    off[-1] = 2 ** 32
    assert lib.pn_node_ref_pack(4, _lib.np_ptr(off, ctypes.c_int64), _lib.np_ptr(ref, ctypes.c_uint32)) == _lib.PN_ERR_ARG
    bad = np.array([0, 5, 3], np.int64)
    assert lib.pn_node_ref_pack(2, _lib.np_ptr(bad, ctypes.c_int64), _lib.np_ptr(ref, ctypes.c_uint32)) == _lib.PN_ERR_ARG


SAN_SCRIPT = r'''
import ctypes, os, sys, tempfile, types
import numpy as np
root, san = sys.argv[1], sys.argv[2]
# the host helpers of the package without importing the package (torch and the HIP runtime stay out of the sanitized process)
pkg = types.ModuleType("pathnet_amd"); pkg.__path__ = [os.path.join(root, "pathnet_amd")]; sys.modules["pathnet_amd"] = pkg
sys.modules["torch"] = types.ModuleType("torch")
from pathnet_amd import _lib
_lib.LIB_PATH = san
probe = ctypes.CDLL(san)
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(probe, k)}
assert {"pn_edges_read_text", "pn_alias_build", "pn_hops_dense", "pn_csr_build", "pn_paths_write_text", "pn_paths_read_text",
        "pn_paths_write_bin", "pn_paths_read_bin", "pn_pairs_read_text", "pn_uniform_build", "pn_glibc_draws",
        "pn_alias_pack", "pn_node_ref_pack"} <= set(_lib.SIGNATURES)
from pathnet_amd import sampler, pathfile
sys.path.insert(0, root)
from oracle import merw
g = dict(np.load(os.path.join(root, "tests", "golden", "sampler_citeseer_10_4.npz")))     # negative / > 1 "probabilities"
n, u, v, p = int(g["n"]), g["u"], g["v"], g["p"]
d = tempfile.mkdtemp()
for threads in ("1", "5"):
    os.environ["PN_HOST_THREADS"] = threads
    path = os.path.join(d, "e.in")
    merw.write_edge_file(path, n, u, v, p)
    n2, u2, v2, p2 = sampler.read_edge_file(path)
    assert n2 == n and (u2 == u).all() and (v2 == v).all() and (p2 == p).all()
    off, A, B, S, thr = sampler.build_alias(n, u, v, p)
    oo, oA, oB, oS = merw.alias_build(n, u, v, p)
    assert (off == oo).all() and (A == oA).all() and (B == oB).all() and (S == oS).all()
    packed = np.empty(max(len(A), 1) * 4, np.int32)
    _lib.check(_lib.load().pn_alias_pack(len(A), _lib.np_ptr(A, ctypes.c_int32), _lib.np_ptr(B, ctypes.c_int32),
                                         _lib.np_ptr(thr, ctypes.c_uint32), _lib.np_ptr(packed, ctypes.c_int32)))
    ref = np.empty(2 * n, np.uint32)
    _lib.check(_lib.load().pn_node_ref_pack(n, _lib.np_ptr(np.ascontiguousarray(off), ctypes.c_int64), _lib.np_ptr(ref, ctypes.c_uint32)))
    dis, d2 = sampler.hops_dense(n, u, v, 4), merw.bfs_dense(n, u, v, 4)
    within = (d2 >= 1) & (d2 <= 4)          # what a walk of 4 nodes can reach (the reference's bfs labels one ring more)
    assert (dis[within] == d2[within]).all() and (dis[~within] == 0).all()
    for rev in (False, True):
        o, a = sampler.csr_build(n, u, v, reverse=rev)
        assert o[-1] == len(a) and (np.diff(o) >= 0).all()
    pairs = os.path.join(d, "p.in")
    merw.write_pair_file(pairs, n, u[:500], v[:500])
    n3, u3, v3 = sampler.read_pair_file(pairs)
    sampler.build_uniform(n3, u3, v3)
    assert (sampler.glibc_draws(7, 1000, 257) == merw.glibc_stream(7, 257, skip=1000)).all()
    ids, codes = g["ids"].reshape(-1, 4), g["codes"].reshape(-1, 4)
    t = os.path.join(d, "paths.txt")
    pathfile.write_paths(t, ids, codes)
    ri, rc = pathfile.read_paths(t, 4)
    assert (ri == ids).all() and (rc == codes).all()
    pathfile.write_paths_binary(t + ".bin", ids, codes)
    bi, bc = pathfile.read_paths_binary(t + ".bin")
    assert (bi == ids).all() and (bc == codes).all()
    # error paths: malformed lines, truncated binary, missing file
    open(t, "a").write("[1, 2, x, 4, 0, 1, 2, 3]\n")
    for bad in (lambda: pathfile.read_paths(t, 4), lambda: pathfile.read_paths(os.path.join(d, "nope"), 4)):
        try:
            bad()
            raise SystemExit("expected an error")
        except _lib.PnError:
            pass
    open(t + ".bin", "r+b").truncate(40)
    try:
        pathfile.read_paths_binary(t + ".bin")
        raise SystemExit("expected an error")
    except _lib.PnError:
        pass
print("SAN_OK")
'''


def test_host_code_under_asan_and_ubsan(tmp_path):
    """SURVEY.md section 5 ("race detection": an ASan / UBSan build of the host C++ is cheap): pn_host.cpp -- edge / pair file
    parsers (threaded and not), alias build, dense hop table, CSR lists, glibc stream, path-file writers and readers with
    their error paths -- compiled with -fsanitize=address,undefined (make sanitize) and run in a child process."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "pathnet_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "sanitize"], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", SAN_SCRIPT, ROOT, os.path.join(csrc, "_obj", "libpn_host_san.so")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SAN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _tile_ranges(P, tg, small_first):
    """the path ranges seq_tile_of (pn_seqh.hip) gives the workgroups of a launch, restated"""
    n_big, n_small, rows, blocks = tg
    out = []
    for j in range(blocks):
        if rows == 0:
            q0, r = 32 * j, min(32, P - 32 * j)
        elif small_first:
            q0, r = (j * rows, min(rows, P - j * rows)) if j < n_small else (
                n_small * rows + (j - n_small) * 32, min(32, P - (n_small * rows + (j - n_small) * 32)))
        else:
            q0, r = (32 * j, 32) if j < n_big else (
                32 * n_big + (j - n_big) * rows, min(rows, P - (32 * n_big + (j - n_big) * rows)))
        out.append((q0, r))
    return out


def test_remainder_round_tiling_covers_every_path_once():
    """The fp16 recurrent launches cut their remainder round into tiles of 8 / 16 / 24 paths, one per CU (pn_seqh.hip "tile
    geometry"): whatever the split, the tiles are disjoint, in order, non-empty and cover [0, P)."""
    import ctypes
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.pn_debug_seq_tiling.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int32)]
    seen_small = set()
    for P in (1, 7, 31, 32, 33, 679, 3480, 8192, 16384 + 5, 24576, 51960, 52000, 378560, 1234567):
        for slots, cus in ((512, 256), (768, 256), (256, 256), (12, 4)):
            for mode in (0, 1, 8, 16, 24):
                for small_first in (0, 1):
                    out = (ctypes.c_int32 * 4)()
                    assert lib.pn_debug_seq_tiling(P, slots, cus, mode, small_first, out) == 0
                    tg = tuple(out)
                    tiles = _tile_ranges(P, tg, small_first)
                    assert len(tiles) == tg[3] and tg[3] == tg[0] + tg[1] if tg[2] else tg[3] == (P + 31) // 32
                    pos = 0
                    for q0, r in tiles:
                        assert q0 == pos and 1 <= r <= 32, (P, slots, mode, small_first, tg)
                        pos += r
                    assert pos == P
                    if mode == 0:
                        assert tg[2] == 0
                    if tg[2]:
                        seen_small.add(tg[2])
                        assert tg[2] in (8, 16, 24) and tg[1] >= 1
                        if mode == 1:
                            assert tg[1] <= cus          # one small tile per CU at most
    assert seen_small == {8, 16, 24}
    # the headline shape: 1624 tiles on 512 slots -> three full rounds + 88 tiles; cut into 16-path tiles on 176 CUs
    out = (ctypes.c_int32 * 4)()
    lib.pn_debug_seq_tiling(51960, 512, 256, 1, 1, out)
    assert tuple(out) == (1536, 176, 16, 1712)
    lib.pn_debug_seq_tiling(51960, 768, 256, 1, 0, out)
    assert tuple(out) == (1536, 176, 16, 1712)

