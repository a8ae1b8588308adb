#!/usr/bin/env python
"""Host-side set-up of the sampler (gen_merw.cpp:162-179: edge file -> per-node lists -> alias tables -> bfs/dis),
timed stage by stage on the CPU (no GPU needed).

    python tools/bench_sampler_setup.py [n_nodes=19717] [L=4] [dense|otf]
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import merw  # noqa: E402  (edge-file writer only)
from pathnet_amd import sampler  # noqa: E402
from tools.bench_sampler_large import er_graph  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 19717
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    mode = sys.argv[3] if len(sys.argv) > 3 else ("dense" if n <= 50000 else "otf")
    g = er_graph(n, 16, 0)
    d = tempfile.mkdtemp(prefix="pn_setup_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    f = os.path.join(d, "g.in")
    try:
        merw.write_edge_file(f, *g)
        t = [time.perf_counter()]
        n2, u, v, p = sampler.read_edge_file(f)
        t.append(time.perf_counter())
        sampler.build_alias(n2, u, v, p)
        t.append(time.perf_counter())
        if mode == "dense":
            sampler.hops_dense(n2, u, v, L)
        else:
            sampler.csr_build(n2, u, v)
            sampler.csr_build(n2, u, v, reverse=True)
        t.append(time.perf_counter())
        print("n=%d rows=%d L=%d file %.1f MB, PN_HOST_THREADS=%s" % (n2, len(u), L, os.path.getsize(f) / 1e6,
                                                                   os.environ.get("PN_HOST_THREADS", "(auto)")))
        print("edge file read  %.3f s (%.1f M rows/s)" % (t[1] - t[0], len(u) / (t[1] - t[0]) / 1e6))
        print("alias tables    %.3f s" % (t[2] - t[1]))
        print("%-15s %.3f s" % ("hop table" if mode == "dense" else "CSR fwd + rev", t[3] - t[2]))
    finally:
        if os.path.exists(f):
            os.remove(f)
        os.rmdir(d)


if __name__ == "__main__":
    main()
