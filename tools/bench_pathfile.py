#!/usr/bin/env python
"""Host-side path-file I/O rates: text writer, text reader, binary sidecar.

    python tools/bench_pathfile.py [npaths] [L]

Runs on the CPU only (no GPU needed): the text format is the reference's (`gen_merw.cpp:200-214` writer,
`PathNet_run.py:325-334` reader).  PN_HOST_THREADS=1 gives the single-thread figures.
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathnet_amd import pathfile  # noqa: E402


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    npaths = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 19717, (npaths, L), dtype=np.int32)
    codes = rng.integers(0, L, (npaths, L), dtype=np.uint8)
    d = tempfile.mkdtemp(prefix="pn_io_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    txt, binf = os.path.join(d, "p.txt"), os.path.join(d, "p.bin")
    try:
        t_end = time.perf_counter() + 2.0          # (this container hands a process its other cores only after ~1 s of load)
        while time.perf_counter() < t_end:
            pathfile.write_paths(txt, ids, codes)
        tw = best(lambda: pathfile.write_paths(txt, ids, codes))
        size = os.path.getsize(txt)
        tr = best(lambda: pathfile.read_paths(txt, L))
        i2, c2 = pathfile.read_paths(txt, L)
        assert (i2 == ids).all() and (c2 == codes).all()
        tbw = best(lambda: pathfile.write_paths_binary(binf, ids, codes))
        tbr = best(lambda: pathfile.read_paths_binary(binf))
        print("threads: PN_HOST_THREADS=%s, cores=%d" % (os.environ.get("PN_HOST_THREADS", "(auto)"), os.cpu_count()))
        print("%d paths, L=%d, text file %.1f MB" % (npaths, L, size / 1e6))
        print("text  write %7.1f M paths/s (%6.0f MB/s)" % (npaths / tw / 1e6, size / tw / 1e6))
        print("text  read  %7.1f M paths/s (%6.0f MB/s)" % (npaths / tr / 1e6, size / tr / 1e6))
        print("binary write %6.1f M paths/s, read %6.1f M paths/s" % (npaths / tbw / 1e6, npaths / tbr / 1e6))
    finally:
        for f in (txt, binf):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(d)


if __name__ == "__main__":
    main()
