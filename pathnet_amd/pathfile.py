"""Path-file reader/writer in the reference's exact text format.

One line per path: ``[v0, v1, ..., v_{L-1}, d0, ..., d_{L-1}]\\n`` (writer
/root/reference/preprocess/gen_merw.cpp:189-206; reader PathNet_run.py:418-423 / :325-334).  File names
follow PathNet_run.py:415-416 (whole run) and :320-321 (one file per epoch).
"""
import ctypes

import numpy as np

from . import _lib


def write_paths(path, ids, codes, append=False):
    """ids [..., L] int32, codes [..., L] uint8 (host arrays or CPU tensors) -> text file."""
    ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
    codes = np.ascontiguousarray(np.asarray(codes), dtype=np.uint8)
    L = ids.shape[-1]
    if codes.shape != ids.shape:
        raise ValueError("ids and codes must have the same shape")
    n = ids.size // L if L else 0
    _lib.check(_lib.load().pn_paths_write_text(str(path).encode(), _lib.np_ptr(ids, ctypes.c_int32),
                                               _lib.np_ptr(codes, ctypes.c_uint8), n, L, 1 if append else 0))


def read_paths(path, L):
    """-> (ids [npaths, L] int32, codes [npaths, L] uint8)."""
    lib = _lib.load()
    n = ctypes.c_int64(0)
    _lib.check(lib.pn_paths_read_text(str(path).encode(), L, None, None, 0, ctypes.byref(n)))
    ids = np.empty((n.value, L), dtype=np.int32)
    codes = np.empty((n.value, L), dtype=np.uint8)
    if n.value:
        _lib.check(lib.pn_paths_read_text(str(path).encode(), L, _lib.np_ptr(ids, ctypes.c_int32),
                                          _lib.np_ptr(codes, ctypes.c_uint8), n.value, ctypes.byref(n)))
    return ids, codes


def write_paths_binary(path, ids, codes):
    """Binary sidecar ("PNPATHS1": 32-byte header, int32 ids, uint8 codes) -- same content as the text file."""
    ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
    codes = np.ascontiguousarray(np.asarray(codes), dtype=np.uint8)
    L = ids.shape[-1]
    if codes.shape != ids.shape:
        raise ValueError("ids and codes must have the same shape")
    _lib.check(_lib.load().pn_paths_write_bin(str(path).encode(), _lib.np_ptr(ids, ctypes.c_int32),
                                              _lib.np_ptr(codes, ctypes.c_uint8), ids.size // L if L else 0, L))


def read_paths_binary(path):
    """-> (ids [npaths, L] int32, codes [npaths, L] uint8)."""
    lib = _lib.load()
    n, L = ctypes.c_int64(0), ctypes.c_int32(0)
    _lib.check(lib.pn_paths_read_bin(str(path).encode(), ctypes.byref(L), None, None, 0, ctypes.byref(n)))
    ids = np.empty((n.value, L.value), dtype=np.int32)
    codes = np.empty((n.value, L.value), dtype=np.uint8)
    if n.value:
        _lib.check(lib.pn_paths_read_bin(str(path).encode(), ctypes.byref(L), _lib.np_ptr(ids, ctypes.c_int32),
                                         _lib.np_ptr(codes, ctypes.c_uint8), n.value, ctypes.byref(n)))
    return ids, codes


def whole_run_name(root, name, W, L, marker="merw"):
    """PathNet_run.py:415-416: '{paths_root}{name}_{W}_{L}_{marker}.txt'"""
    return "%s%s_%d_%d_%s.txt" % (root, name, W, L, marker)


def per_epoch_name(root, name, W, L, epoch, marker="merw"):
    """PathNet_run.py:320-321: '{paths_root}{name}_{W}_{L}_{epoch}_{marker}.txt'"""
    return "%s%s_%d_%d_%d_%s.txt" % (root, name, W, L, epoch, marker)
