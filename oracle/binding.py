"""ctypes binding of the oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  It wraps

* ``oracle/libpfac_oracle.so``   the plain-C restatement (``pfac_oracle.c``)
* ``oracle/_ref/libpfac_ref.so`` the reference's own CPU code, compiled unmodified
  (optional; present when ``make -C oracle ref`` ran in the build container)
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "libpfac_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libpfac_ref.so")


class _Int2(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int)]


class _Edge(C.Structure):
    _fields_ = [("next_state", C.c_int), ("ch", C.c_int)]


class _OraclePfac(C.Structure):
    _fields_ = [
        ("file", C.POINTER(C.c_ubyte)), ("file_size", C.c_long), ("num_patterns", C.c_int),
        ("sorted_off", C.POINTER(C.c_int)), ("sorted_id", C.POINTER(C.c_int)),
        ("pattern_len", C.POINTER(C.c_int)), ("pattern_off", C.POINTER(C.c_int)),
        ("max_pattern_len", C.c_int), ("initial_state", C.c_int), ("num_states", C.c_int),
        ("num_leaves", C.c_int),
        ("row", C.POINTER(C.POINTER(_Edge))), ("row_n", C.POINTER(C.c_int)), ("row_cap", C.POINTER(C.c_int)),
        ("rows_alloc", C.c_int),
        ("dense", C.POINTER(C.c_int)),
        ("hash_row", C.POINTER(_Int2)), ("hash_val", C.POINTER(_Int2)), ("hash_total", C.c_long),
        ("initial_row", C.POINTER(C.c_int)),
    ]


class _RefTrie(C.Structure):
    _fields_ = [
        ("num_patterns", C.c_int), ("num_states", C.c_int), ("initial_state", C.c_int),
        ("max_pattern_len", C.c_int), ("num_edges", C.c_int),
        ("edge_state", C.POINTER(C.c_int)), ("edge_ch", C.POINTER(C.c_int)), ("edge_next", C.POINTER(C.c_int)),
        ("pattern_len", C.POINTER(C.c_int)), ("sorted_id", C.POINTER(C.c_int)),
    ]


_olib: Optional[C.CDLL] = None
_rlib: Optional[C.CDLL] = None


def build(force: bool = False) -> None:
    """Compile the C restatement (and, when the reference tree is present, oracle/_ref)."""
    src = os.path.join(_HERE, "pfac_oracle.c")
    if force or not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpfac_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/PFAC/src") and (force or not os.path.exists(REF_LIB)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _oracle() -> C.CDLL:
    global _olib
    if _olib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        lib = C.CDLL(ORACLE_LIB)
        P = C.POINTER(_OraclePfac)
        lib.oracle_load.argtypes = [C.c_char_p, C.POINTER(P)]
        lib.oracle_free.argtypes = [P]
        lib.oracle_free.restype = None
        lib.oracle_build_dense.argtypes = [P]
        lib.oracle_build_hash.argtypes = [P]
        for name in ("oracle_match_dense", "oracle_match_hash"):
            getattr(lib, name).argtypes = [P, C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("oracle_match_dense_omp", "oracle_match_hash_omp"):
            getattr(lib, name).argtypes = [P, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        lib.oracle_dump_table_to_path.argtypes = [P, C.c_char_p]
        lib.oracle_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.oracle_reduce.restype = C.c_long
        lib.oracle_digest.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        lib.oracle_digest.restype = None
        _olib = lib
    return _olib


def have_reference() -> bool:
    return os.path.exists(REF_LIB)


def _ref() -> C.CDLL:
    global _rlib
    if _rlib is None:
        lib = C.CDLL(REF_LIB)
        lib.ref_build.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_RefTrie))]
        lib.ref_free.argtypes = [C.POINTER(_RefTrie)]
        lib.ref_free.restype = None
        lib.ref_match_dense.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.ref_match_hash.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _rlib = lib
    return _rlib


class OracleError(RuntimeError):
    def __init__(self, status: int, where: str):
        super().__init__(f"{where}: oracle status {status}")
        self.status = status


class Oracle:
    """A pattern set compiled by the C restatement."""

    def __init__(self, pattern_file: str, dense: bool = True, hashed: bool = True):
        lib = _oracle()
        self._p = C.POINTER(_OraclePfac)()
        st = lib.oracle_load(os.fsencode(pattern_file), C.byref(self._p))
        if st != 0:
            raise OracleError(st, "oracle_load")
        if dense:
            st = lib.oracle_build_dense(self._p)
            if st != 0:
                raise OracleError(st, "oracle_build_dense")
        if hashed:
            st = lib.oracle_build_hash(self._p)
            if st != 0:
                raise OracleError(st, "oracle_build_hash")

    def close(self):
        if self._p:
            _oracle().oracle_free(self._p)
            self._p = C.POINTER(_OraclePfac)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # facts
    @property
    def c(self):
        return self._p.contents

    num_patterns = property(lambda s: s.c.num_patterns)
    num_states = property(lambda s: s.c.num_states)
    initial_state = property(lambda s: s.c.initial_state)
    max_pattern_len = property(lambda s: s.c.max_pattern_len)
    num_leaves = property(lambda s: s.c.num_leaves)
    hash_total = property(lambda s: s.c.hash_total)

    def edges(self):
        """[(state, ch, next)] in state order, insertion order within a state."""
        c = self.c
        out = []
        for s in range(c.num_states):
            for j in range(c.row_n[s]):
                e = c.row[s][j]
                out.append((s, e.ch, e.next_state))
        return out

    def dense_table(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.dense, shape=(self.c.num_states * 256,)).copy()

    def hash_row(self) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(self.c.hash_row, C.POINTER(C.c_int)), shape=(self.c.num_states * 2,)).copy()

    def hash_val(self) -> np.ndarray:
        n = int(self.c.hash_total)
        if n == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(C.cast(self.c.hash_val, C.POINTER(C.c_int)), shape=(n * 2,)).copy()

    def initial_row(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.initial_row, shape=(256,)).copy()

    # matching
    def match(self, data, hashed: bool = False, omp: bool = False, threads: int = 0) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.full(data.size, -9, dtype=np.int32)
        lib = _oracle()
        if omp:
            fn = lib.oracle_match_hash_omp if hashed else lib.oracle_match_dense_omp
            st = fn(self._p, data.ctypes.data, data.size, out.ctypes.data, threads)
        else:
            fn = lib.oracle_match_hash if hashed else lib.oracle_match_dense
            st = fn(self._p, data.ctypes.data, data.size, out.ctypes.data)
        if st != 0:
            raise OracleError(st, "oracle_match")
        return out

    def dump_table(self, path: str) -> None:
        st = _oracle().oracle_dump_table_to_path(self._p, os.fsencode(path))
        if st != 0:
            raise OracleError(st, "oracle_dump_table")


def omp_max_threads() -> int:
    return int(_oracle().oracle_omp_max_threads())


def reduce(result: np.ndarray):
    """(ids, positions) of the non-zero results, ascending position (ref PFAC.cpp:1055-1066)."""
    r = np.ascontiguousarray(result, dtype=np.int32).copy()
    pos = np.zeros(r.size, dtype=np.int32)
    k = _oracle().oracle_reduce(r.ctypes.data, pos.ctypes.data, r.size)
    return r[:k].copy(), pos[:k].copy()


def digest(result: np.ndarray):
    r = np.ascontiguousarray(result, dtype=np.int32)
    f, c = C.c_ulonglong(), C.c_ulonglong()
    _oracle().oracle_digest(r.ctypes.data, r.size, C.byref(f), C.byref(c))
    return int(f.value), int(c.value)


class Reference:
    """The reference's own parser / trie builder / CPU matchers (oracle/_ref)."""

    def __init__(self, pattern_file: str):
        self._t = C.POINTER(_RefTrie)()
        st = _ref().ref_build(os.fsencode(pattern_file), C.byref(self._t))
        if st != 0:
            raise OracleError(st, "ref_build")

    def close(self):
        if self._t:
            _ref().ref_free(self._t)
            self._t = C.POINTER(_RefTrie)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def t(self):
        return self._t.contents

    def edges(self):
        t = self.t
        return [(t.edge_state[i], t.edge_ch[i], t.edge_next[i]) for i in range(t.num_edges)]

    @staticmethod
    def match_dense(data, dense_table, num_final, initial, omp=False) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        tab = np.ascontiguousarray(dense_table, dtype=np.int32)
        out = np.full(data.size, -9, dtype=np.int32)
        st = _ref().ref_match_dense(data.ctypes.data, data.size, tab.ctypes.data, num_final, initial,
                                    out.ctypes.data, 1 if omp else 0)
        if st != 0:
            raise OracleError(st, "ref_match_dense")
        return out

    @staticmethod
    def match_hash(data, hash_row, hash_val, num_final, initial, omp=False) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        row = np.ascontiguousarray(hash_row, dtype=np.int32)
        val = np.ascontiguousarray(hash_val, dtype=np.int32)
        out = np.full(data.size, -9, dtype=np.int32)
        st = _ref().ref_match_hash(data.ctypes.data, data.size, row.ctypes.data, val.ctypes.data, num_final,
                                   initial, out.ctypes.data, 1 if omp else 0)
        if st != 0:
            raise OracleError(st, "ref_match_hash")
        return out
