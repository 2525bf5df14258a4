"""Deterministic synthetic workloads for the BASELINE.json configurations.

Bench / test infrastructure only (never on the match path).  Every generator
is a pure function of its seed; pattern sets honour the constraints of the
reference's pattern-file format: patterns are
unique, contain no 0x0A byte, the file ends with a newline and has no blank
lines.

    C1  README example                      example_patterns() / example_input()
    C2  1 000 random patterns, len 4..32    random_patterns(); random_bytes()
    C3  ~30 k Snort-style patterns          snort_patterns(); http_stream()
    C4  C3 patterns over N x 1 GiB slices   http_stream(seed + slice)
    C5  adversarial near-miss               adversarial_patterns(); adversarial_stream()
    C6  the C5 stream over C3's set + C5's  snort_patterns() + the 1 000 shared-prefix patterns of C5: the worst input on
        shared-prefix patterns              the FULL set (what PFAC_hash_draft.pdf Table 5 measures); not a BASELINE config
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

SEED_C2_PATTERNS = 0x5046414301
SEED_C2_INPUT = 0x5046414302
SEED_C3_PATTERNS = 0x5046414303
SEED_C3_INPUT = 0x5046414304
SEED_C5 = 0x5046414305

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpfac_workload.so")
_lib = None


def build_library(force: bool = False) -> str:
    src = os.path.join(_HERE, "gen.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _LIB_PATH, src])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build_library()
        lib = C.CDLL(_LIB_PATH)
        lib.wl_fill_random.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
        lib.wl_fill_random.restype = None
        lib.wl_fill_from_pool.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                          C.c_uint32, C.c_void_p, C.c_uint64]
        lib.wl_fill_from_pool.restype = None
        lib.wl_fnv1a.argtypes = [C.c_void_p, C.c_uint64]
        lib.wl_fnv1a.restype = C.c_uint64
        lib.wl_fnv1a_sparse_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        lib.wl_fnv1a_sparse_i32.restype = C.c_uint64
        _lib = lib
    return _lib


def fnv1a(data: np.ndarray) -> int:
    data = np.ascontiguousarray(data).view(np.uint8)
    return int(_load().wl_fnv1a(data.ctypes.data, data.size))


def fnv1a_sparse_i32(positions, ids, n: int) -> int:
    """FNV-1a-64 of the int32 vector of n elements that is `ids` at `positions` (ascending) and 0 elsewhere."""
    pos = np.ascontiguousarray(positions, dtype=np.int64)
    val = np.ascontiguousarray(ids, dtype=np.int32)
    assert pos.size == val.size and (pos.size == 0 or (pos[0] >= 0 and pos[-1] < n and np.all(np.diff(pos) > 0)))
    return int(_load().wl_fnv1a_sparse_i32(pos.ctypes.data, val.ctypes.data, pos.size, n))


# ----------------------------------------------------------------------------- patterns

def write_pattern_file(path: str, patterns: Sequence[bytes]) -> str:
    """One pattern per line, trailing newline (reference format, PFAC_reorder_Table.cpp:166-193)."""
    seen = set()
    with open(path, "wb") as f:
        for p in patterns:
            assert p and b"\n" not in p, "patterns are non-empty and contain no newline"
            assert p not in seen, "patterns must be unique"
            seen.add(p)
            f.write(p + b"\n")
    return path


def example_patterns() -> List[bytes]:
    """C1: the reference's PFAC/test/pattern/example_pattern (README.md:13-21)."""
    return [b"AB", b"ABG", b"BEDE", b"ED"]


def example_input() -> bytes:
    """C1: PFAC/test/data/example_input, 10 bytes including the trailing newline."""
    return b"ABEDEDABG\n"


def example2_patterns() -> List[bytes]:
    """PFAC/test/pattern/example_pattern2 (PFAC_hash_draft.pdf Fig. 1)."""
    return [b"s", b"h", b"he", b"she", b"hers", b"her", b"his", b"iis", b"is", b"ii"]


def example2_input() -> bytes:
    return b"sheshershisiis\n"


def random_patterns(count: int = 1000, min_len: int = 4, max_len: int = 32,
                    seed: int = SEED_C2_PATTERNS) -> List[bytes]:
    """C2: unique patterns, length uniform in [min_len, max_len], bytes uniform over 0..255 minus 0x0A."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out, seen = [], set()
    while len(out) < count:
        n = int(rng.integers(min_len, max_len + 1))
        b = rng.integers(0, 255, size=n, dtype=np.uint8)     # 0..254
        b = np.where(b >= 0x0A, b + 1, b).astype(np.uint8)   # skip 0x0A
        p = b.tobytes()
        if p not in seen:
            seen.add(p)
            out.append(p)
    return out


HTTP_KEYWORDS = [
    b"GET ", b"POST ", b"HEAD ", b"PUT ", b"Host: ", b"User-Agent: ", b"Accept: ", b"Cookie: ",
    b"Content-Type: ", b"Content-Length: ", b"Referer: ", b"Authorization: ", b"/cgi-bin/", b"cmd.exe",
    b"/etc/passwd", b"<script>", b"../", b"%00", b"/bin/sh", b"SELECT ", b"UNION ", b"/admin/",
    b"/wp-content/", b".php?", b"passwd=", b"HTTP/1.1", b"Set-Cookie: ", b"Location: ", b"eval(", b"%2e%2e/",
]
URL_SAFE = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-._~/?=&%+", dtype=np.uint8)
ALNUM = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)


def snort_patterns(count: int = 30000, seed: int = SEED_C3_PATTERNS) -> List[bytes]:
    """C3: Snort-style set -- 60 % keyword + 2..24 URL-safe chars, 30 % 3..40 alnum/URL chars,
    10 % 4..60 arbitrary binary bytes (no 0x0A); every length in 3..60 (< 512: halo-legal in
    the reference, PFAC_kernel.cu:108)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out, seen = [], set()
    while len(out) < count:
        u = rng.random()
        if u < 0.6:
            kw = HTTP_KEYWORDS[int(rng.integers(0, len(HTTP_KEYWORDS)))]
            tail = URL_SAFE[rng.integers(0, URL_SAFE.size, size=int(rng.integers(2, 25)))].tobytes()
            p = (kw + tail)[:60]
        elif u < 0.9:
            alpha = URL_SAFE if rng.random() < 0.5 else ALNUM
            p = alpha[rng.integers(0, alpha.size, size=int(rng.integers(3, 41)))].tobytes()
        else:
            b = rng.integers(0, 255, size=int(rng.integers(4, 61)), dtype=np.uint8)
            p = np.where(b >= 0x0A, b + 1, b).astype(np.uint8).tobytes()
        if p not in seen:
            seen.add(p)
            out.append(p)
    return out


def adversarial_patterns(count: int = 1000, seed: int = SEED_C5) -> List[bytes]:
    """C5: `count` patterns sharing one 24-byte prefix with distinct 8..40-byte tails, plus the C2 set."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prefix = ALNUM[rng.integers(0, ALNUM.size, size=24)].tobytes()
    out, seen = [], set()
    while len(out) < count:
        tail = ALNUM[rng.integers(0, ALNUM.size, size=int(rng.integers(8, 41)))].tobytes()
        p = prefix + tail
        if p not in seen:
            seen.add(p)
            out.append(p)
    for p in random_patterns():
        if p not in seen:
            seen.add(p)
            out.append(p)
    return out


# ------------------------------------------------------------------------------- inputs

def random_bytes(n: int, seed: int = SEED_C2_INPUT) -> np.ndarray:
    """C2 input: n uniform random bytes (splitmix64 stream)."""
    out = np.empty(n, dtype=np.uint8)
    if n:
        _load().wl_fill_random(seed, out.ctypes.data, n)
    return out


def _pool_arrays(records: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    flat = np.frombuffer(b"".join(records), dtype=np.uint8)
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
    return flat, offs


def _fill_from_pool(records: Sequence[bytes], alphabet: np.ndarray, n: int, seed: int) -> np.ndarray:
    flat, offs = _pool_arrays(records)
    alphabet = np.ascontiguousarray(alphabet, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    if n:
        _load().wl_fill_from_pool(seed, flat.ctypes.data, offs.ctypes.data, len(records), alphabet.ctypes.data,
                                  alphabet.size, out.ctypes.data, n)
    return out


_PLACEHOLDER = b"\x00"


def http_message_pool(patterns: Sequence[bytes], pool_size: int = 8192, embed_fraction: float = 0.01,
                      seed: int = SEED_C3_INPUT) -> List[bytes]:
    """Message templates for the C3/C4 stream.  0x00 bytes are placeholders the C generator
    replaces with fresh URL-safe characters, so paths / header values / bodies differ in every
    instance; `embed_fraction` of the templates carry one verbatim pattern (a guaranteed match)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x9E3779B97F4A7C15))
    printable = [p for p in patterns if all(32 <= c < 127 for c in p)] or list(patterns)
    methods = [b"GET ", b"POST ", b"HEAD ", b"PUT "]
    agents = [b"Mozilla/5.0 (X11; Linux x86_64)", b"curl/8.4.0", b"python-requests/2.31", b"Wget/1.21"]
    types = [b"text/html", b"application/json", b"application/x-www-form-urlencoded", b"image/png"]

    def blanks(lo, hi):
        return _PLACEHOLDER * int(rng.integers(lo, hi + 1))

    pool = []
    for _ in range(pool_size):
        request = rng.random() < 0.6
        lines = []
        if request:
            lines.append(methods[int(rng.integers(0, 4))] + b"/" + blanks(4, 40) + b" HTTP/1.1\r\n")
            lines.append(b"Host: " + blanks(4, 16) + b".example.com\r\n")
            lines.append(b"User-Agent: " + agents[int(rng.integers(0, 4))] + b"\r\n")
            lines.append(b"Accept: */*\r\n")
            if rng.random() < 0.5:
                lines.append(b"Cookie: sid=" + blanks(16, 32) + b"\r\n")
            if rng.random() < 0.3:
                lines.append(b"Referer: http://" + blanks(6, 24) + b"/\r\n")
        else:
            lines.append(b"HTTP/1.1 200 OK\r\n")
            lines.append(b"Content-Type: " + types[int(rng.integers(0, 4))] + b"\r\n")
            lines.append(b"Set-Cookie: sid=" + blanks(16, 32) + b"\r\n")
        body = blanks(0, 400)
        if rng.random() < embed_fraction:
            pat = printable[int(rng.integers(0, len(printable)))]
            cut = int(rng.integers(0, len(body) + 1))
            body = body[:cut] + pat + body[cut:]
        lines.append(b"Content-Length: " + str(len(body)).encode() + b"\r\n\r\n")
        pool.append(b"".join(lines) + body)
    return pool


def http_stream(n: int, pool: Sequence[bytes], seed: int = SEED_C3_INPUT) -> np.ndarray:
    """C3/C4 input: n bytes of synthetic HTTP traffic assembled from `pool`."""
    return _fill_from_pool(pool, URL_SAFE, n, seed)


def adversarial_pool(patterns: Sequence[bytes], pool_size: int = 4096, seed: int = SEED_C5) -> List[bytes]:
    """C5 records: a shared-prefix pattern whose last 1..4 bytes are replaced by placeholders
    (near miss: the walk goes almost the full pattern length before it traps)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xC5))
    long_pats = [p for p in patterns if len(p) >= 32]
    pool = []
    for _ in range(pool_size):
        p = long_pats[int(rng.integers(0, len(long_pats)))]
        k = int(rng.integers(1, 5))
        pool.append(p[:-k] + _PLACEHOLDER * k)
    return pool


def adversarial_stream(n: int, pool: Sequence[bytes], seed: int = SEED_C5) -> np.ndarray:
    return _fill_from_pool(pool, ALNUM, n, seed)


# ------------------------------------------------------------------------ named configs

class Config:
    """One BASELINE.json configuration: pattern set, table mode and an input-slice generator."""

    def __init__(self, name, description, patterns, perf_mode, make_slice):
        self.name = name
        self.description = description
        self.patterns = patterns
        self.perf_mode = perf_mode          # 0 = PFAC_TIME_DRIVEN (dense), 1 = PFAC_SPACE_DRIVEN (hashed)
        self._make_slice = make_slice

    def input_slice(self, n: int, slice_index: int = 0) -> np.ndarray:
        """n bytes of slice `slice_index` of the stream (slices are independent seeds, C4)."""
        return self._make_slice(n, slice_index)


def make_config(name: str) -> Config:
    name = name.lower()
    if name == "c2":
        pats = random_patterns()
        return Config("c2", "1000 random patterns (len 4-32) over uniform-random bytes, dense 2-D table",
                      pats, 0, lambda n, i: random_bytes(n, SEED_C2_INPUT + i))
    if name in ("c3", "c4"):
        pats = snort_patterns()
        pool = http_message_pool(pats)
        return Config("c3", "30000 Snort-style patterns over synthetic HTTP text, hashed table",
                      pats, 1, lambda n, i: http_stream(n, pool, SEED_C3_INPUT + i))
    if name == "c5":
        pats = adversarial_patterns()
        pool = adversarial_pool(pats)
        return Config("c5", "adversarial near-miss stream (24-byte shared prefix), dense table",
                      pats, 0, lambda n, i: adversarial_stream(n, pool, SEED_C5 + i))
    if name == "c6":
        c5 = adversarial_patterns()
        pool = adversarial_pool(c5)                        # the C5 stream itself, byte for byte
        prefix_pats = c5[:1000]                            # C5's shared-prefix patterns (its other thousand is the C2 set)
        seen = set(prefix_pats)
        pats = [p for p in snort_patterns() if p not in seen] + prefix_pats
        return Config("c6", "adversarial near-miss stream of C5 over C3's 30000 Snort-style patterns + C5's 1000 shared-prefix patterns, hashed table",
                      pats, 1, lambda n, i: adversarial_stream(n, pool, SEED_C5 + i))
    raise ValueError(f"unknown workload {name!r} (c2, c3, c5, c6)")
