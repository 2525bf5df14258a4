"""ctypes mirror of the PFAC C ABI (``include/PFAC.h`` + ``include/pfac_ext.h``).

Method names, argument meaning and status codes are those of the reference API
(``/root/reference/PFAC/include/PFAC.h:27-215``) so the tests read like the
reference's own example programs (``PFAC/test/simple_example.cpp``):

    h = PFAC.create()
    h.readPatternFromFile(path)
    h.matchFromDevice(d_in_ptr, n, d_out_ptr)

Pointers are plain integers (``tensor.data_ptr()`` / ``ndarray.ctypes.data``);
this layer never touches torch.  Functions return the ``PFAC_status_t`` value;
the ``check=True`` default raises :class:`PFACError` on a non-zero status.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

# enum values, include/PFAC.h
PFAC_PLATFORM_GPU, PFAC_PLATFORM_CPU, PFAC_PLATFORM_CPU_OMP = 0, 1, 2
PFAC_AUTOMATIC, PFAC_TEXTURE_ON, PFAC_TEXTURE_OFF = 0, 1, 2
PFAC_TIME_DRIVEN, PFAC_SPACE_DRIVEN = 0, 1

PFACX_KERNEL_FILTER, PFACX_KERNEL_NAIVE, PFACX_KERNEL_AUTO, PFACX_KERNEL_REFTABLE = 0, 1, 2, 3
PFACX_WALKER_AUTO, PFACX_WALKER_WINDOW, PFACX_WALKER_STAGE, PFACX_WALKER_VETO = 0, 1, 2, 3
PFACX_READ_STRICT, PFACX_READ_STRIP_CR = 1, 2
(PFACX_TABLE_DENSE, PFACX_TABLE_HASH_ROWPTR, PFACX_TABLE_HASH_VALPTR, PFACX_TABLE_INITIAL_ROW,
 PFACX_TABLE_FILTER_GRAM3, PFACX_TABLE_FILTER_SHORT, PFACX_TABLE_FILTER_LADDER, PFACX_TABLE_FILTER_FINAL3,
 PFACX_TABLE_CHAIN) = range(9)
PFACX_TABLE_FILTER_GRAM1, PFACX_TABLE_FILTER_PREFIX4, PFACX_TABLE_FILTER_TAIL, PFACX_TABLE_FILTER_TAIL_GLOBAL, PFACX_TABLE_FILTER_SKIP = 9, 10, 11, 12, 13


class STATUS:
    SUCCESS = 0
    BASE = 10000
    ALLOC_FAILED = 10001
    CUDA_ALLOC_FAILED = 10002
    INVALID_HANDLE = 10003
    INVALID_PARAMETER = 10004
    PATTERNS_NOT_READY = 10005
    FILE_OPEN_ERROR = 10006
    LIB_NOT_EXIST = 10007
    ARCH_MISMATCH = 10008
    MUTEX_ERROR = 10009
    INTERNAL_ERROR = 10010


class PFACError(RuntimeError):
    def __init__(self, status: int, where: str, message: str):
        super().__init__(f"{where}: status {status}: {message}")
        self.status = status


class PFACX_info(C.Structure):
    _fields_ = [
        ("structSize", C.c_size_t),
        ("numOfPatterns", C.c_int), ("numOfStates", C.c_int), ("numOfFinalStates", C.c_int),
        ("initialState", C.c_int), ("maxPatternLen", C.c_int), ("numOfLeaves", C.c_int),
        ("perfMode", C.c_int), ("textureMode", C.c_int), ("platform", C.c_int), ("hasDevice", C.c_int),
        ("numOfTableEntry", C.c_size_t), ("sizeOfTableEntry", C.c_size_t), ("sizeOfTableInBytes", C.c_size_t),
        ("filterLog2Bits", C.c_int), ("filterHasShort", C.c_int), ("filterBitsSet", C.c_size_t),
        ("kernelVariant", C.c_int), ("multiProcessorCount", C.c_int),
        ("filterLog2BitsLadder", C.c_int), ("filterLog2BitsFinal3", C.c_int), ("filterBitsSetLadder", C.c_size_t),
        ("chainJumpLog2", C.c_int), ("chainSlots", C.c_size_t),
        ("ladderStops", C.c_size_t), ("ladderGoOns", C.c_size_t), ("ladderThin", C.c_int), ("ladderExtend", C.c_int),
        ("trailingBytesIgnored", C.c_size_t), ("deviceTableBytes", C.c_size_t), ("deviceScratchBytes", C.c_size_t),
        ("streamNearMisses", C.c_int), ("streamDense", C.c_int), ("filterLadderLast", C.c_int), ("filterTailEntries", C.c_size_t),
        ("filterTailGlobalEntries", C.c_size_t), ("filterLog2TailGlobal", C.c_int), ("filterLadderSalt", C.c_uint), ("filterSkipTags", C.c_int),
    ]


class PFACX_scan_stats(C.Structure):
    _fields_ = [("structSize", C.c_size_t), ("walkerRounds", C.c_ulonglong), ("laneSteps", C.c_ulonglong), ("walksStarted", C.c_ulonglong),
                ("level1Hits", C.c_ulonglong), ("tilesPerChunk", C.c_int), ("walksPerLane", C.c_int),
                ("ladderCandidates", C.c_ulonglong), ("denseChunks", C.c_ulonglong), ("filterKernelMs", C.c_double),
                ("stageModeWaves", C.c_ulonglong), ("walker", C.c_int), ("veto", C.c_int)]


_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_lib: Optional[C.CDLL] = None

EXPORTED_SYMBOLS = (
    # include/PFAC.h
    "PFAC_create", "PFAC_destroy", "PFAC_setPlatform", "PFAC_setTextureMode", "PFAC_setPerfMode",
    "PFAC_getErrorString", "PFAC_dumpTransitionTable", "PFAC_readPatternFromFile",
    "PFAC_matchFromDevice", "PFAC_matchFromHost", "PFAC_matchFromDeviceReduce", "PFAC_matchFromHostReduce",
    # include/pfac_ext.h
    "PFACX_createHostOnly", "PFACX_getInfo", "PFACX_getTable", "PFACX_setKernelVariant",
    "PFACX_readPatternFromMemory", "PFACX_getScanStats", "PFACX_saveCompiled", "PFACX_loadCompiled",
    "PFACX_matchFromHostMultiGPU", "PFACX_matchFromHostReduceMultiGPU", "PFACX_readPatternFromFileEx", "PFACX_readPatternFromMemoryEx", "PFACX_trim",
    "PFACX_setKernelTiming", "PFACX_setWalker", "PFACX_prepare",
)
MODULE_SYMBOLS = (  # include/pfac_module.h, exported by libpfac_gfx950.so
    "PFAC_kernel_timeDriven_warpper", "PFAC_kernel_spaceDriven_warpper",
    "PFAC_reduce_kernel", "PFAC_reduce_inplace_kernel", "PFACX_streamProbe", "PFACX_buildInfo",
)


def library_paths() -> Tuple[str, str]:
    # PFAC_HOST_LIB: another build of the HOST library (the sanitizer builds of `make -C pfac_amd/csrc san`: tests/test_sanitizers.py)
    return os.environ.get("PFAC_HOST_LIB") or os.path.join(_LIB_DIR, "libpfac.so"), os.path.join(_LIB_DIR, "libpfac_gfx950.so")


def load_library() -> C.CDLL:
    """Load libpfac.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    host, module = library_paths()
    for p in (host, module):
        if not os.path.exists(p):
            raise ImportError(
                f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C pfac_amd/csrc`. pfac_amd has no non-HIP fallback.")
    lib = C.CDLL(host)
    H = C.c_void_p
    lib.PFAC_create.argtypes = [C.POINTER(H)]
    lib.PFACX_createHostOnly.argtypes = [C.POINTER(H)]
    lib.PFAC_destroy.argtypes = [H]
    lib.PFAC_setPlatform.argtypes = [H, C.c_int]
    lib.PFAC_setTextureMode.argtypes = [H, C.c_int]
    lib.PFAC_setPerfMode.argtypes = [H, C.c_int]
    lib.PFAC_getErrorString.argtypes = [C.c_int]
    lib.PFAC_getErrorString.restype = C.c_char_p
    lib.PFAC_dumpTransitionTable.argtypes = [H, C.c_void_p]
    lib.PFAC_readPatternFromFile.argtypes = [H, C.c_char_p]
    lib.PFAC_matchFromDevice.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.PFAC_matchFromHost.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.PFAC_matchFromDeviceReduce.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    lib.PFAC_matchFromHostReduce.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    lib.PFACX_getInfo.argtypes = [H, C.POINTER(PFACX_info)]
    lib.PFACX_getTable.argtypes = [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.PFACX_setKernelVariant.argtypes = [H, C.c_int]
    if hasattr(lib, "PFACX_setWalker"):                  # (tools/ab.py also loads the libraries of earlier revisions)
        lib.PFACX_setWalker.argtypes = [H, C.c_int]
    lib.PFACX_readPatternFromMemory.argtypes = [H, C.c_char_p, C.c_size_t]
    lib.PFACX_trim.argtypes = [H]
    if hasattr(lib, "PFACX_prepare"):
        lib.PFACX_prepare.argtypes = [H, C.c_size_t]
    lib.PFACX_setKernelTiming.argtypes = [H, C.c_int]
    lib.PFACX_readPatternFromFileEx.argtypes = [H, C.c_char_p, C.c_uint]
    lib.PFACX_readPatternFromMemoryEx.argtypes = [H, C.c_char_p, C.c_size_t, C.c_uint]
    lib.PFACX_getScanStats.argtypes = [H, C.POINTER(PFACX_scan_stats)]
    lib.PFACX_saveCompiled.argtypes = [H, C.c_char_p]
    lib.PFACX_loadCompiled.argtypes = [H, C.c_char_p]
    lib.PFACX_matchFromHostMultiGPU.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    if hasattr(lib, "PFACX_matchFromHostReduceMultiGPU"):
        lib.PFACX_matchFromHostReduceMultiGPU.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    for name in EXPORTED_SYMBOLS:
        if os.environ.get("PFAC_AB_OLD_LIBS") and not hasattr(lib, name):     # tools/ab.py: the library of an earlier revision
            continue
        fn = getattr(lib, name)
        if name != "PFAC_getErrorString":
            fn.restype = C.c_int
    _lib = lib
    return lib


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


def error_string(status: int) -> str:
    s = load_library().PFAC_getErrorString(int(status))
    return s.decode("latin1") if s else ""


class PFAC:
    """One PFAC handle (``PFAC_handle_t``)."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self._lib = load_library()

    # -- lifecycle -----------------------------------------------------------------
    @classmethod
    def create(cls, check: bool = True) -> "PFAC":
        """``PFAC_create``: binds the current HIP device and loads the gfx950 module."""
        lib = load_library()
        h = C.c_void_p()
        st = lib.PFAC_create(C.byref(h))
        if st != 0:
            if h:
                lib.PFAC_destroy(h)
            if check:
                raise PFACError(st, "PFAC_create", error_string(st))
            obj = cls(C.c_void_p())
            obj.create_status = st
            return obj
        obj = cls(h)
        obj.create_status = 0
        # test harness: PFAC_TEST_WALKER=window|stage runs a whole test session with one walker of the full-result kernel
        forced = {"window": PFACX_WALKER_WINDOW, "stage": PFACX_WALKER_STAGE, "veto": PFACX_WALKER_VETO}.get(os.environ.get("PFAC_TEST_WALKER", "").lower())
        if forced is not None:
            obj.setWalker(forced)
        return obj

    @classmethod
    def createHostOnly(cls) -> "PFAC":
        """``PFACX_createHostOnly``: pattern compiler + CPU platforms, no device."""
        lib = load_library()
        h = C.c_void_p()
        st = lib.PFACX_createHostOnly(C.byref(h))
        if st != 0:
            raise PFACError(st, "PFACX_createHostOnly", error_string(st))
        obj = cls(h)
        obj.create_status = 0
        return obj

    def destroy(self) -> int:
        st = self._lib.PFAC_destroy(self._h)
        self._h = C.c_void_p()
        return st

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._h:
            self.destroy()

    def _ret(self, st: int, where: str, check: bool) -> int:
        if check and st != 0:
            raise PFACError(st, where, error_string(st))
        return st

    # -- configuration -------------------------------------------------------------
    def setPlatform(self, platform: int, check: bool = True) -> int:
        return self._ret(self._lib.PFAC_setPlatform(self._h, platform), "PFAC_setPlatform", check)

    def setTextureMode(self, mode: int, check: bool = True) -> int:
        return self._ret(self._lib.PFAC_setTextureMode(self._h, mode), "PFAC_setTextureMode", check)

    def setPerfMode(self, mode: int, check: bool = True) -> int:
        return self._ret(self._lib.PFAC_setPerfMode(self._h, mode), "PFAC_setPerfMode", check)

    def setKernelVariant(self, variant: int, check: bool = True) -> int:
        return self._ret(self._lib.PFACX_setKernelVariant(self._h, variant), "PFACX_setKernelVariant", check)

    def setWalker(self, walker: int, check: bool = True) -> int:
        """PFACX_WALKER_AUTO / _WINDOW / _STAGE: the walker of the full-result filter kernel (include/pfac_ext.h)"""
        return self._ret(self._lib.PFACX_setWalker(self._h, walker), "PFACX_setWalker", check)

    def readPatternFromFile(self, filename, check: bool = True) -> int:
        name = None if filename is None else os.fsencode(filename)
        return self._ret(self._lib.PFAC_readPatternFromFile(self._h, name), "PFAC_readPatternFromFile", check)

    def readPatternFromMemory(self, data: bytes, check: bool = True) -> int:
        """``PFACX_readPatternFromMemory``: the pattern-file bytes without a file."""
        return self._ret(self._lib.PFACX_readPatternFromMemory(self._h, data, len(data)), "PFACX_readPatternFromMemory", check)

    def prepare(self, max_bytes: int = 0, check: bool = True) -> int:
        """``PFACX_prepare``: staging, scratch and code objects of the host paths ahead of the first call."""
        return self._ret(self._lib.PFACX_prepare(self._h, max_bytes), "PFACX_prepare", check)

    def trim(self, check: bool = True) -> int:
        """``PFACX_trim``: free the handle's grow-only device temporaries."""
        return self._ret(self._lib.PFACX_trim(self._h), "PFACX_trim", check)

    def setKernelTiming(self, on: bool, check: bool = True) -> int:
        """``PFACX_setKernelTiming``: HIP events around the filter kernel's launch; ``scanStats().filterKernelMs``."""
        return self._ret(self._lib.PFACX_setKernelTiming(self._h, 1 if on else 0), "PFACX_setKernelTiming", check)

    def readPatternFromFileEx(self, filename, flags: int, check: bool = True) -> int:
        """``PFACX_readPatternFromFileEx``: flags = PFACX_READ_STRICT | PFACX_READ_STRIP_CR."""
        name = None if filename is None else os.fsencode(filename)
        return self._ret(self._lib.PFACX_readPatternFromFileEx(self._h, name, flags), "PFACX_readPatternFromFileEx", check)

    def readPatternFromMemoryEx(self, data: bytes, flags: int, check: bool = True) -> int:
        return self._ret(self._lib.PFACX_readPatternFromMemoryEx(self._h, data, len(data), flags), "PFACX_readPatternFromMemoryEx", check)

    def saveCompiled(self, filename, check: bool = True) -> int:
        """``PFACX_saveCompiled``: the compiled pattern set (trie, tables, prefilter) to a file."""
        return self._ret(self._lib.PFACX_saveCompiled(self._h, os.fsencode(filename)), "PFACX_saveCompiled", check)

    def loadCompiled(self, filename, check: bool = True) -> int:
        """``PFACX_loadCompiled``: replaces the pattern set with a saved one."""
        return self._ret(self._lib.PFACX_loadCompiled(self._h, os.fsencode(filename)), "PFACX_loadCompiled", check)

    def matchFromHostMultiGPU(self, h_input: int, size: int, h_result: int, devices=None, check: bool = True) -> int:
        """``PFACX_matchFromHostMultiGPU``: `devices` = list of device ordinals (None = every visible device)."""
        if devices is None:
            st = self._lib.PFACX_matchFromHostMultiGPU(self._h, h_input, size, h_result, 0, None)
        else:
            arr = (C.c_int * len(devices))(*devices)
            st = self._lib.PFACX_matchFromHostMultiGPU(self._h, h_input, size, h_result, len(devices), arr)
        return self._ret(st, "PFACX_matchFromHostMultiGPU", check)

    def matchFromHostReduceMultiGPU(self, h_input: int, size: int, h_result: int, h_pos: int, devices=None, check: bool = True):
        """``PFACX_matchFromHostReduceMultiGPU`` -> (status, number of pairs)."""
        n = C.c_int(0)
        if devices is None:
            st = self._lib.PFACX_matchFromHostReduceMultiGPU(self._h, h_input, size, h_result, h_pos, C.byref(n), 0, None)
        else:
            arr = (C.c_int * len(devices))(*devices)
            st = self._lib.PFACX_matchFromHostReduceMultiGPU(self._h, h_input, size, h_result, h_pos, C.byref(n), len(devices), arr)
        return self._ret(st, "PFACX_matchFromHostReduceMultiGPU", check), n.value

    def dumpTransitionTable(self, path: str, check: bool = True) -> int:
        fp = _libc.fopen(os.fsencode(path), b"w")
        if not fp:
            raise OSError(f"cannot open {path}")
        try:
            st = self._lib.PFAC_dumpTransitionTable(self._h, fp)
        finally:
            _libc.fclose(fp)
        return self._ret(st, "PFAC_dumpTransitionTable", check)

    # -- matching ------------------------------------------------------------------
    def matchFromDevice(self, d_input: int, size: int, d_result: int, check: bool = True) -> int:
        return self._ret(self._lib.PFAC_matchFromDevice(self._h, d_input, size, d_result),
                         "PFAC_matchFromDevice", check)

    def matchFromHost(self, h_input: int, size: int, h_result: int, check: bool = True) -> int:
        return self._ret(self._lib.PFAC_matchFromHost(self._h, h_input, size, h_result),
                         "PFAC_matchFromHost", check)

    def matchFromDeviceReduce(self, d_input: int, size: int, d_result: int, d_pos: int, check: bool = True):
        n = C.c_int(0)
        st = self._lib.PFAC_matchFromDeviceReduce(self._h, d_input, size, d_result, d_pos, C.byref(n))
        return self._ret(st, "PFAC_matchFromDeviceReduce", check), n.value

    def matchFromHostReduce(self, h_input: int, size: int, h_result: int, h_pos: int, check: bool = True):
        n = C.c_int(0)
        st = self._lib.PFAC_matchFromHostReduce(self._h, h_input, size, h_result, h_pos, C.byref(n))
        return self._ret(st, "PFAC_matchFromHostReduce", check), n.value

    # -- numpy conveniences over matchFromHost (still the C ABI underneath) ----------
    def match_host_array(self, data):
        import numpy as np
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.full(data.size, -7, dtype=np.int32)   # poison: every element must be written
        if data.size:
            self.matchFromHost(data.ctypes.data, data.size, out.ctypes.data)
        return out

    # -- extensions ----------------------------------------------------------------
    def info(self) -> PFACX_info:
        info = PFACX_info()
        info.structSize = C.sizeof(PFACX_info)
        self._ret(self._lib.PFACX_getInfo(self._h, C.byref(info)), "PFACX_getInfo", True)
        return info

    def scanStats(self, positions: int = 0):
        """``PFACX_getScanStats`` of the last filter-kernel launch, plus the derived SURVEY 8(d) C5 figures
        (`positions` = input bytes of that launch)."""
        st = PFACX_scan_stats()
        st.structSize = C.sizeof(PFACX_scan_stats)
        self._ret(self._lib.PFACX_getScanStats(self._h, C.byref(st)), "PFACX_getScanStats", True)
        d = {name: int(getattr(st, name)) for name, _ in PFACX_scan_stats._fields_ if name != "filterKernelMs"}
        if st.filterKernelMs >= 0:
            d["filterKernelMs"] = float(st.filterKernelMs)
        if d["walkerRounds"]:
            d["avg_table_steps_per_walk"] = round(d["laneSteps"] / max(1, d["walksStarted"]), 3)
            d["lane_utilisation"] = round(d["laneSteps"] / (d["walkerRounds"] * 64.0 * d["walksPerLane"]), 4)
            if positions:
                d["early_out_rate"] = round(1.0 - d["walksStarted"] / positions, 6)       # positions that never left LDS
                d["level1_hit_rate"] = round(d["level1Hits"] / positions, 6)
                d["walker_rounds_per_KiB"] = round(d["walkerRounds"] * 64.0 / (positions / 1024.0) / 64.0, 4)
        return d

    def table(self, which: int):
        """Host copy of a compiled table as a numpy array (copy)."""
        import numpy as np
        ptr = C.c_void_p()
        nbytes = C.c_size_t()
        self._ret(self._lib.PFACX_getTable(self._h, which, C.byref(ptr), C.byref(nbytes)), "PFACX_getTable", True)
        dtype = np.uint32 if which >= PFACX_TABLE_FILTER_GRAM3 else np.int32
        if nbytes.value == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * nbytes.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).copy()
