"""pfac_amd -- MI355X-native PFAC (Parallel Failureless Aho-Corasick) match path.

The product is two shared libraries built in-tree from ``pfac_amd/csrc``:

* ``lib/libpfac.so``         the reference-compatible C ABI (``include/PFAC.h``)
* ``lib/libpfac_gfx950.so``  the hand-written HIP kernel module for CDNA4

This Python package is only the host-side convenience layer used by the tests
and the bench harness: a ctypes mirror of the C API (:mod:`pfac_amd.api`),
deterministic workload generators (:mod:`pfac_amd.workloads`) and the slice
planner for multi-GPU runs (:mod:`pfac_amd.sharding`).  Nothing in here
computes matches: every match goes through the C ABI into the HIP kernels, and
importing :mod:`pfac_amd.api` fails loudly if the libraries are missing.
"""

from .api import (  # noqa: F401
    PFAC,
    PFACError,
    PFAC_AUTOMATIC,
    PFAC_PLATFORM_CPU,
    PFAC_PLATFORM_CPU_OMP,
    PFAC_PLATFORM_GPU,
    PFAC_SPACE_DRIVEN,
    PFAC_TEXTURE_OFF,
    PFAC_TEXTURE_ON,
    PFAC_TIME_DRIVEN,
    STATUS,
    library_paths,
    load_library,
)
