"""Minimal ctypes view of the HIP runtime for the bench harness: events on the NULL stream
(the stream the PFAC library launches on, like the reference's default-stream launches,
PFAC_kernel.cu:188-220) and a device synchronize.  Plumbing only."""

from __future__ import annotations

import ctypes as C

_hip = None


def _rt():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise ImportError("HIP runtime (libamdhip64.so) not found")
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
    return _hip


def _ok(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed with hipError {err}")


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        _ok(_rt().hipEventCreate(C.byref(self.h)), "hipEventCreate")

    def record(self, stream: int = 0):
        _ok(_rt().hipEventRecord(self.h, C.c_void_p(stream)), "hipEventRecord")

    def synchronize(self):
        _ok(_rt().hipEventSynchronize(self.h), "hipEventSynchronize")

    def elapsed_ms(self, later: "Event") -> float:
        ms = C.c_float()
        _ok(_rt().hipEventElapsedTime(C.byref(ms), self.h, later.h), "hipEventElapsedTime")
        return float(ms.value)

    def __del__(self):
        try:
            if self.h:
                _rt().hipEventDestroy(self.h)
        except Exception:
            pass


def pci_bus_id(device: int) -> str:
    """"0000:05:00.0" of a HIP device (hipDeviceGetPCIBusId), or "" if the runtime does not say."""
    rt = _rt()
    buf = C.create_string_buffer(64)
    try:
        rt.hipDeviceGetPCIBusId.argtypes = [C.c_char_p, C.c_int, C.c_int]
        if rt.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return ""
    except AttributeError:
        return ""
    return buf.value.decode(errors="replace").lower()


def pci_code(bus_id: str) -> int:
    """domain:bus:device.function as one integer (domain << 24 | bus << 16 | device << 8 | function): what a rank puts among its gathered int64 facts; -1 = unknown"""
    try:
        dom, bus, rest = bus_id.split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 24) | (int(bus, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16)
    except ValueError:
        return -1


def pci_string(code: int) -> str:
    return "" if code < 0 else "%04x:%02x:%02x.%x" % (code >> 24, (code >> 16) & 0xFF, (code >> 8) & 0xFF, code & 0xFF)


def numa_node_of_pci(bus_id: str) -> int:
    """NUMA node of a PCI device from sysfs (-1: unknown, or a single-node machine that says so)"""
    try:
        return int(open(f"/sys/bus/pci/devices/{bus_id}/numa_node").read().strip())
    except (OSError, ValueError):
        return -1


def device_synchronize():
    _ok(_rt().hipDeviceSynchronize(), "hipDeviceSynchronize")
