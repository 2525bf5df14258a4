"""tools/clock_watch.py [workloads] [seconds per workload]   (GPU box only)
Kernel time of back-to-back PFAC_matchFromDevice calls on the 1 GiB stream next to what the driver says about the GPU while they run: shader
clock (sysfs freq1_input / pp_dpm_sclk of the device's PCI function), socket power and its cap (hwmon), sampled every 50 ms by a thread.
For a launch that is bound by vector-instruction issue (C6) the kernel time is the instruction count over the clock: this shows whether the
two classes of that kernel's time (1.07 / 1.16 ms: profiles/r06_experiments.md section 5a) are two clocks."""
import glob, os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfac_amd import api, hiprt, workloads as wl


def sysfs_paths():
    bdf = hiprt.pci_bus_id(0)
    base = f"/sys/bus/pci/devices/{bdf}"
    hw = sorted(glob.glob(base + "/hwmon/hwmon*"))
    return bdf, base, (hw[0] if hw else None)


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sample(base, hw):
    d = {}
    if hw:
        for name in ("freq1_input", "power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input"):
            v = read(os.path.join(hw, name))
            if v is not None and v.lstrip("-").isdigit():
                d[name] = int(v)
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        s = read(base + "/" + name)
        if s:
            cur = [l for l in s.splitlines() if l.rstrip().endswith("*")]
            if cur:
                d[name[7:]] = cur[0].split(":")[1].replace("*", "").strip()
    return d


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "c6,c3").split(",")
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    bdf, base, hw = sysfs_paths()
    print("device", bdf, "hwmon", hw, "idle sample", sample(base, hw))
    n = 1 << 30
    for name in names:
        cfg = wl.make_config(name)
        pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
        h = api.PFAC.create()
        h.setPerfMode(cfg.perf_mode)
        h.readPatternFromFile(pf)
        h.setKernelTiming(True)
        d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
        d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
        for _ in range(8):
            h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
        torch.cuda.synchronize()
        samples, stop = [], threading.Event()

        def watch():
            while not stop.is_set():
                samples.append((time.perf_counter(), sample(base, hw)))
                time.sleep(0.05)
        th = threading.Thread(target=watch)
        th.start()
        rows = []
        t_end = time.perf_counter() + secs
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            for _ in range(40):                # back to back, as bench.py times them
                h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rows.append((t1, (t1 - t0) / 40 * 1e3, h.scanStats(n).get("filterKernelMs")))
        stop.set()
        th.join()
        print(f"== {name}: {len(rows)} groups of 40 calls, {len(samples)} sysfs samples")
        for t, call_ms, last_kernel in rows:
            near = min(samples, key=lambda s: abs(s[0] - t))[1] if samples else {}
            print(f"  ms per call {call_ms:.4f} (last launch's filter kernel {last_kernel:.4f}) | " + " ".join(f"{k}={v}" for k, v in near.items()))
        h.destroy()
        del d_in, d_out
        time.sleep(2.0)                        # the next workload starts from an idle GPU


if __name__ == "__main__":
    main()
