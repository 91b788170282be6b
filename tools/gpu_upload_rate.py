#!/usr/bin/env python3
"""Host -> device rate of one C3 meteo snapshot through mphip_update_met (GPU box): the PCIe-inclusive side
of DESIGN.md section 8."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 10)
s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
nbytes = sum(a.nbytes for a in m1.f3.values()) + sum(a.nbytes for a in m1.f2.values())
for rep in range(3):
    t0 = time.perf_counter()
    s.set_met(1, m1)
    s.synchronize()
    dt = time.perf_counter() - t0
    print(f"update_met: {nbytes / 1e6:.0f} MB in {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s", flush=True)
t0 = time.perf_counter()
s.update_atm(atm)
s.synchronize()
dt = time.perf_counter() - t0
print(f"update_atm: {n_local * 8 * (4 + len(atm['q'])) / 1e6:.0f} MB in {dt * 1e3:.1f} ms")
t0 = time.perf_counter()
g = s.get_atm()
dt = time.perf_counter() - t0
print(f"get_atm:    {dt * 1e3:.1f} ms (pageable host arrays)")
# a C caller's atm_t is persistent: its arrays are page-locked on first use (option pin_host_atm)
s.set_option("pin_host_atm", 1)
for rep in range(3):
    t0 = time.perf_counter()
    s.get_atm(out=g)
    dt = time.perf_counter() - t0
    print(f"get_atm:    {dt * 1e3:.1f} ms (page-locked{', first call registers' if rep == 0 else ''})")
t0 = time.perf_counter()
s.update_atm(g)
s.synchronize()
dt = time.perf_counter() - t0
print(f"update_atm: {dt * 1e3:.1f} ms (page-locked)")
s.close()
