#!/usr/bin/env python3
"""Module ablation on the C3 grid (GPU box): time of the fused kernel for
growing module sets, cell-sorted particles."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

VARIANTS = {
    "advect": dict(diffusion=0, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1),
    "advect+turb": dict(diffusion=1, turb_mesox=0.0, turb_mesoz=0.0, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1),
    "advect+turb+meso": dict(diffusion=1, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1),
    "advect+turb+conv+sedi": dict(diffusion=1, turb_mesox=0.0, turb_mesoz=0.0),
    "full C3": dict(),
}


def run(name, over, n=None, steps=10, interval=10, workload="C3", tile=None, opts=None):
    ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs(workload, 0, 1, steps + 4)
    ctl.update(over)
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.set_option("locality_sort_interval", interval)
    if tile:
        s.set_option("locality_tile", tile)
    for k, v in (opts or {}).items():
        s.set_option(k, v)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    k = 0
    for _ in range(3):
        s.run_timestep(k * dt)
        k += 1
    s.synchronize()
    s.profile_begin()
    for _ in range(steps):
        s.run_timestep(k * dt)
        k += 1
    nl, ms = s.profile_end()
    print(f"{name:24s}: step_kernel {ms / nl:.3f} ms  -> {n_local / (ms / nl * 1e-3):.3e} p-steps/s (kernel only)", flush=True)
    s.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grid":
        for blocks in (1024, 2048, 4096, 8192, 16384, 40000):
            for xm in (1, 0):
                run(f"blocks={blocks} xcd_map={xm} full C3", VARIANTS["full C3"], opts={"step_blocks": blocks, "xcd_map": xm})
    elif len(sys.argv) > 1 and sys.argv[1] == "tiles":
        for tile in (1, 2, 4, 8, 16, 32):
            for name in ("advect", "full C3"):
                run(f"tile={tile} {name}", VARIANTS[name], tile=tile)
    else:
        for name, over in VARIANTS.items():
            run(name, over)
