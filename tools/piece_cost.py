#!/usr/bin/env python3
"""Dynamic instruction counts of the step kernel's building blocks (GPU box).

  tools/piece_cost.py run            launch every piece_kernel<K> on the C3 workload (cell-sorted particles)
  tools/piece_cost.sh <tag>          the same under rocprofv3 --pmc, table per piece (minus the empty piece 0)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

PIECES = {0: "empty (load state, store)", 1: "stencil_3d", 2: "stencil_2d", 3: "RK stage: stencil_3d + 12 loads + u,v,w",
          4: "normal_triple", 5: "module_position", 6: "stencil_2d + {ps,pbl} interpolation", 7: "temperature_at",
          8: "dx2coord + dy2coord", 9: "tropo_weight", 10: "sedi()", 11: "uniform01", 12: "module_diff_turb",
          13: "module_convection + module_sedi", 14: "module_diff_meso", 15: "module_advect RK4 (no corner cache)",
          16: "lean: stencil_3d", 17: "lean: horizontal stencil", 18: "lean: RK stage (stencil + loads + u,v,w)",
          19: "lean: module_position", 20: "lean: module_diff_turb", 21: "lean: convection + sedi",
          22: "lean: module_diff_meso", 23: "lean: module_advect RK4 (corner cache)", 24: "normal_pair (uniforms included)",
          25: "log_tab(squares)", 26: "sincosf(2 pi uniform01)"}


def main():
    ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 8)
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(3):          # establishes the locality order and generic positions
        s.run_timestep(k * dt)
    s.synchronize()
    for piece in PIECES:
        for _ in range(2):
            s.test_piece(piece, 1)
    s.close()


if __name__ == "__main__":
    main()
