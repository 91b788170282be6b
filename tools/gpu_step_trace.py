#!/usr/bin/env python3
"""Step-kernel time of every single time step of workload C3 (HIP events of the library around each launch):
shows warm-up effects -- clocks, the age of the internal locality order -- that an average hides.
  python tools/gpu_step_trace.py [steps] [spin=MS] [NAME=VALUE options ...]
spin=MS: keep the GPU busy for about MS milliseconds first with a kernel that does not touch the particles
(mphip_test_piece) -- separates clock ramp-up from effects of the particle distribution."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 70
opts = [a for a in sys.argv[1:] if "=" in a and not a.startswith(("spin=", "pause_at="))]
spin = [float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("spin=")]
ctl, clim, met0, met1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, steps + 2)
sim = hip.Simulation(ctl, clim, met0, met1, atm)
for kv in opts:
    name, value = kv.split("=")
    sim.set_option(name, float(value))
sim.timesteps_init(0.0, 0.0)
dt = sim.ctl.dt_mod
sim.run_timestep(0.0)
sim.synchronize()
if spin:
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < spin[0]:
        sim.test_piece(2, 4)
        n += 1
    print(f"spun {n} launches, {(time.perf_counter() - t0) * 1e3:.1f} ms")
replay = "replay" in sys.argv[1:]
pause = [a for a in sys.argv[1:] if a.startswith("pause_at=")]      # pause_at=K: idle for half a second before step K
pause_at = int(pause[0].split("=")[1]) if pause else -1
out = []
for k in list(range(1, steps + 1)) * (2 if replay else 1):
    if replay and k == 1 and out:        # the same particles and times once more, on a warm device
        sim.replace_particles(atm)
        sim.run_timestep(0.0)
        sim.synchronize()
    if k == pause_at:
        time.sleep(0.5)
    sim.profile_begin()
    t0 = time.perf_counter()
    sim.run_timestep(k * dt)
    sim.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    n, ms = sim.profile_end()
    out.append((k, ms, wall))   # all launches of the step (one; what tools measured as two was the removed option split_step)
for k, ms, wall in out:
    print(f"step {k:3d}  kernel {ms:7.4f} ms   wall {wall:7.3f} ms")
tail = sorted(ms for _, ms, _ in out[len(out) // 2:])
print(f"median of the second half: {tail[len(tail) // 2]:.4f} ms")
sim.close()
