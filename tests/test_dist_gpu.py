"""Two ranks with index-range shards on the GPU box (both on device 0, host
collective): the in-step exchange of module_mixing's cell sums and the
gridded-output reduction reproduce the single-context / oracle result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r"""
import os, sys
sys.path[:0] = [%(root)r, %(here)r]
import numpy as np
import cases
from mptrac_amd import dist as mdist, hip
from oracle import binding as B

d = mdist.init_process_group("gloo")
rank, _, world = mdist.env_rank_world()
n = 8001
ctl, clim, m0, m1, atm = cases.make_case("full", n=n)
ctl["sort_dt"] = -999.0            # module_sort under sharding orders each shard (documented deviation)
lo, hi = hip.shard_range(n, rank, world)
s = hip.Simulation(ctl, clim, m0, m1, atm, device=0, shard=(lo, hi))
s.set_allreduce(mdist.make_allreduce_hook("cuda_staged"))
s.timesteps_init(0.0, 0.0)
o = B.Oracle(ctl, clim, m0, m1, atm)
o.timesteps_init()
ts = cases.step_times(o.ctl)[:7]
for t in ts:
    s.run_timestep(t)
    o.run_timestep(t)
g, r = s.state(), o.state()
for k in ("lon", "lat", "p"):
    assert cases.rel_err(g[k], r[k][lo:hi]) <= 1e-10, k
assert cases.q_rows_err(o.ctl, g["q"], r["q"][:, lo:hi])[0] <= 1e-10, cases.q_rows_err(o.ctl, g["q"], r["q"][:, lo:hi])
assert np.abs(g["q"][0] - atm["q"][0][lo:hi]).max() > 1e-6          # mixing + decay did something
cnt, mean, sig = s.grid_sums(ts[-1])
co, mo, so = o.grid_sums(ts[-1])
assert np.array_equal(cnt, co) and cases.rel_err(mean, mo) <= 1e-12
d.barrier()
print("rank", rank, "ok", int(cnt.sum()))
"""


def test_two_ranks_mixing_exchange_and_grid_reduction():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "here": HERE}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert f"rank {rank} ok" in out, out[-3000:]
