"""N > 1 logic on CPU: two gloo ranks, index-range shards, all-reduce hook.

The GPU path cannot run here, so each rank computes its shard's gridded sums
with the oracle into a host buffer and hands the buffer's address to the same
hook the C ABI calls on the GPU box (there with a device address and RCCL).
"""
import os
import socket
import subprocess
import sys

import numpy as np

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
hook = mdist.make_allreduce_hook("cpu")
n = 6001
ctl, clim, m0, m1, atm = cases.make_case("full", n=n, grid="tiny")
lo, hi = hip.shard_range(n, rank, world)
sub = {k: (v[lo:hi] if k != "q" else v[:, lo:hi]) for k, v in atm.items()}
cnt, mean, sig = B.Oracle(ctl, clim, m0, m1, sub).grid_sums(0.0)
buf = np.concatenate([cnt.astype(np.float64), mean.ravel(), sig.ravel()])
hook(buf.ctypes.data, buf.size)
cg, mg, sg = B.Oracle(ctl, clim, m0, m1, atm).grid_sums(0.0)
ref = np.concatenate([cg.astype(np.float64), mg.ravel(), sg.ravel()])
assert np.array_equal(buf[:cg.size], ref[:cg.size]), "counts"
assert np.max(np.abs(buf - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-13
# the timing reduction bench.py uses: MAX over ranks
import torch
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
d.all_reduce(t, op=d.ReduceOp.MAX)
assert t.item() == world
d.barrier()
print("rank", rank, "ok", int(buf[:cg.size].sum()))
"""


def test_two_rank_gloo_shards_and_allreduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "here": HERE}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok 6001" in out, out
