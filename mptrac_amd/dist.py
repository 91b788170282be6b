"""One process per GPU: process-group plumbing and the all-reduce hook.

PyTorch is used only as plumbing (``torch.distributed`` with backend "nccl",
which is RCCL on ROCm, over xGMI).  The time-step loop needs no communication
(particles are sharded by index range with replicated meteo grids,
SURVEY.md 8(e)); the only exchanges are sums of small gridded buffers:
write_grid's sums (src/mptrac.c:13862-13872) and module_mixing's cell sums
(src/mptrac.c:5289-5303).  The C ABI calls back with a raw buffer address and a
count of doubles; the hook wraps the address as a tensor without copying and
all-reduces it in place.
"""
import ctypes
import os


class _DeviceBuffer:
    """Minimal __cuda_array_interface__ carrier for a raw device address."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank, _, world = env_rank_world()
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def init_rccl(sim, dist=None):
    """Gives `sim` (hip.Simulation) an RCCL communicator over all ranks of the process group: rank 0 creates
    the identifier, torch.distributed only carries its 128 bytes to the others; from then on the gridded
    reductions are ncclAllReduce calls of the C library on the simulation's own stream (no Python, no host
    synchronisation in the data path).  Without a process group: a single-rank communicator."""
    from .hip import Simulation
    if dist is None or not dist.is_initialized():
        sim.comm_init(1, 0, Simulation.comm_unique_id())
        return
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = Simulation.comm_unique_id() if rank == 0 else bytes(128)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    sim.comm_init(world, rank, bytes(t.cpu().tolist()))


def make_allreduce_hook(device_kind):
    """Returns fn(ptr, count) summing `count` doubles at address `ptr` over all
    ranks in place.  device_kind "cuda": HIP device memory (RCCL);
    "cpu": host memory (gloo; used by the CPU tests of the N > 1 logic)."""
    import torch
    import torch.distributed as dist

    if device_kind == "cuda":
        def hook(ptr, count):
            t = torch.as_tensor(_DeviceBuffer(ptr, count), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
    elif device_kind == "cuda_staged":
        # device buffer, host collective (gloo): lets several ranks share one
        # GPU in tests, which RCCL does not allow
        import numpy as np
        hiprt = ctypes.CDLL("libamdhip64.so")
        hiprt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

        def hook(ptr, count):
            host = np.empty(count, dtype=np.float64)
            if hiprt.hipMemcpy(host.ctypes.data, ctypes.c_void_p(int(ptr)), count * 8, 2):   # device -> host
                raise RuntimeError("hipMemcpy D2H failed")
            t = torch.from_numpy(host)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if hiprt.hipMemcpy(ctypes.c_void_p(int(ptr)), host.ctypes.data, count * 8, 1):   # host -> device
                raise RuntimeError("hipMemcpy H2D failed")
    else:
        import numpy as np

        def hook(ptr, count):
            buf = (ctypes.c_double * count).from_address(int(ptr))
            t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return hook
