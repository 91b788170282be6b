#!/usr/bin/env python3
"""Benchmark of the MPTRAC per-particle time-step loop on MI355X.

Metric (BASELINE.json): particle-steps/s.  A "step" is one
mptrac_run_timestep over this rank's particles.  Default workload "C3" is
BASELINE configs[2]: 10^7 particles per GPU, RK4 advection + turbulent and
mesoscale diffusion + convection + sedimentation on the synthetic
0.5 deg x 0.5 deg x 137-level ERA5-shaped grid (721 x 361 x 137 incl. the
periodic column), fp64, inputs resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, particles sharded by global index range (weak
scaling: 10^7 per GPU), meteo grids replicated, no communication inside the
step; one gridded-output reduction (RCCL all-reduce issued by the C library on
the simulation's stream, include/mptrac_hip.h: mphip_comm_init) closes the timed
region, as BASELINE configs[3] prescribes.

The JSON line also carries the HBM roofline of the fused step kernel
(algorithmic bytes / measured kernel time, HIP events on the launch stream)
and a CPU baseline (this repo's OpenMP oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, MI355X_MICROARCH.md

WORKLOADS = {
    # name: (grid, particles per GPU, control, quantities, meteo fields)
    "C3": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1),
           ("m", "rp", "rhop"), ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel")),
    # C3 with the reference's DEFAULT integrator (ADVECT 2, the midpoint scheme) instead of the baseline's RK4
    "C3d": ("C3", 10 ** 7, dict(advect=2, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1),
            ("m", "rp", "rhop"), ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel")),
    # C3 with module_meteo every step (the reference's default MET_DT_OUT 0.1) filling the quantity set of the
    # reference's tests/trac_test (t, u, v, w, zg, pv, ps, pt); not the headline configuration (SURVEY row 22:
    # benchmarks of the reference set MET_DT_OUT 0)
    "C3m": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1),
            ("m", "rp", "rhop", "t", "u", "v", "w", "zg", "pv", "ps", "pt"),
            ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel", "z", "pv", "pt")),
    # C3 with the closure inside the boundary layer (TURB_PBL_SCHEME 1: module_diff_pbl) on top; the
    # mesoscale part stays horizontal (the closure keeps uvwp[2] in m/s, module_diff_meso in hPa/s: with both vertical
    # parts the reference itself drives pressures negative, tests/cases.py)
    "C3p": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1, turb_pbl_scheme=1, turb_mesoz=0.0),
            ("m", "rp", "rhop"), ("u", "v", "w", "t", "h2o", "ps", "pbl", "cape", "cin", "pel", "ess", "nss", "shf")),
    # BASELINE configs[4] / SURVEY C5: C3 + module_sort and inter-parcel mixing every step, decay, wet and dry
    # deposition (per-GPU part; the survey's control line)
    "C5": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1, sort_dt=180.0,
                               mixing_trop=1e-3, mixing_strat=1e-6, mixing_dt=180.0, tdec_trop=259200.0,
                               tdec_strat=259200.0, dry_depo_vdep=0.15, wet_depo_ic_a=1e-4, wet_depo_ic_b=0.8,
                               wet_depo_bc_a=5e-5, wet_depo_bc_b=0.6),
           ("m", "rp", "rhop"), ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel", "pct", "pcb", "cl", "lwc",
                                 "rwc", "iwc", "swc")),
    # C5 without module_mixing: the deposition modules run inside the launch that moves the particles
    # (not a baseline configuration; measures what the split around module_mixing costs)
    "C5n": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1, sort_dt=180.0,
                                tdec_trop=259200.0, tdec_strat=259200.0, dry_depo_vdep=0.15, wet_depo_ic_a=1e-4,
                                wet_depo_ic_b=0.8, wet_depo_bc_a=5e-5, wet_depo_bc_b=0.6),
            ("m", "rp", "rhop"), ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel", "pct", "pcb", "cl", "lwc",
                                  "rwc", "iwc", "swc")),
    # C3 + inter-parcel mixing every step WITHOUT module_sort: the ordered cell sums on the internal locality
    # order (not a baseline configuration; measures that path)
    "C3x": ("C3", 10 ** 7, dict(advect=4, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1,
                                mixing_trop=1e-3, mixing_strat=1e-6, mixing_dt=180.0),
            ("m", "rp", "rhop"), ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel")),
    # C3 with model-level (zeta) advection (ADVECT_VERT_COORD 1), otherwise as C3
    "C3z": ("C3", 10 ** 7, dict(advect=4, advect_vert_coord=1, dt_mod=180.0, diffusion=1, conv_cape=0.0, rng_type=1),
            ("m", "rp", "rhop", "zeta"),
            ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel", "pl", "ul", "vl", "zetal", "zeta_dotl")),
    "C2": ("C2", 10 ** 6, dict(advect=4, dt_mod=180.0, diffusion=1, turb_mesox=0.0, turb_mesoz=0.0, rng_type=1),
           ("m",), ("u", "v", "w", "ps", "pbl")),
    "C1": ("C1", 10 ** 4, dict(advect=4, dt_mod=180.0, rng_type=1), ("m",), ("u", "v", "w", "ps")),
}


def algorithmic_bytes_per_pstep(workload, met, np_local):
    """A = A_state + A_met / np.  A_state = particle state a
    fused step must read and write once; A_met = every packed grid byte the
    step can touch, once per launch."""
    c3 = 64 + 24 + 16              # time,lon,lat,p R+W; uvwp R+W; rp,rhop R
    nq = len(WORKLOADS[workload][3])
    mixing = 24                    # per mixed quantity (here: m): the quantity R+W and the cell mean
    sort = 4 * 16 + 16 * (4 + nq)  # module_sort on a sort step: radix passes over (key, index) + the permutation of the arrays
    state = {"C3": c3,
             "C3m": c3,            # (the step kernel's bytes; module_meteo is a separate kernel)
             "C3d": c3, "C3p": c3,
             # C5 = C3 + 16 (q[m] R+W: decay, deposition) + mixing + sort, every step; the cloud-water
             # grids of module_wet_depo are NOT counted (read below cloud tops only): the conservative figure
             "C5": c3 + 16 + mixing + sort, "C3x": c3 + 16 + mixing, "C5n": c3 + 16 + sort,
             "C3z": c3 + 16,
             "C2": 64, "C1": 64}[workload]
    wind = met.nx * met.ny * met.np * 32            # {u,v,w,t} x 2 snapshots, float
    sfc = met.nx * met.ny * 64                      # 8 surface fields x 2 snapshots
    # few particles on a large grid cannot touch every grid byte: the grid term is capped by the bytes of
    # the distinct records one particle's stencils cover (8 corners x 32 B {u,v,w,t} x 2 snapshots, 4 columns
    # x 64 B surface fields; the Runge-Kutta stages and the other modules re-use them)
    gathered = {"C1": 256, "C2": 512}.get(workload, 512) * float(np_local)
    met_bytes = min(float(wind + sfc), gathered)
    return state + met_bytes / float(np_local), state, met_bytes


def build_inputs(workload, rank, world, steps_total, particles=None):
    from mptrac_amd.clim import load_clim_tropo
    from mptrac_amd.ctl import ctl_from_quantities
    from mptrac_amd.synth import synthetic_met, synthetic_particles
    grid, n_per_gpu, ctl, quantities, fields = WORKLOADS[workload]
    if particles:
        n_per_gpu = int(particles)
    ctl = dict(ctl)
    ctl.update(ctl_from_quantities(quantities))
    # one meteo interval long enough for all steps (synthetic: no file boundary)
    dt_met = 3600.0 * max(1, int(np.ceil((steps_total + 1) * ctl["dt_mod"] / 3600.0)))
    ctl.update(dt_met=dt_met, t_stop=dt_met)
    met0 = synthetic_met(grid, 0.0, 1.0, fields=fields)
    met1 = synthetic_met(grid, dt_met, 1.25, fields=fields)
    n_total = n_per_gpu * world
    # every rank generates only its own index range of the global seeded set
    atm = synthetic_particles(n_per_gpu, seed=12345, quantities=quantities, first=rank * n_per_gpu)
    if "zeta" in quantities:      # a vertical coordinate inside the range of the synthetic zetal field
        atm["q"][list(quantities).index("zeta")] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
    return ctl, load_clim_tropo(), met0, met1, atm, n_per_gpu, n_total


# What pins the oracle each module is checked against (tests/test_oracle_pins.py, tests/test_gpu_parity.py).
# "reference": an artefact the reference's own tests hold is reproduced digit for digit through this module;
# "restatement": no such artefact exists here (the reference's meteo files for them are not in the tree and the
# reference cannot be built in this image) -- the HIP path is checked against a line-by-line restatement only.
PARITY_PINS = {
    "pinned_by_reference_goldens": [
        "module_advect ADVECT 2 (dd_test, coord_test)", "module_position", "module_timesteps",
        "module_diff_turb horizontal branch (coord_test)", "module_diff_meso (coord_test)",
        "module_rng Squares + Box-Muller (coord_test, known-answer values)", "module_decay bookkeeping (dd_test)",
        "sedi() (tools_test sedi.tab)", "intpol_met_* 2-D / 3-D / time (coord_test)", "write_grid (atm_test, dt_test, trac_test grids)",
        "module_meteo t, u, v, w (coord_test), humidity macros (met_test)", "clim_tropo, locate_* (known-answer values)"],
    "restatement_only": [
        "module_advect ADVECT 4 (RK4: the headline integrator; old-latitude rule mptrac.c:3672 by review)",
        "module_diff_turb vertical branch", "module_convection", "module_sedi inside a run", "module_mixing",
        "module_wet_depo", "module_dry_depo", "model-level advection (intpol_met_4d_zeta)", "module_diff_pbl",
        "module_isosurf", "module_bound_cond", "module_sort tie order (stable by index; GSL's is unspecified)"],
    # (no reference-held golden can be generated here; every entry of restatement_only but the sort's tie order is stated
    # a second time in numpy from the reference's text and agrees with the oracle to 1e-13)
    "second_opinion": "tests/refmodules.py + tests/test_oracle_second_opinion.py",
}


def alu_roof(workload, n_local, kernel_ms):
    """"state both bounds honestly".  The fused step is bound by VALU issue, not by HBM: per-instruction
    issue costs of gfx950 measured with the SIMDs full (tools/micro/valu_issue.hip -> profiles/r04_valu_issue.txt)
    x the dynamic instruction mix of the kernel (rocprofv3 SQ_INSTS_VALU_* -> profiles/*_instruction_mix.txt) give
    the cycles one wave needs per particle-step; every SIMD of the 256 CUs works through n / 64 / 1024 waves."""
    f = os.path.join(ROOT, "profiles", "alu_model.json")
    if not os.path.exists(f):
        return None
    m = json.load(open(f)).get(workload)
    if not m:
        return None
    wave_steps_per_simd = n_local / 64.0 / 1024.0
    model_ms = m["cycles_per_wave_step"] * wave_steps_per_simd / (m["sustained_clock_ghz"] * 1e9) * 1e3
    return {"bound": "valu_issue", "model_ms_per_step": model_ms, "frac_alu": model_ms / kernel_ms if kernel_ms else None,
            "cycles_per_wave_step": m["cycles_per_wave_step"], "valu_insts_per_64_particle_steps": m["valu_insts_per_64_particle_steps"],
            "sustained_clock_ghz": m["sustained_clock_ghz"], "source": m["source"]}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, ctl, clim, met0, met1, atm, n_sample, n_steps):
    """The OpenMP oracle (oracle/: this repository's C restatement of the reference's CPU path, "port" -- the
    reference itself cannot be built in this image) on the first n_sample particles of the same
    workload: all usable host cores, and one thread on a tenth of the sample."""
    from oracle import binding as B

    def timed(n, steps, threads):
        used = B.lib().orc_set_num_threads(threads)
        sub = {k: (v[:n].copy() if k != "q" else v[:, :n].copy()) for k, v in atm.items()}
        o = B.Oracle(ctl, clim, met0, met1, sub)
        o.timesteps_init()
        dt = o.ctl.dt_mod
        o.run_timestep(0.0)            # the dt = 0 first call (moves nothing)
        o.run_timestep(dt)             # warm-up step
        t0 = time.time()
        for k in range(2, 2 + steps):
            o.run_timestep(k * dt)
        return n * steps / (time.time() - t0), used
    rate, cores = timed(n_sample, n_steps, B.usable_cores())
    n1 = max(1000, n_sample // 10)
    rate1, _ = timed(n1, max(2, n_steps // 2), 1)
    return {"value": rate, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "nproc": os.cpu_count(), "value_1_thread": rate1,
            "sample": f"first {n_sample} particles of workload {workload}, {n_steps} steps, OpenMP oracle "
                      f"({cores} threads); 1 thread: first {n1} particles, {max(2, n_steps // 2)} steps"}


def visible_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        return n.value if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def multi_gpu_probe(args):
    """A one-GPU invocation on a box with several GPUs: the same workload once more over N = min(8, visible
    GPUs) ranks, launched as the driver launches N > 1 (torch.distributed.run, one process per GPU, the
    library's RCCL communicator), so that any multi-GPU box produces an N > 1 line without further action.  Not
    part of `value`; None on a one-GPU box."""
    import subprocess
    ndev = visible_gpus()
    forced = os.environ.get("MPTRAC_PROBE_RANKS")      # (dry run of this path on a one-GPU box: MPTRAC_PROBE_RANKS=1)
    if ndev < 2 and not forced:
        return None
    n = int(forced) if forced else min(8, ndev)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__), "--gpus", str(n),
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload, "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    except subprocess.TimeoutExpired:
        return {"n_gpus": n, "error": "timeout after 420 s"}
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    if res.returncode != 0 or len(lines) != 1:
        return {"n_gpus": n, "error": res.stderr.decode()[-800:]}
    line = json.loads(lines[0])
    return {"n_gpus": line["n_gpus"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
            "scaling": line["scaling"], "rccl_ranks": line["config"]["rccl_ranks"], "reduction": line["config"]["reduction"],
            "particles_total": line["config"]["particles_total"],
            "kernel_ms_per_rank": line["roofline"]["kernel_ms_per_rank"]}


def build_id():
    """Hash of the kernel sources: ties the committed PMC profile (profiles/pmc_traffic.json) to the build it
    was taken from."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "mptrac_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp", ".h")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 60 steps = three meteo intervals of the survey's control set (DT_MOD 180) and exactly one re-sort of the
    # internal locality order (every 60 steps) inside the timed region, whatever its phase
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--particles", type=float, default=0,
                    help="particles per GPU instead of the workload's own count (density sweeps; the bench line says so)")
    ap.add_argument("--eager-meteo", action="store_true",
                    help="workload C3m: launch module_meteo inside every time step instead of before each output")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2 * 10 ** 6)     # x 20 steps: about 10 s on 16 host cores
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--use-torch", action="store_true", help="go through torch.distributed even at N = 1")
    ap.add_argument("--torch-allreduce", action="store_true",
                    help="N > 1: reduce through the torch.distributed callback instead of the library's RCCL communicator")
    ap.add_argument("--rccl-single", action="store_true",
                    help="N = 1: give the context a one-rank RCCL communicator (exercises the native reduction path)")
    ap.add_argument("--atomic-sums", action="store_true",
                    help="cell sums of module_mixing / the gridded output with floating-point atomics (option "
                         "deterministic_sums 0) instead of the reference's order")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="mphip_set_option(NAME, VALUE) before the run (tuning experiments)")
    ap.add_argument("--device-warmup-ms", type=float, default=150.0,
                    help="keep the GPU under the step kernel's load for this long right before the timed region (on a "
                         "scratch copy of the workload; 0 = off).  An MI355X that comes from idle runs the step kernel "
                         "10-25 %% slower for its first ~30 ms under load (power management: 0.99 -> 1.13 -> 0.88 ms, "
                         "profiles/r03_clock_ramp.txt); a production run is thousands of steps long, so the timed "
                         "steps are taken at the settled clocks")
    ap.add_argument("--multi-step", choices=("on", "off"), default="on",
                    help="on: the K timed steps go to the library as ONE call (mphip_run_timesteps: the reference's time "
                         "loop, trac.c:204-226; steps with nothing scheduled between them -- no module_sort, mixing, "
                         "output -- share a kernel launch in which every particle takes its steps one after the other: "
                         "same bits, the particle state and the meteo lines it uses stay in the caches from step to "
                         "step).  off: K mphip_run_timestep calls, one launch per step.  The line reports the step "
                         "kernel's time under both")
    ap.add_argument("--no-multi-gpu-probe", action="store_true",
                    help="N = 1 on a box with several GPUs: do not append the short run over min(8, visible GPUs) RCCL "
                         "ranks (`multi_gpu_probe` in the line; outside the timed region, a separate launch)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="(diagnostic) do not bracket the step kernel with HIP events; roofline is then not reported")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    use_dist = world > 1 or args.use_torch

    dist = None
    if use_dist:
        import torch   # noqa: F401  (loads the HIP runtime before the C ABI library)
        from mptrac_amd import dist as mdist
        torch.cuda.set_device(local_rank)
        dist = mdist.init_process_group("nccl")

    from mptrac_amd import hip
    if use_dist:
        # should the in-tree library look stale on this box, one rank rebuilds it and the others wait
        if local_rank == 0:
            hip.load()
        dist.barrier()

    steps_total = args.warmup + args.steps + 1 + 10     # (+ the ten single-launch steps behind the timed region)
    ctl, clim, met0, met1, atm, n_local, n_total = build_inputs(args.workload, rank, world, steps_total, particles=args.particles)
    sim = hip.Simulation(ctl, clim, met0, met1, atm, device=local_rank,
                         shard=(rank * n_local, (rank + 1) * n_local), n_total=n_total)
    # a second, throw-away copy of the workload: its time steps bring the device to its settled clocks without
    # touching the particles that are timed (at most 4e7 particles: the effect is one of short launches)
    scratch = None
    if args.device_warmup_ms > 0:
        n_scr = min(n_local, 4 * 10 ** 7)
        scr_atm = {k: (v[:n_scr] if k != "q" else v[:, :n_scr]) for k, v in atm.items()}
        # (its own, long meteo interval -- the same arrays under a later time stamp -- so that it can take as many
        # steps as the warm-up needs)
        scr_span = 3600.0 * 24
        scr_met1 = type(met1)(scr_span, met1.lon, met1.lat, met1.p, met1.f3, met1.f2)
        # (the very same module set: measured with other instantiations of the step kernel as the load -- the
        # general one, the one without convection and sedimentation -- the timed kernel still starts 4-10 % slow:
        # the power management re-settles whenever the load changes character.  A kernel trace of this command
        # therefore averages step_kernel<255u> over the scratch launches too: ~160 of them, of which the ~30 of
        # the ramp move that average by a percent or two)
        scr_ctl = dict(ctl, dt_met=scr_span, t_stop=scr_span)
        scratch = hip.Simulation(scr_ctl, clim, met0, scr_met1, scr_atm, device=local_rank, shard=(0, n_scr),
                                 n_total=n_scr)
        scr_steps = int(scr_span / ctl["dt_mod"]) - 1
    reduction = "none (one rank)"
    if use_dist and args.torch_allreduce:
        from mptrac_amd import dist as mdist
        sim.set_allreduce(mdist.make_allreduce_hook("cuda"))
        reduction = "torch.distributed callback"
    elif use_dist or args.rccl_single:
        # the library's own RCCL communicator: all-reduces on the simulation's stream, no Python in the data path
        from mptrac_amd import dist as mdist
        # No silent fall-back: a run that was asked to reduce through RCCL and cannot, stops (every rank raises:
        # the identifier broadcast has happened or failed on all of them alike).  --torch-allreduce is the
        # explicit way to reduce through torch.distributed instead.
        try:
            mdist.init_rccl(sim, dist)
        except Exception as exc:
            raise SystemExit(f"rank {rank}: native RCCL communicator unavailable ({exc}); "
                             "pass --torch-allreduce to reduce through torch.distributed instead")
        reduction = "rccl (native, on the step stream)"
    rccl_ranks, rccl_rank = sim.comm_query()
    if (use_dist and world > 1 and not args.torch_allreduce) and (rccl_ranks != world or rccl_rank != rank):
        raise SystemExit(f"rank {rank}: the RCCL communicator reports rank {rccl_rank} of {rccl_ranks}, expected {rank} of {world}")
    if args.eager_meteo:
        sim.set_option("lazy_meteo", 0)
    if args.atomic_sums:
        sim.set_option("deterministic_sums", 0)
    for kv in args.option:
        name, value = kv.split("=")
        sim.set_option(name, float(value))
    sim.timesteps_init(0.0, 0.0)
    dt = sim.ctl.dt_mod

    def barrier():
        sim.synchronize()
        if use_dist:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # the reference's first call (t = t_start) has dt = 0 and moves nothing
    sim.run_timestep(0.0)
    k = 1
    # (the W warm-up steps are the first load the device sees: their kernel time is reported as the cold figure)
    if not args.no_kernel_events:
        sim.profile_begin()
    for _ in range(args.warmup):
        sim.run_timestep(k * dt)
        k += 1
    cold_launches, cold_ms = (0, float("nan")) if args.no_kernel_events else sim.profile_end()
    grid_out = sim.grid_sums((k - 1) * dt)     # warm the reduction path (buffers, RCCL communicator) with a real output
    warm_steps, warm_ms = 0, 0.0
    if scratch is not None:
        # Device warm-up, directly in front of the timed region and behind everything that allocates (the first
        # steps of a simulation create its sort buffers: hipMalloc leaves the device idle for milliseconds, and
        # the clocks start over).  The launches of the scratch copy queue up without host synchronisation in
        # between; the timed region starts behind them with no idle gap.
        scratch.timesteps_init(0.0, 0.0)
        scratch.run_timestep(0.0)
        scratch.run_timestep(dt)    # (first step: sort into the locality order)
        scratch.synchronize()
        t_w = time.perf_counter()
        j = 2
        while (time.perf_counter() - t_w) * 1e3 < args.device_warmup_ms and j < scr_steps:
            for _ in range(8):
                if j < scr_steps:
                    scratch.run_timestep(j * dt)
                    j += 1
            scratch.synchronize()
        warm_steps, warm_ms = j - 2, (time.perf_counter() - t_w) * 1e3
    barrier()

    if not args.no_kernel_events:
        sim.profile_begin()
    batched = args.multi_step == "on"
    t0 = time.perf_counter()
    if batched:
        sim.run_timesteps(k * dt, args.steps)      # exactly K steps: t = k dt, (k + 1) dt, ...
        k += args.steps
    else:
        for _ in range(args.steps):
            sim.run_timestep(k * dt)
            k += 1
    cnt, mean, _sig = sim.grid_sums((k - 1) * dt, out=grid_out)    # gridded output + all-reduce (the caller's buffers of the warm-up output)
    barrier()
    wall = time.perf_counter() - t0
    launches, kernel_ms = (0, float("nan")) if args.no_kernel_events else sim.profile_end()

    # launches of the step kernel family bracketed per time step (one; two where module_mixing splits the step)
    launches_per_step = launches / max(args.steps, 1)
    kernel_ms_per_step = kernel_ms / max(args.steps, 1)       # their SUM per step, not a mean over unlike kernels
    per_rank_kernel_ms = [kernel_ms_per_step]
    if use_dist:
        import torch
        tw = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = tw.item()
        km = torch.zeros(world, dtype=torch.float64, device="cuda")
        km[rank] = kernel_ms_per_step
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        per_rank_kernel_ms = [float(x) for x in km.tolist()]
        kernel_ms_per_step = max(per_rank_kernel_ms)

    # outside the timed region: the step kernel with one launch per time step (what --multi-step off times)
    single_ms = None
    if not args.no_kernel_events:
        sim.set_option("multi_step", 0)
        sim.profile_begin()
        for _ in range(min(10, args.steps)):
            sim.run_timestep(k * dt)
            k += 1
        sim.synchronize()
        n1, ms1 = sim.profile_end()
        single_ms = ms1 / max(1, min(10, args.steps))

    # sanity: every particle took every step and the output grid saw all of them
    g = sim.get_atm()
    assert np.all(g["time"] == (k - 1) * dt), "not all particles advanced"
    assert np.all(np.isfinite(g["lon"])) and np.all(np.isfinite(g["p"]))
    assert int(cnt.sum()) >= 0.999 * n_total, (int(cnt.sum()), n_total)   # all ranks' particles binned

    if rank == 0:
        value = n_total * args.steps / wall
        a_per, a_state, a_met = algorithmic_bytes_per_pstep(args.workload, met0, n_local)
        bytes_per_step = a_per * n_local          # (one time step of all resident particles; a launch may take several)
        # One fused launch per step: the roofline of that kernel.  A step of several unlike kernels (module_sort,
        # module_mixing, the deposition launch: C5, C3x ...) has no single dominant launch to price -- its
        # algorithmic bytes are set against the whole step's wall time.
        one_launch = launches_per_step <= 1.0 + 1e-9 and not (ctl.get("sort_dt", 0) > 0 or "mixing_dt" in ctl)
        roof_ms = kernel_ms_per_step if one_launch else wall / args.steps * 1e3
        achieved = bytes_per_step / (roof_ms * 1e-3) / 1e9
        cache_resident = a_met <= 1.25 * 256e6
        traffic = valu_busy = fp64_frac = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            prof = json.load(open(tfile))
            if prof.get("_build_id") == build_id():      # counters of THIS build only (tools/profile.sh)
                traffic = prof.get(args.workload)
                valu_busy = prof.get("_valu_busy_frac", {}).get(args.workload)
                fp64_frac = prof.get("_fp64_valu_frac", {}).get(args.workload)
        if args.particles:      # the committed PMC profile belongs to the workload's own particle count
            traffic = valu_busy = fp64_frac = None
        other_library = os.path.basename(hip.lib_path()) != "libmptrac_hip.so"
        if other_library:       # ... and to the default build (MPTRAC_AMD_EXACT=1: the reference-rounding build; MPHIP_LIB)
            traffic = valu_busy = fp64_frac = None
        out = {
            "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: BASELINE configs[2] -- 1e7 particles/GPU, RK4 advection + "
                                   "turbulent + mesoscale diffusion + convection + sedimentation, 721x361x137 "
                                   "synthetic ERA5-shaped grid" if args.workload == "C3" else
                                   ("C3 + module_meteo every step (t, u, v, w, zg, pv, ps, pt); "
                                    + ("launched in every step" if args.eager_meteo else
                                       "scheduled every step, launched when its result can be seen -- here once, "
                                       "before the gridded output (lazy_meteo)")
                                    if args.workload == "C3m" else
                                    ("C5: C3 + module_sort + mixing every step, decay, wet and dry deposition"
                                     if args.workload == "C5" else args.workload)),
                       "particles_per_gpu": n_local, "particles_total": n_total,
                       **({"particles_override": True} if args.particles else {}),
                       # libmptrac_hip.so unless another build was asked for; the reference-rounding build
                       # (MPTRAC_AMD_EXACT=1: the CPU reference's bits) measured beside it: profiles/r06_ab_exact_library.txt
                       "library": os.path.basename(hip.lib_path()),
                       "grid": [met0.nx, met0.ny, met0.np], "dt_mod": dt,
                       "parallelism": f"index-range shards x{world}, replicated met, grid-output all-reduce",
                       "reduction": reduction, "rccl_ranks": rccl_ranks,
                       "time_loop": ("one mphip_run_timesteps call for the K steps (steps with nothing between them "
                                     "share a launch)" if batched else "K mphip_run_timestep calls"),
                       "device_warmup": (f"{warm_steps} untimed steps of a scratch copy ({warm_ms:.0f} ms) before "
                                         "the timed region: settled clocks" if scratch is not None else "none")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         # a grid that stays in the 256 MB Infinity Cache is not read from HBM every step: the
                         # figure is then no share of the HBM roof, and none is claimed
                         "frac": None if cache_resident else achieved / HBM_PEAK_GBS,
                         **({"frac_note": "null: the %.0f MB of meteo records this workload touches stay in the 256 MB "
                                          "Infinity Cache; `achieved` is an algorithmic rate, not HBM traffic" % (a_met / 1e6)}
                            if cache_resident else {}),
                         "traffic": traffic,
                         "kernel": ("step_kernel (fused time step)" if one_launch else
                                    "whole time step (step kernel, module_sort, module_mixing, deposition launch ...): "
                                    "wall time per step"),
                         "kernel_ms": roof_ms, "step_kernel_ms_per_step": kernel_ms_per_step,
                         "step_kernel_launches_per_step": launches_per_step,
                         # what a kernel trace of this command shows for the timed launches: `launch_ms` each
                         "steps_per_launch": (args.steps / launches) if launches else None,
                         "launch_ms": (kernel_ms / launches) if launches else None,
                         "kernel_ms_per_rank": per_rank_kernel_ms,
                         # the same steps as one launch per time step (rank 0, ten steps behind the timed region)
                         "kernel_ms_one_launch_per_step": single_ms,
                         # the same kernel over the W warm-up launches, i.e. on a device that comes from idle
                         # (clock ramp, profiles/r03_clock_ramp.txt); not part of `achieved`
                         "kernel_ms_from_idle": (cold_ms / max(args.warmup, 1)) if cold_launches else None,
                         "algorithmic_bytes_per_step": bytes_per_step,
                         "bytes_per_particle_step": a_per,
                         # SURVEY 8(d) caveat: the fused step is fp64-VALU-bound, not HBM-bound; share of SIMD
                         # cycles executing VALU instructions from the committed rocprofv3 PMC profile
                         "valu_busy_frac": valu_busy, "fp64_valu_frac": fp64_frac, "build_id": build_id(),
                         # `achieved` / `frac` price SURVEY 8(d)'s ALGORITHMIC bytes per particle-step against the kernel's
                         # time per step.  A launch that takes several steps per particle keeps the particle and most
                         # of its meteo lines on the chip between them and moves fewer bytes than that (`traffic`):
                         # the figure is an algorithmic throughput on the HBM scale, not a measured HBM share.  The
                         # same pricing with one launch per step:
                         "frac_one_launch_per_step": (bytes_per_step / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if single_ms and not cache_resident else None,
                         "basis": "algorithmic bytes per particle-step (SURVEY 8d) / kernel time per step",
                         # the bound that actually holds: VALU issue (modelled from measured per-instruction costs)
                         "alu": alu_roof(args.workload, n_local, kernel_ms_per_step) if not args.particles else None},
            "parity": PARITY_PINS,
        }
        dropin = os.path.join(ROOT, "profiles", "trac_dropin.json")
        if world == 1 and args.workload == "C3" and not args.particles and os.path.exists(dropin):
            out["dropin_trac"] = json.load(open(dropin))      # (a recorded measurement of the C driver at this size, labelled as such)
        exact = os.path.join(ROOT, "profiles", "exact_library.json")
        if world == 1 and args.workload == "C3" and not args.particles and not other_library and os.path.exists(exact):
            out["reference_rounding_build"] = json.load(open(exact))      # (the second library of the tree; recorded, labelled as such)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, ctl, clim, met0, met1, atm,
                                               min(args.cpu_sample, n_local), args.cpu_steps)
        if world == 1 and not use_dist and not args.no_multi_gpu_probe and not args.particles:
            probe = multi_gpu_probe(args)
            if probe is not None:
                out["multi_gpu_probe"] = probe
        print(json.dumps(out), flush=True)

    sim.close()
    if scratch is not None:
        scratch.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
