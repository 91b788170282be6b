"""For a failing seed of tests/test_gpu_fuzz.py: the first step at which cache->uvwp differs from the oracle's, and
for that particle the state of both sides one step earlier (were the inputs of the step the same bits?).
    [MPTRAC_FUZZ_PARTICLES=n] python tools/gpu_fuzz_divergence.py <seed> [...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_fuzz as F          # noqa: E402


def ulps(a, b):
    return float(abs(a - b) / np.spacing(abs(b))) if b != 0 else float(abs(a - b))


def main(seed):
    ctl, names, o, s, times = F._contexts(seed)
    keys = {k: v for k, v in ctl.items() if k.startswith(("turb", "conv", "advect", "dt_mod", "sort", "mixing", "direction"))}
    print("seed", seed, keys)
    prev = None
    for k, t in enumerate(times):
        o.run_timestep(t)
        s.run_timestep(t)
        g, r = s.state(), o.state()
        d = np.abs(g["uvwp"].astype(np.float64) - r["uvwp"])
        nd = {q: int(np.count_nonzero(g[q] != r[q])) for q in ("lon", "lat", "p")}
        print(f"step {k}: t = {t}, positions that differ in any bit {nd}, uvwp values that differ {int((d > 0).sum())}")
        if d.max() > 0:
            for i, c in zip(*np.nonzero(d)):
                print(f"  particle {i} component {c}: gpu {g['uvwp'][i, c]!r} oracle {r['uvwp'][i, c]!r}")
                for q in ("lon", "lat", "p"):
                    print(f"    {q}: now gpu {g[q][i]!r} oracle {r[q][i]!r} ({ulps(g[q][i], r[q][i]):.1f} ulp)", end="")
                    if prev:
                        print(f"; before {prev[0][q][i]!r} / {prev[1][q][i]!r} ({ulps(prev[0][q][i], prev[1][q][i]):.1f} ulp)")
                    else:
                        print()
                if prev:
                    print("    uvwp before: gpu", prev[0]["uvwp"][i], "oracle", prev[1]["uvwp"][i])
            break
        prev = (g, r)
    s.close()


if __name__ == "__main__":
    for a in sys.argv[1:]:
        main(int(a))
