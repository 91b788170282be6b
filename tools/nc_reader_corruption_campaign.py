#!/usr/bin/env python3
"""Robustness campaign for the host layer's netCDF readers (nc_classic.c, nc_hdf5.c): random byte flips,
truncations and 8-byte all-ones / all-zero stamps in the header region of sample files, read by tests/c/nc_dump.c
built with -fsanitize=address,undefined.  Every run has to end with exit code 0 or 1 and no sanitizer report.
  gcc -O1 -g -std=gnu99 -fsanitize=address,undefined -I mptrac_amd/host -o /tmp/nc_dump_san tests/c/nc_dump.c \\
      mptrac_amd/host/nc_classic.c mptrac_amd/host/nc_hdf5.c -lz -lm
  python tools/nc_reader_corruption_campaign.py /tmp/nc_dump_san [trials per file and seed]"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))


def sample_files(tmp="/tmp"):
    import numpy as np
    import h5write
    rng = np.random.default_rng(12)
    w = h5write.Writer()
    w.dataset("t", rng.normal(250.0, 20.0, (5, 13, 17)).astype("<f4"), chunks=(2, 5, 8), shuffle=True, deflate=4)
    w.dataset("q", rng.integers(-30000, 30000, (7, 10)).astype("<i2"), chunks=(4, 4), deflate=1,
              attrs=(("scale_factor", np.float64(0.25)),))
    w.dataset("lev", np.arange(5.0))
    w.close(os.path.join(tmp, "old_style.nc"))
    w = h5write.Writer("new")
    w.dataset("t", rng.normal(250.0, 20.0, (5, 13, 17)).astype("<f4"), chunks=(2, 5, 8), shuffle=True, deflate=4)
    w.dataset("q", rng.integers(-30000, 30000, (7, 10)).astype("<i2"), chunks=(4, 4), deflate=1,
              attrs=(("scale_factor", np.float64(0.25)),))
    w.dataset("lev", np.arange(5.0))
    w.close(os.path.join(tmp, "new_style.nc"))
    g = os.path.join(os.path.dirname(HERE), "tests", "golden")
    return {os.path.join(tmp, "old_style.nc"): ["t", "q", "lev"],
            os.path.join(tmp, "new_style.nc"): ["t", "q", "lev"],
            "/root/reference/data/tuv_photolysis_rates.nc": ["press", "o2"],
            os.path.join(g, "ref_dd_test", "init", "data.0.nc"): ["LON", "LAT", "idx", "time"],
            os.path.join(g, "ref_data", "cams_H2O2.nc"): ["H2O2", "press"],
            os.path.join(g, "ref_data", "gozcards_HNO3.nc"): ["HNO3", "press"],
            os.path.join(g, "ref_coord_test", "era5_utm32_2025_05_01_00.nc"): ["t", "lev"]}


def main():
    exe = sys.argv[1]
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1", UBSAN_OPTIONS="halt_on_error=1")
    bad = total = 0
    for seed in (7, 21, 99, 1234, 4321):
        random.seed(seed)
        for path, names in sample_files().items():
            if not os.path.exists(path):      # (the reference's dense-group file: this container only)
                continue
            data = bytearray(open(path, "rb").read())
            hdr = min(len(data), 16000)
            for trial in range(trials):
                d = bytearray(data)
                if trial % 3 == 0:
                    for _ in range(random.randint(1, 4)):
                        d[random.randrange(hdr)] = random.randrange(256)
                elif trial % 3 == 1:
                    d = d[:random.randrange(8, len(d))]
                else:
                    i = random.randrange(hdr - 8)
                    d[i:i + 8] = bytes([0xff] * 8) if random.random() < 0.5 else bytes(8)
                open("/tmp/corrupt.nc", "wb").write(d)
                try:
                    r = subprocess.run([exe, "/tmp/corrupt.nc"] + names, capture_output=True, timeout=30, env=env)
                except subprocess.TimeoutExpired:
                    r = None
                total += 1
                err = r.stderr.decode(errors="replace") if r else "TIMEOUT"
                if r is None or r.returncode not in (0, 1) or "ERROR: AddressSanitizer" in err or "runtime error" in err:
                    bad += 1
                    open("/tmp/bad_%d.nc" % bad, "wb").write(d)
                    print("FAILURE", path, seed, trial, err[:600].replace("\n", " | "))
    print("runs", total, "failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
