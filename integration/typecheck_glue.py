#!/usr/bin/env python3
"""Type-check of route A: compiles integration/mptrac_hip_glue.c (syntax and types only, `gcc -fsyntax-only`) in
one translation unit behind the reference's own header /root/reference/src/mptrac.h, the way a maintainer would
include it at the end of src/mptrac.c.

The reference's header includes GSL and netCDF headers this image does not have.  Its declarations use none of
their types, so for this check -- and for nothing else: no object file is produced, nothing of the reference is
built or run -- the compiler is pointed at EMPTY files of those names in a temporary directory.  `rng_ctr`, a
file-scope variable of src/mptrac.c (mptrac.c:34-35) the glue hands to the back end, is declared the way that file
declares it.

  python integration/typecheck_glue.py [reference root]     exit code 0 = every expression of the glue type-checks
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABSENT_HEADERS = ("gsl/gsl_fft_complex.h", "gsl/gsl_math.h", "gsl/gsl_randist.h", "gsl/gsl_rng.h", "gsl/gsl_sort.h",
                  "gsl/gsl_spline.h", "gsl/gsl_statistics.h", "netcdf.h")

UNIT = """#include "mptrac.h"
static uint64_t rng_ctr;          /* src/mptrac.c:34-35 */
#define MPTRAC_HIP
#include "mptrac_hip_glue.c"
"""


def typecheck(reference="/root/reference"):
    """Returns (ok, compiler output)."""
    src = os.path.join(reference, "src")
    if not os.path.exists(os.path.join(src, "mptrac.h")):
        raise FileNotFoundError(src)
    with tempfile.TemporaryDirectory() as tmp:
        for h in ABSENT_HEADERS:
            os.makedirs(os.path.dirname(os.path.join(tmp, h)), exist_ok=True)
            open(os.path.join(tmp, h), "w").close()
        with open(os.path.join(tmp, "unit.c"), "w") as f:
            f.write(UNIT)
        cmd = ["gcc", "-fsyntax-only", "-std=gnu99", "-fopenmp", "-Wall", "-Wextra", "-Wno-unused-function",
               "-Wno-unused-variable", "-Wno-unused-parameter", "-I" + tmp, "-I" + src, "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "integration"), os.path.join(tmp, "unit.c")]
        r = subprocess.run(cmd, capture_output=True, text=True)
    return r.returncode == 0, r.stdout + r.stderr


if __name__ == "__main__":
    ok, out = typecheck(*sys.argv[1:2])
    sys.stdout.write(out)
    print("route-A glue type-checks against the reference's mptrac.h" if ok else "route-A glue does NOT type-check")
    sys.exit(0 if ok else 1)
