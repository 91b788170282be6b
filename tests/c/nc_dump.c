/* Test program (tests/test_host_logic.py): header and values of a netCDF file as the host layer's reader sees it
 * (classic formats and netCDF-4 / HDF5).
 *   nc_dump <file> [variable ...]
 * prints "header <name> <axis>=<length> ..." for every variable, "dim <name> <length>", "var <name> <ndims> <dim names...> <nelem>", "att <var|-> <name> <first value>", and for
 * each named variable "values <name>" followed by all its values (17 significant digits). */
#include "nc_classic.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char *argv[]) {
  char why[256];
  if (argc < 2)
    return 2;
  ncc_file *nc = ncc_open(argv[1], why, sizeof(why));
  if (!nc) {
    printf("ERROR %s\n", why);
    return 1;
  }
  for (int v = 0; v < ncc_num_vars(nc); v++) {   /* every variable with its axes */
    printf("header %s", ncc_var_name(nc, v));
    for (int d = 0; d < ncc_var_ndims(nc, v); d++) {
      const char *name;
      const long long len = ncc_var_dim(nc, v, d, &name);
      printf(" %s=%lld", name, len);
    }
    printf("\n");
  }
  for (int a = 2; a < argc; a++) {
    const int var = ncc_find_var(nc, argv[a]);
    if (var < 0) {
      printf("missing %s\n", argv[a]);
      continue;
    }
    const int nd = ncc_var_ndims(nc, var);
    long long n = 1;
    printf("var %s %d", argv[a], nd);
    for (int d = 0; d < nd; d++) {
      const char *name;
      const long long len = ncc_var_dim(nc, var, d, &name);
      printf(" %s=%lld", name, len);
      n *= len;
    }
    printf("\n");
    static const char *atts[] = { "scale_factor", "add_offset", "_FillValue", "missing_value", "_Netcdf4Dimid" };
    for (int k = 0; k < 5; k++) {
      double x;
      if (ncc_get_att(nc, var, atts[k], &x))
        printf("att %s %s %.17g\n", argv[a], atts[k], x);
    }
    double *x = n < (1LL << 28) ? malloc((size_t) (n ? n : 1) * sizeof(double)) : NULL;
    if (!x) {
      printf("ERROR variable too large for this tool\n");
      return 1;
    }
    if (!ncc_read_double(nc, var, 0, 0, n, x)) {
      printf("ERROR %s\n", ncc_error(nc));
      return 1;
    }
    printf("values %s", argv[a]);
    for (long long i = 0; i < n; i++)
      printf(" %.17g", x[i]);
    printf("\n");
    free(x);
  }
  static const char *dims[] = { "time", "press", "lat", "lon", "lev", "obs", "NPARTS", "level" };
  for (int k = 0; k < 8; k++) {
    long long len;
    if (ncc_find_dim(nc, dims[k], &len) >= 0)
      printf("dim %s %lld\n", dims[k], len);
  }
  ncc_close(nc);
  printf("RESULT done\n");
  return 0;
}
