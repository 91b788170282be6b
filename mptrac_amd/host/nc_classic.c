/*
 * nc_classic.c -- reader for netCDF files in the classic formats (CDF-1, and CDF-2 with 64-bit offsets).
 *
 * The image has no netCDF library; the classic format is a flat big-endian header followed by the arrays
 * ("The NetCDF Classic Format Specification", Unidata), which is all the reference's own test meteo files
 * (tests/data/era5_utm32_*.nc) use.  Written from that specification.  netCDF-4 / HDF5 files are rejected.
 *
 *   header   = magic numrecs dim_list gatt_list var_list
 *   dim      = name length                      (length 0: the record dimension)
 *   attr     = name type nelems values          (padded to 4 bytes)
 *   var      = name ndims dimid... vatt_list type vsize begin
 *   data     = fixed-size variables at `begin`; record r of a record variable at begin + r * recsize
 */
#include "nc_classic.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { TAG_DIM = 10, TAG_VAR = 11, TAG_ATT = 12 };
enum { T_BYTE = 1, T_CHAR = 2, T_SHORT = 3, T_INT = 4, T_FLOAT = 5, T_DOUBLE = 6 };

typedef struct {
  char *name;
  int type;
  size_t n;
  unsigned char *raw;   /* big-endian values as stored */
} ncc_att;

typedef struct {
  char *name;
  int ndims, dimid[8];
  int natt;
  ncc_att *att;
  int type;
  long long vsize, begin;
  int is_record;
  long long nelem;      /* elements of one record (record variables) or of the whole variable */
} ncc_var;

struct ncc_file {
  FILE *f;
  int version;
  long long numrecs, recsize;
  int ndim;
  char **dim_name;
  long long *dim_len;
  int natt;
  ncc_att *att;
  int nvar;
  ncc_var *var;
  char err[256];
};

static int type_size(int t) {
  switch (t) {
  case T_BYTE: case T_CHAR: return 1;
  case T_SHORT: return 2;
  case T_INT: case T_FLOAT: return 4;
  case T_DOUBLE: return 8;
  }
  return 0;
}

static int fail(ncc_file *nc, const char *msg) {
  snprintf(nc->err, sizeof(nc->err), "%s", msg);
  return 0;
}

static int get_u32(ncc_file *nc, uint32_t *v) {
  unsigned char b[4];
  if (fread(b, 1, 4, nc->f) != 4)
    return fail(nc, "unexpected end of the netCDF header");
  *v = ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | b[3];
  return 1;
}

static int get_u64(ncc_file *nc, uint64_t *v) {
  uint32_t hi, lo;
  if (!get_u32(nc, &hi) || !get_u32(nc, &lo))
    return 0;
  *v = ((uint64_t) hi << 32) | lo;
  return 1;
}

static int get_name(ncc_file *nc, char **out) {
  uint32_t n;
  if (!get_u32(nc, &n) || n > 65536)
    return fail(nc, "bad name length in the netCDF header");
  const size_t padded = ((size_t) n + 3) & ~(size_t) 3;
  char *s = calloc(padded + 1, 1);
  if (!s || fread(s, 1, padded, nc->f) != padded) {
    free(s);
    return fail(nc, "unexpected end of the netCDF header");
  }
  s[n] = '\0';
  *out = s;
  return 1;
}

static int get_atts(ncc_file *nc, int *natt, ncc_att **att) {
  uint32_t tag, n;
  if (!get_u32(nc, &tag) || !get_u32(nc, &n))
    return 0;
  *natt = 0;
  *att = NULL;
  if (tag == 0 && n == 0)
    return 1;
  if (tag != TAG_ATT || n > 100000)
    return fail(nc, "malformed attribute list");
  *att = calloc(n ? n : 1, sizeof(ncc_att));
  if (!*att)
    return fail(nc, "out of memory");
  for (uint32_t i = 0; i < n; i++) {
    ncc_att *a = &(*att)[i];
    uint32_t type, nelems;
    if (!get_name(nc, &a->name) || !get_u32(nc, &type) || !get_u32(nc, &nelems))
      return 0;
    if (!type_size((int) type))
      return fail(nc, "attribute of unknown type");
    a->type = (int) type;
    a->n = nelems;
    const size_t bytes = (size_t) nelems * (size_t) type_size(a->type), padded = (bytes + 3) & ~(size_t) 3;
    a->raw = calloc(padded + 8, 1);
    if (!a->raw || fread(a->raw, 1, padded, nc->f) != padded)
      return fail(nc, "unexpected end of the netCDF header");
    (*natt)++;
  }
  return 1;
}

static double decode(const unsigned char *p, int type) {
  switch (type) {
  case T_BYTE: return (double) (signed char) p[0];
  case T_CHAR: return (double) p[0];
  case T_SHORT: return (double) (int16_t) (((uint16_t) p[0] << 8) | p[1]);
  case T_INT: return (double) (int32_t) (((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]);
  case T_FLOAT: {
    const uint32_t u = ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3];
    float f;
    memcpy(&f, &u, 4);
    return (double) f;
  }
  case T_DOUBLE: {
    uint64_t u = 0;
    for (int k = 0; k < 8; k++)
      u = (u << 8) | p[k];
    double d;
    memcpy(&d, &u, 8);
    return d;
  }
  }
  return 0.0;
}

ncc_file *ncc_open(const char *path, char *err, size_t errlen) {
  ncc_file *nc = calloc(1, sizeof(*nc));
  if (!nc)
    return NULL;
  int ok = 0;
  unsigned char magic[4];
  uint32_t tag, n;
  if (!(nc->f = fopen(path, "rb")))
    fail(nc, "cannot open file");
  else if (fread(magic, 1, 4, nc->f) != 4)
    fail(nc, "file too short");
  else if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F')
    fail(nc, "netCDF-4 / HDF5 file: only the classic formats (CDF-1, CDF-2) are read by this build");
  else if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2))
    fail(nc, "not a classic netCDF file (CDF-1 / CDF-2)");
  else {
    nc->version = magic[3];
    uint32_t numrecs;
    do {
      if (!get_u32(nc, &numrecs))
        break;
      nc->numrecs = numrecs == 0xffffffffu ? -1 : (long long) numrecs;
      /* dimensions */
      if (!get_u32(nc, &tag) || !get_u32(nc, &n))
        break;
      if (!((tag == 0 && n == 0) || (tag == TAG_DIM && n < 100000))) {
        fail(nc, "malformed dimension list");
        break;
      }
      nc->dim_name = calloc(n ? n : 1, sizeof(char *));
      nc->dim_len = calloc(n ? n : 1, sizeof(long long));
      int bad = 0;
      for (uint32_t i = 0; i < n && !bad; i++) {
        uint32_t len;
        if (!get_name(nc, &nc->dim_name[i]) || !get_u32(nc, &len))
          bad = 1;
        else {
          nc->dim_len[i] = len;
          nc->ndim++;
        }
      }
      if (bad || !get_atts(nc, &nc->natt, &nc->att))
        break;
      /* variables */
      if (!get_u32(nc, &tag) || !get_u32(nc, &n))
        break;
      if (!((tag == 0 && n == 0) || (tag == TAG_VAR && n < 100000))) {
        fail(nc, "malformed variable list");
        break;
      }
      nc->var = calloc(n ? n : 1, sizeof(ncc_var));
      for (uint32_t i = 0; i < n && !bad; i++) {
        ncc_var *v = &nc->var[i];
        uint32_t nd, type, vsize;
        if (!get_name(nc, &v->name) || !get_u32(nc, &nd) || nd > 8) {
          bad = 1;
          break;
        }
        v->ndims = (int) nd;
        v->nelem = 1;
        for (uint32_t d = 0; d < nd && !bad; d++) {
          uint32_t id;
          if (!get_u32(nc, &id) || (int) id >= nc->ndim)
            bad = 1;
          else {
            v->dimid[d] = (int) id;
            if (nc->dim_len[id] == 0 && d == 0)
              v->is_record = 1;
            else
              v->nelem *= nc->dim_len[id];
          }
        }
        if (bad || !get_atts(nc, &v->natt, &v->att) || !get_u32(nc, &type) || !get_u32(nc, &vsize)) {
          bad = 1;
          break;
        }
        if (!type_size((int) type)) {
          fail(nc, "variable of unknown type");
          bad = 1;
          break;
        }
        v->type = (int) type;
        v->vsize = vsize;
        if (nc->version == 1) {
          uint32_t b;
          if (!get_u32(nc, &b))
            bad = 1;
          v->begin = b;
        } else {
          uint64_t b = 0;
          if (!get_u64(nc, &b))
            bad = 1;
          v->begin = (long long) b;
        }
        nc->nvar++;
      }
      if (bad)
        break;
      /* size of one record: the sum of the record variables' vsize -- except that a single record variable
       * is stored without padding between its records */
      int nrec = 0;
      for (int i = 0; i < nc->nvar; i++)
        if (nc->var[i].is_record) {
          nc->recsize += nc->var[i].vsize;
          nrec++;
        }
      if (nrec == 1)
        for (int i = 0; i < nc->nvar; i++)
          if (nc->var[i].is_record)
            nc->recsize = nc->var[i].nelem * type_size(nc->var[i].type);
      if (nc->numrecs < 0) {   /* "streaming" files: derive the record count from the file size */
        fail(nc, "netCDF file without a record count (streaming mode) is not supported");
        break;
      }
      ok = 1;
    } while (0);
  }
  if (!ok) {
    if (err && errlen)
      snprintf(err, errlen, "%s", nc->err[0] ? nc->err : "malformed netCDF header");
    ncc_close(nc);
    return NULL;
  }
  return nc;
}

static void free_atts(int n, ncc_att *a) {
  for (int i = 0; i < n; i++) {
    free(a[i].name);
    free(a[i].raw);
  }
  free(a);
}

void ncc_close(ncc_file *nc) {
  if (!nc)
    return;
  if (nc->f)
    fclose(nc->f);
  for (int i = 0; i < nc->ndim; i++)
    free(nc->dim_name[i]);
  free(nc->dim_name);
  free(nc->dim_len);
  free_atts(nc->natt, nc->att);
  for (int i = 0; i < nc->nvar; i++) {
    free(nc->var[i].name);
    free_atts(nc->var[i].natt, nc->var[i].att);
  }
  free(nc->var);
  free(nc);
}

const char *ncc_error(const ncc_file *nc) {
  return nc->err;
}

int ncc_find_dim(const ncc_file *nc, const char *name, long long *len) {
  for (int i = 0; i < nc->ndim; i++)
    if (strcmp(nc->dim_name[i], name) == 0) {
      if (len)
        *len = nc->dim_len[i] == 0 ? nc->numrecs : nc->dim_len[i];
      return i;
    }
  return -1;
}

int ncc_find_var(const ncc_file *nc, const char *name) {
  for (int i = 0; i < nc->nvar; i++)
    if (strcmp(nc->var[i].name, name) == 0)
      return i;
  return -1;
}

int ncc_var_ndims(const ncc_file *nc, int var) {
  return nc->var[var].ndims;
}

long long ncc_var_dim(const ncc_file *nc, int var, int d, const char **name) {
  const int id = nc->var[var].dimid[d];
  if (name)
    *name = nc->dim_name[id];
  return nc->dim_len[id] == 0 ? nc->numrecs : nc->dim_len[id];
}

int ncc_var_is_packed(const ncc_file *nc, int var) {
  return nc->var[var].type == T_SHORT || nc->var[var].type == T_BYTE;
}

int ncc_get_att(const ncc_file *nc, int var, const char *name, double *value) {
  const int n = var < 0 ? nc->natt : nc->var[var].natt;
  const ncc_att *a = var < 0 ? nc->att : nc->var[var].att;
  for (int i = 0; i < n; i++)
    if (strcmp(a[i].name, name) == 0 && a[i].n >= 1 && a[i].type != T_CHAR) {
      *value = decode(a[i].raw, a[i].type);
      return 1;
    }
  return 0;
}

/* elements [first, first + count) of record `rec` (record variables) or of the whole variable (rec ignored) */
static int read_raw(ncc_file *nc, int var, long long rec, long long first, long long count, unsigned char **buf) {
  ncc_var *v = &nc->var[var];
  if (first < 0 || count < 0 || first + count > v->nelem)
    return fail(nc, "read beyond the end of a netCDF variable");
  if (v->is_record && (rec < 0 || rec >= nc->numrecs))
    return fail(nc, "record index out of range");
  const int ts = type_size(v->type);
  const long long off = v->begin + (v->is_record ? rec * nc->recsize : 0) + first * ts;
  *buf = malloc((size_t) (count * ts) + 8);
  if (!*buf)
    return fail(nc, "out of memory");
  if (fseeko(nc->f, (off_t) off, SEEK_SET) != 0 || fread(*buf, (size_t) ts, (size_t) count, nc->f) != (size_t) count) {
    free(*buf);
    *buf = NULL;
    return fail(nc, "unexpected end of netCDF data");
  }
  return 1;
}

int ncc_read_double(ncc_file *nc, int var, long long rec, long long first, long long count, double *out) {
  unsigned char *buf;
  if (!read_raw(nc, var, rec, first, count, &buf))
    return 0;
  const int type = nc->var[var].type, ts = type_size(type);
  for (long long i = 0; i < count; i++)
    out[i] = decode(buf + i * ts, type);
  free(buf);
  return 1;
}

int ncc_read_float(ncc_file *nc, int var, long long rec, long long first, long long count, float *out) {
  unsigned char *buf;
  if (!read_raw(nc, var, rec, first, count, &buf))
    return 0;
  const int type = nc->var[var].type, ts = type_size(type);
  if (type == T_FLOAT)
    for (long long i = 0; i < count; i++) {
      const unsigned char *p = buf + 4 * i;
      const uint32_t u = ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3];
      memcpy(&out[i], &u, 4);
    }
  else
    for (long long i = 0; i < count; i++)
      out[i] = (float) decode(buf + i * ts, type);
  free(buf);
  return 1;
}

int ncc_read_short(ncc_file *nc, int var, long long rec, long long first, long long count, short *out) {
  if (nc->var[var].type != T_SHORT)
    return fail(nc, "variable is not of type short");
  unsigned char *buf;
  if (!read_raw(nc, var, rec, first, count, &buf))
    return 0;
  for (long long i = 0; i < count; i++)
    out[i] = (short) (int16_t) (((uint16_t) buf[2 * i] << 8) | buf[2 * i + 1]);
  free(buf);
  return 1;
}
