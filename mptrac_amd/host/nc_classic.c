/*
 * nc_classic.c -- reader for netCDF files in the classic formats (CDF-1, and CDF-2 with 64-bit offsets).
 *
 * The image has no netCDF library; the classic format is a flat big-endian header followed by the arrays
 * ("The NetCDF Classic Format Specification", Unidata), which is all the reference's own test meteo files
 * (tests/data/era5_utm32_*.nc) use.  Written from that specification.  netCDF-4 / HDF5 files go to nc_hdf5.c.
 *
 *   header   = magic numrecs dim_list gatt_list var_list
 *   dim      = name length                      (length 0: the record dimension)
 *   attr     = name type nelems values          (padded to 4 bytes)
 *   var      = name ndims dimid... vatt_list type vsize begin
 *   data     = fixed-size variables at `begin`; record r of a record variable at begin + r * recsize
 */
#include "nc_internal.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int type_size(int t) {
  switch (t) {
  case T_BYTE: case T_CHAR: return 1;
  case T_SHORT: return 2;
  case T_INT: case T_FLOAT: return 4;
  case T_DOUBLE: case T_INT64: return 8;
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
  case T_INT64: {
    uint64_t u = 0;
    for (int k = 0; k < 8; k++)
      u = (u << 8) | p[k];
    return (double) (int64_t) u;
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
  else if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') {
    nc->is_hdf5 = 1;
    ok = h5_load(nc);
  }
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
          if (!get_u32(nc, &id) || id >= (uint32_t) nc->ndim)
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
    h5_free_dataset(nc->var[i].h5);
  }
  free(nc->var);
  h5_free(nc);
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

int ncc_num_vars(const ncc_file *nc) {
  return nc->nvar;
}

const char *ncc_var_name(const ncc_file *nc, int var) {
  return nc->var[var].name;
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

int ncc_var_is_record(const ncc_file *nc, int var) {
  return nc->var[var].is_record;
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
  if (nc->is_hdf5)
    return h5_read_raw(nc, var, first, count, buf);
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

/* ==========================================================================================================
 * Writer: CDF-2 files (classic data model, 64-bit offsets) -- what the host layer's netCDF outputs need:
 * dimensions (one of them may be the record dimension), text attributes, variables of type int / float /
 * double, whole-variable writes and record-by-record appends.  Files written here can be read by any netCDF
 * library; the reference writes netCDF-4 (HDF5) files with the same names, dimensions and attributes.
 * ========================================================================================================== */

typedef struct {
  char *name;
  char *text;
} nccw_att;

typedef struct {
  char *name;
  int type, ndims, dimid[8];
  int natt;
  nccw_att att[8];
  int is_record;
  long long nelem;   /* elements of the whole variable, or of one record */
  long long vsize, begin;
} nccw_var;

struct nccw_file {
  FILE *f;
  char *path;
  int defined;
  int ndim, rec_dim;
  char *dim_name[16];
  long long dim_len[16];
  int natt;
  nccw_att att[16];
  int nvar;
  nccw_var var[64];
  long long recsize, rec_begin, numrecs;
  char err[256];
};

static char *dup_str(const char *s) {
  char *d = malloc(strlen(s) + 1);
  if (d)
    strcpy(d, s);
  return d;
}

nccw_file *nccw_create(const char *path) {
  nccw_file *w = calloc(1, sizeof(nccw_file));
  if (!w)
    return NULL;
  w->f = fopen(path, "w+b");
  w->path = dup_str(path);
  w->rec_dim = -1;
  if (!w->f || !w->path) {
    nccw_close(w);
    return NULL;
  }
  return w;
}

const char *nccw_error(const nccw_file *w) {
  return w ? w->err : "no file";
}

static int wfail(nccw_file *w, const char *msg) {
  snprintf(w->err, sizeof(w->err), "%s", msg);
  return -1;
}

int nccw_def_dim(nccw_file *w, const char *name, long long len) {
  if (w->defined || w->ndim >= 16)
    return wfail(w, "cannot define another dimension");
  if (len == 0) {
    if (w->rec_dim >= 0)
      return wfail(w, "only one record dimension");
    w->rec_dim = w->ndim;
  }
  w->dim_name[w->ndim] = dup_str(name);
  w->dim_len[w->ndim] = len;
  return w->ndim++;
}

int nccw_def_var(nccw_file *w, const char *name, int type, int ndims, const int *dimids) {
  if (w->defined || w->nvar >= 64 || ndims > 8 || !type_size(type))
    return wfail(w, "cannot define another variable");
  nccw_var *v = &w->var[w->nvar];
  memset(v, 0, sizeof(*v));
  v->name = dup_str(name);
  v->type = type;
  v->ndims = ndims;
  v->nelem = 1;
  for (int d = 0; d < ndims; d++) {
    if (dimids[d] < 0 || dimids[d] >= w->ndim)
      return wfail(w, "unknown dimension");
    v->dimid[d] = dimids[d];
    if (dimids[d] == w->rec_dim) {
      if (d != 0)
        return wfail(w, "the record dimension must come first");
      v->is_record = 1;
    } else
      v->nelem *= w->dim_len[dimids[d]];
  }
  return w->nvar++;
}

int nccw_put_att_text(nccw_file *w, int var, const char *name, const char *text) {
  if (w->defined)
    return wfail(w, "attributes belong to the definition phase");
  nccw_att *a;
  if (var < 0) {
    if (w->natt >= 16)
      return wfail(w, "too many global attributes");
    a = &w->att[w->natt++];
  } else {
    if (var >= w->nvar || w->var[var].natt >= 8)
      return wfail(w, "too many attributes");
    a = &w->var[var].att[w->var[var].natt++];
  }
  a->name = dup_str(name);
  a->text = dup_str(text);
  return 0;
}

static void put_u32(FILE *f, uint32_t v) {
  const unsigned char b[4] = { (unsigned char) (v >> 24), (unsigned char) (v >> 16), (unsigned char) (v >> 8),
                               (unsigned char) v };
  fwrite(b, 1, 4, f);
}

static void put_u64(FILE *f, uint64_t v) {
  put_u32(f, (uint32_t) (v >> 32));
  put_u32(f, (uint32_t) v);
}

static void put_padded(FILE *f, const char *s, size_t n) {
  static const char zero[4] = { 0, 0, 0, 0 };
  fwrite(s, 1, n, f);
  fwrite(zero, 1, (4 - n % 4) % 4, f);
}

static void put_name(FILE *f, const char *s) {
  put_u32(f, (uint32_t) strlen(s));
  put_padded(f, s, strlen(s));
}

static void put_atts(FILE *f, int n, const nccw_att *a) {
  put_u32(f, n ? TAG_ATT : 0);
  put_u32(f, (uint32_t) n);
  for (int i = 0; i < n; i++) {
    put_name(f, a[i].name);
    put_u32(f, T_CHAR);
    put_u32(f, (uint32_t) strlen(a[i].text));
    put_padded(f, a[i].text, strlen(a[i].text));
  }
}

static size_t name_bytes(const char *s) {
  return 4 + ((strlen(s) + 3) & ~(size_t) 3);
}

static size_t atts_bytes(int n, const nccw_att *a) {
  size_t b = 8;
  for (int i = 0; i < n; i++)
    b += name_bytes(a[i].name) + 8 + ((strlen(a[i].text) + 3) & ~(size_t) 3);
  return b;
}

static void write_header(nccw_file *w) {
  FILE *f = w->f;
  fseeko(f, 0, SEEK_SET);
  fwrite("CDF\002", 1, 4, f);
  put_u32(f, (uint32_t) w->numrecs);
  put_u32(f, w->ndim ? TAG_DIM : 0);
  put_u32(f, (uint32_t) w->ndim);
  for (int d = 0; d < w->ndim; d++) {
    put_name(f, w->dim_name[d]);
    put_u32(f, (uint32_t) w->dim_len[d]);
  }
  put_atts(f, w->natt, w->att);
  put_u32(f, w->nvar ? TAG_VAR : 0);
  put_u32(f, (uint32_t) w->nvar);
  for (int i = 0; i < w->nvar; i++) {
    const nccw_var *v = &w->var[i];
    put_name(f, v->name);
    put_u32(f, (uint32_t) v->ndims);
    for (int d = 0; d < v->ndims; d++)
      put_u32(f, (uint32_t) v->dimid[d]);
    put_atts(f, v->natt, v->att);
    put_u32(f, (uint32_t) v->type);
    put_u32(f, v->vsize > 0xffffffffLL ? 0xffffffffu : (uint32_t) v->vsize);
    put_u64(f, (uint64_t) v->begin);
  }
}

int nccw_enddef(nccw_file *w) {
  if (w->defined)
    return 0;
  size_t header = 4 + 4 + 8;
  for (int d = 0; d < w->ndim; d++)
    header += name_bytes(w->dim_name[d]) + 4;
  header += atts_bytes(w->natt, w->att) + 8;
  for (int i = 0; i < w->nvar; i++)
    header += name_bytes(w->var[i].name) + 4 + 4 * (size_t) w->var[i].ndims + atts_bytes(w->var[i].natt, w->var[i].att)
      + 4 + 4 + 8;
  long long at = (long long) header;
  for (int i = 0; i < w->nvar; i++) {   /* fixed-size variables first, in order of definition */
    nccw_var *v = &w->var[i];
    v->vsize = (v->nelem * type_size(v->type) + 3) & ~3LL;
    if (!v->is_record) {
      v->begin = at;
      at += v->vsize;
    }
  }
  w->rec_begin = at;
  w->recsize = 0;
  for (int i = 0; i < w->nvar; i++)
    if (w->var[i].is_record) {
      w->var[i].begin = at + w->recsize;
      w->recsize += w->var[i].vsize;
    }
  w->defined = 1;
  write_header(w);
  /* the file has its final fixed-size extent from the start (unwritten variables read as zeros) */
  if (at > (long long) header) {
    fseeko(w->f, (off_t) at - 1, SEEK_SET);
    fputc(0, w->f);
  }
  return ferror(w->f) ? wfail(w, "cannot write the netCDF header") : 0;
}

static void encode(unsigned char *p, int type, double x) {
  if (type == T_DOUBLE) {
    uint64_t u;
    memcpy(&u, &x, 8);
    for (int k = 0; k < 8; k++)
      p[k] = (unsigned char) (u >> (56 - 8 * k));
  } else if (type == T_FLOAT) {
    const float fl = (float) x;
    uint32_t u;
    memcpy(&u, &fl, 4);
    for (int k = 0; k < 4; k++)
      p[k] = (unsigned char) (u >> (24 - 8 * k));
  } else if (type == T_INT) {
    const uint32_t u = (uint32_t) (int32_t) x;
    for (int k = 0; k < 4; k++)
      p[k] = (unsigned char) (u >> (24 - 8 * k));
  } else if (type == T_SHORT) {
    const uint16_t u = (uint16_t) (int16_t) x;
    p[0] = (unsigned char) (u >> 8);
    p[1] = (unsigned char) u;
  } else
    p[0] = (unsigned char) (signed char) x;
}

/* values of record `rec` (record variables; appends when rec == number of records so far) or of the whole
 * variable, converted to the variable's type */
int nccw_put_double(nccw_file *w, int var, long long rec, const double *data) {
  if (!w->defined || var < 0 || var >= w->nvar)
    return wfail(w, "bad variable");
  const nccw_var *v = &w->var[var];
  const int ts = type_size(v->type);
  long long at = v->begin;
  if (v->is_record) {
    if (rec < 0 || rec > w->numrecs)
      return wfail(w, "records are appended one after the other");
    at += rec * w->recsize;
  }
  enum { CHUNK = 65536 };
  unsigned char *buf = malloc((size_t) CHUNK * 8);
  if (!buf)
    return wfail(w, "out of memory");
  fseeko(w->f, (off_t) at, SEEK_SET);
  for (long long i0 = 0; i0 < v->nelem; i0 += CHUNK) {
    const long long m = v->nelem - i0 < CHUNK ? v->nelem - i0 : CHUNK;
    for (long long i = 0; i < m; i++)
      encode(buf + i * ts, v->type, data[i0 + i]);
    fwrite(buf, (size_t) ts, (size_t) m, w->f);
  }
  free(buf);
  const long long pad = v->vsize - v->nelem * ts;
  for (long long k = 0; k < pad; k++)
    fputc(0, w->f);
  if (v->is_record && rec == w->numrecs) {
    w->numrecs = rec + 1;
    /* the other record variables of this record exist (zeros) even if they are never written */
    fseeko(w->f, (off_t) (w->rec_begin + w->numrecs * w->recsize - 1), SEEK_SET);
    int c = fgetc(w->f);
    if (c == EOF) {
      fseeko(w->f, (off_t) (w->rec_begin + w->numrecs * w->recsize - 1), SEEK_SET);
      fputc(0, w->f);
    }
    fseeko(w->f, 4, SEEK_SET);
    put_u32(w->f, (uint32_t) w->numrecs);
  }
  return ferror(w->f) ? wfail(w, "write error") : 0;
}

int nccw_put_int(nccw_file *w, int var, long long rec, const int *data) {
  if (!w->defined || var < 0 || var >= w->nvar)
    return wfail(w, "bad variable");
  const long long n = w->var[var].nelem;
  double *tmp = malloc((size_t) (n > 0 ? n : 1) * sizeof(double));
  if (!tmp)
    return wfail(w, "out of memory");
  for (long long i = 0; i < n; i++)
    tmp[i] = data[i];
  const int rc = nccw_put_double(w, var, rec, tmp);
  free(tmp);
  return rc;
}

int nccw_find_var(const nccw_file *w, const char *name) {
  for (int i = 0; i < w->nvar; i++)
    if (strcmp(w->var[i].name, name) == 0)
      return i;
  return -1;
}

long long nccw_numrecs(const nccw_file *w) {
  return w->numrecs;
}

int nccw_close(nccw_file *w) {
  if (!w)
    return 0;
  int rc = 0;
  if (w->f) {
    if (!w->defined)
      nccw_enddef(w);
    rc = ferror(w->f) || fclose(w->f) ? -1 : 0;
  }
  for (int d = 0; d < w->ndim; d++)
    free(w->dim_name[d]);
  for (int i = 0; i < w->natt; i++) {
    free(w->att[i].name);
    free(w->att[i].text);
  }
  for (int i = 0; i < w->nvar; i++) {
    free(w->var[i].name);
    for (int k = 0; k < w->var[i].natt; k++) {
      free(w->var[i].att[k].name);
      free(w->var[i].att[k].text);
    }
  }
  free(w->path);
  free(w);
  return rc;
}
