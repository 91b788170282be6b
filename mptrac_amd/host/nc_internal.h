/* nc_internal.h -- the in-memory description of an open netCDF file, shared by the classic-format reader
 * (nc_classic.c) and the netCDF-4 / HDF5 reader (nc_hdf5.c) */
#ifndef MPTRAC_AMD_NC_INTERNAL_H
#define MPTRAC_AMD_NC_INTERNAL_H
#include "nc_classic.h"

#include <stdint.h>
#include <stdio.h>

enum { TAG_DIM = 10, TAG_VAR = 11, TAG_ATT = 12 };
enum { T_BYTE = 1, T_CHAR = 2, T_SHORT = 3, T_INT = 4, T_FLOAT = 5, T_DOUBLE = 6,
       T_INT64 = 10 /* netCDF-4 only */ };

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
  struct h5_dataset *h5;   /* netCDF-4: where and how the values are stored (nc_hdf5.c) */
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
  int is_hdf5;
  struct h5_file *h5;    /* netCDF-4: reader state (nc_hdf5.c) */
};


/* nc_hdf5.c: fills dimensions, variables and attributes of a netCDF-4 file; 0 + nc->err on failure */
int h5_load(ncc_file *nc);
void h5_free(ncc_file *nc);
void h5_free_dataset(struct h5_dataset *d);
/* elements [first, first + count) of a variable as big-endian values of its netCDF type (as the classic
 * format stores them), malloc'ed */
int h5_read_raw(ncc_file *nc, int var, long long first, long long count, unsigned char **buf);
#endif
