/* nc_classic.h -- classic netCDF (CDF-1 / CDF-2) reader of the host layer; see nc_classic.c */
#ifndef MPTRAC_AMD_NC_CLASSIC_H
#define MPTRAC_AMD_NC_CLASSIC_H
#define _FILE_OFFSET_BITS 64
#include <stddef.h>
#include <sys/types.h>

typedef struct ncc_file ncc_file;

ncc_file *ncc_open(const char *path, char *err, size_t errlen);   /* NULL + message on failure */
void ncc_close(ncc_file *nc);
const char *ncc_error(const ncc_file *nc);
/* index of a dimension / variable, -1 if absent; *len = current length (record dimension: number of records) */
int ncc_find_dim(const ncc_file *nc, const char *name, long long *len);
int ncc_find_var(const ncc_file *nc, const char *name);
int ncc_var_ndims(const ncc_file *nc, int var);
long long ncc_var_dim(const ncc_file *nc, int var, int d, const char **name);
int ncc_var_is_packed(const ncc_file *nc, int var);                /* stored as short / byte */
/* first value of a numeric attribute of variable `var` (-1: global); 0 if absent */
int ncc_get_att(const ncc_file *nc, int var, const char *name, double *value);
/* `count` elements from element `first` of record `rec` of a record variable (leading dimension unlimited)
 * or of the whole variable (rec ignored), converted from the stored type; 1 = ok */
int ncc_read_double(ncc_file *nc, int var, long long rec, long long first, long long count, double *out);
int ncc_read_float(ncc_file *nc, int var, long long rec, long long first, long long count, float *out);
int ncc_read_short(ncc_file *nc, int var, long long rec, long long first, long long count, short *out);
#endif
