/* nc_classic.h -- classic netCDF (CDF-1 / CDF-2) reader and CDF-2 writer of the host layer; see nc_classic.c */
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
int ncc_num_vars(const ncc_file *nc);
const char *ncc_var_name(const ncc_file *nc, int var);
int ncc_var_ndims(const ncc_file *nc, int var);
long long ncc_var_dim(const ncc_file *nc, int var, int d, const char **name);
int ncc_var_is_packed(const ncc_file *nc, int var);                /* stored as short / byte */
int ncc_var_is_record(const ncc_file *nc, int var);                /* leading dimension is the record dimension */
/* first value of a numeric attribute of variable `var` (-1: global); 0 if absent */
int ncc_get_att(const ncc_file *nc, int var, const char *name, double *value);
/* `count` elements from element `first` of record `rec` of a record variable (leading dimension unlimited)
 * or of the whole variable (rec ignored), converted from the stored type; 1 = ok */
int ncc_read_double(ncc_file *nc, int var, long long rec, long long first, long long count, double *out);
int ncc_read_float(ncc_file *nc, int var, long long rec, long long first, long long count, float *out);
int ncc_read_short(ncc_file *nc, int var, long long rec, long long first, long long count, short *out);

/* ---- writer (CDF-2) ---- */
enum { NCC_BYTE = 1, NCC_CHAR = 2, NCC_SHORT = 3, NCC_INT = 4, NCC_FLOAT = 5, NCC_DOUBLE = 6 };
typedef struct nccw_file nccw_file;

nccw_file *nccw_create(const char *path);                           /* NULL: cannot create the file */
const char *nccw_error(const nccw_file *w);
int nccw_def_dim(nccw_file *w, const char *name, long long len);    /* len 0: the record dimension; returns the id, -1 on error */
int nccw_def_var(nccw_file *w, const char *name, int type, int ndims, const int *dimids);   /* id, -1 on error */
int nccw_put_att_text(nccw_file *w, int var, const char *name, const char *text);           /* var -1: global */
int nccw_enddef(nccw_file *w);
/* the whole variable, or record `rec` of a record variable (rec == records so far appends one); 0 = ok */
int nccw_put_double(nccw_file *w, int var, long long rec, const double *data);
int nccw_put_int(nccw_file *w, int var, long long rec, const int *data);
int nccw_find_var(const nccw_file *w, const char *name);
long long nccw_numrecs(const nccw_file *w);
int nccw_close(nccw_file *w);
#endif
