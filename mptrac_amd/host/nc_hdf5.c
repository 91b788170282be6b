/*
 * nc_hdf5.c -- reader for netCDF-4 files, i.e. the subset of HDF5 the netCDF library writes.
 *
 * The image has no netCDF or HDF5 library.  Written from the published "HDF5 File Format Specification"
 * (version 3): superblock versions 0-3; object headers of version 1 and 2 with continuation blocks; groups as
 * symbol tables (B-tree v1 + local heap) or as link messages, stored compactly or in a fractal heap; dataspace,
 * datatype (integers and floating point numbers of 1-8 bytes in either byte order, fixed-length strings for
 * attributes), fill value, filter pipeline (deflate through zlib, shuffle, fletcher32), data layout -- compact,
 * contiguous, chunked with a version-1 B-tree (layout version 3) or as a single chunk / implicit / fixed-array
 * index (layout version 4) -- and attributes in the object header or in a fractal heap.  Only the root group is
 * read (netCDF classic data model).  What is not read is refused with a message, never guessed: variable-length
 * and compound types (except the DIMENSION_LIST attributes of netCDF-4, which name the dimension of every axis
 * through references in a global heap; files without them get axis names from the dimension scale of the
 * axis' length), extensible-array and
 * version-2 B-tree chunk indices, external storage, other filters.
 *
 * A variable's values are produced as the classic format stores them (big-endian, row-major), so everything
 * above this file is format-agnostic.  A chunked variable is assembled once, whole, and kept until another
 * variable is read.
 */
#include "nc_internal.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define UNDEF (~(uint64_t) 0)

struct h5_dataset {
  int layout;                 /* 0 compact, 1 contiguous, 2 chunked */
  uint64_t addr, size;        /* contiguous: where; compact: size */
  unsigned char *compact;
  int rank;
  uint64_t dims[8];
  int elsize, little_endian;
  /* chunked */
  uint64_t chunk[9];          /* chunk extent per axis (+ element size) */
  int chunk_nd;               /* dimensionality the layout message states (rank + 1) */
  int chunk_index;            /* 0 B-tree v1, 1 single chunk, 2 implicit, 3 fixed array */
  uint64_t index_addr, single_size;
  uint32_t single_mask;
  int nfilter, filter[8];
  unsigned char fill[8];
  int have_fill;
};

struct h5_file {
  int so, sl;                 /* size of offsets / lengths */
  uint64_t base, eof;
  int cached_var;
  unsigned char *cache;       /* whole variable `cached_var`, big-endian */
};

static int h5fail(ncc_file *nc, const char *msg) {
  snprintf(nc->err, sizeof(nc->err), "netCDF-4 / HDF5: %s", msg);
  return 0;
}

/* ---- raw access ------------------------------------------------------------------------------------------- */

static int rd(ncc_file *nc, uint64_t at, void *dst, size_t n) {
  if (at == UNDEF || fseeko(nc->f, (off_t) (nc->h5->base + at), SEEK_SET) != 0 || fread(dst, 1, n, nc->f) != n)
    return h5fail(nc, "read beyond the end of the file");
  return 1;
}

static uint64_t le(const unsigned char *p, int n) {
  uint64_t v = 0;
  for (int i = n - 1; i >= 0; i--)
    v = (v << 8) | p[i];
  return v;
}

/* an address field: all ones = "undefined" */
static uint64_t la(const unsigned char *p, int n) {
  const uint64_t v = le(p, n);
  return (n < 8 && v == (((uint64_t) 1) << (8 * n)) - 1) ? UNDEF : v;
}

/* a growable buffer holding one object-header message */
typedef struct {
  int type;
  size_t size;
  unsigned char *data;
} h5_msg;

typedef struct {
  int n, cap;
  h5_msg *m;
} h5_msgs;

static int add_msg(ncc_file *nc, h5_msgs *ms, int type, const unsigned char *data, size_t size) {
  if (ms->n == ms->cap) {
    const int want = ms->cap ? 2 * ms->cap : 32;
    h5_msg *grown = realloc(ms->m, (size_t) want * sizeof(h5_msg));
    if (!grown)
      return h5fail(nc, "out of memory");
    ms->m = grown;
    ms->cap = want;
  }
  h5_msg *m = &ms->m[ms->n++];
  m->type = type;
  m->size = size;
  m->data = malloc(size + 8);
  if (!m->data)
    return h5fail(nc, "out of memory");
  memcpy(m->data, data, size);
  memset(m->data + size, 0, 8);
  return 1;
}

static void free_msgs(h5_msgs *ms) {
  for (int i = 0; i < ms->n; i++)
    free(ms->m[i].data);
  free(ms->m);
  memset(ms, 0, sizeof(*ms));
}

/* ---- object headers ---------------------------------------------------------------------------------------- */

/* messages of one header block; continuation messages are followed */
static int parse_block_v2(ncc_file *nc, h5_msgs *ms, const unsigned char *p, size_t n, int track_order, int depth);

static int parse_block_v1(ncc_file *nc, h5_msgs *ms, const unsigned char *p, size_t n, int depth) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  size_t at = 0;
  while (at + 8 <= n) {
    const int type = (int) le(p + at, 2);
    const size_t size = (size_t) le(p + at + 2, 2);
    at += 8;
    if (at + size > n)
      return h5fail(nc, "object header message runs past its block");
    if (type == 0x10) {   /* continuation */
      if (size < (size_t) (so + sl))
        return h5fail(nc, "malformed continuation message");
      const uint64_t addr = la(p + at, so), len = le(p + at + so, sl);
      if (depth > 64 || len > (1u << 26))
        return h5fail(nc, "implausible object header continuation");
      unsigned char *blk = malloc((size_t) len);
      if (!blk)
        return h5fail(nc, "out of memory");
      const int ok = rd(nc, addr, blk, (size_t) len) && parse_block_v1(nc, ms, blk, (size_t) len, depth + 1);
      free(blk);
      if (!ok)
        return 0;
    } else if (type != 0 && !add_msg(nc, ms, type, p + at, size))
      return 0;
    at += size;
  }
  return 1;
}

static int parse_block_v2(ncc_file *nc, h5_msgs *ms, const unsigned char *p, size_t n, int track_order, int depth) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  const size_t hdr = 4 + (track_order ? 2 : 0);
  size_t at = 0;
  while (at + hdr <= n) {
    const int type = p[at];
    const size_t size = (size_t) le(p + at + 1, 2);
    at += hdr;
    if (at + size > n)
      return h5fail(nc, "object header message runs past its chunk");
    if (type == 0x10) {
      if (size < (size_t) (so + sl))
        return h5fail(nc, "malformed continuation message");
      const uint64_t addr = la(p + at, so), len = le(p + at + so, sl);
      if (depth > 64 || len < 8 || len > (1u << 26))
        return h5fail(nc, "implausible object header continuation");
      unsigned char *blk = malloc((size_t) len);
      if (!blk)
        return h5fail(nc, "out of memory");
      int ok = rd(nc, addr, blk, (size_t) len);
      if (ok && memcmp(blk, "OCHK", 4) != 0)
        ok = h5fail(nc, "object header continuation without signature");
      ok = ok && parse_block_v2(nc, ms, blk + 4, (size_t) len - 8, track_order, depth + 1);   /* (- signature, - checksum) */
      free(blk);
      if (!ok)
        return 0;
    } else if (type != 0 && !add_msg(nc, ms, type, p + at, size))
      return 0;
    at += size;
  }
  return 1;
}

static int read_object_header(ncc_file *nc, uint64_t addr, h5_msgs *ms) {
  unsigned char head[64];
  memset(ms, 0, sizeof(*ms));
  if (!rd(nc, addr, head, 16))
    return 0;
  if (memcmp(head, "OHDR", 4) == 0) {
    if (head[4] != 2)
      return h5fail(nc, "unknown object header version");
    const int flags = head[5];
    size_t at = 6;
    if (flags & 0x20)
      at += 16;
    if (flags & 0x10)
      at += 4;
    const int nsz = 1 << (flags & 3);
    if (!rd(nc, addr, head, at + (size_t) nsz))
      return 0;
    const uint64_t size0 = le(head + at, nsz);
    at += (size_t) nsz;
    if (size0 > (1u << 26))
      return h5fail(nc, "implausible object header size");
    unsigned char *blk = malloc((size_t) size0 + 8);
    if (!blk)
      return h5fail(nc, "out of memory");
    const int ok = rd(nc, addr + at, blk, (size_t) size0) && parse_block_v2(nc, ms, blk, (size_t) size0, (flags & 0x04) != 0, 0);
    free(blk);
    return ok;
  }
  if (head[0] != 1)
    return h5fail(nc, "unknown object header version");
  const uint64_t size0 = le(head + 8, 4);
  if (size0 > (1u << 26))
    return h5fail(nc, "implausible object header size");
  unsigned char *blk = malloc((size_t) size0 + 8);
  if (!blk)
    return h5fail(nc, "out of memory");
  const int ok = rd(nc, addr + 16, blk, (size_t) size0) && parse_block_v1(nc, ms, blk, (size_t) size0, 0);
  free(blk);
  return ok;
}

/* ---- fractal heaps (dense link / attribute storage) ------------------------------------------------------- */

typedef int (*heap_visit)(ncc_file *nc, void *user, const unsigned char *p, size_t n);

typedef struct {
  int so, sl;
  uint64_t start_size, max_direct;
  int width, max_bits, has_checksum, filtered;
  uint64_t self;
  heap_visit visit;
  void *user;
  long blocks;                /* indirect blocks visited so far (a crafted file may point every child at one block) */
} frheap;

static int heap_direct(ncc_file *nc, frheap *h, uint64_t addr, uint64_t size) {
  if (addr == UNDEF)
    return 1;
  if (size > (1u << 28))
    return h5fail(nc, "implausible fractal heap block");
  unsigned char *blk = malloc((size_t) size + 8);
  if (!blk)
    return h5fail(nc, "out of memory");
  int ok = rd(nc, addr, blk, (size_t) size);
  if (ok && memcmp(blk, "FHDB", 4) != 0)
    ok = h5fail(nc, "fractal heap direct block without signature");
  if (ok) {
    const size_t head = 5 + (size_t) h->so + (size_t) ((h->max_bits + 7) / 8) + (h->has_checksum ? 4 : 0);
    ok = head < size ? h->visit(nc, h->user, blk + head, (size_t) size - head) : h5fail(nc, "implausible fractal heap block");
  }
  free(blk);
  return ok;
}

static int heap_indirect(ncc_file *nc, frheap *h, uint64_t addr, int nrows, int depth) {
  if (addr == UNDEF)
    return 1;
  if (depth > 8)
    return h5fail(nc, "fractal heap nested too deeply");
  if (++h->blocks > 100000)
    return h5fail(nc, "implausible number of fractal heap blocks");
  /* rows 0, 1: start_size; row r: start_size * 2^(r-1); rows whose blocks exceed max_direct hold indirect blocks */
  int max_direct_rows = 2;
  for (uint64_t s = h->start_size; s < h->max_direct; s *= 2)
    max_direct_rows++;
  const int ndirect_rows = nrows < max_direct_rows ? nrows : max_direct_rows;
  const size_t head = 5 + (size_t) h->so + (size_t) ((h->max_bits + 7) / 8);
  const size_t entries = (size_t) nrows * (size_t) h->width;
  const size_t bytes = head + entries * (size_t) (h->so + (h->filtered ? h->sl + 4 : 0)) + 4;
  unsigned char *blk = malloc(bytes);
  if (!blk)
    return h5fail(nc, "out of memory");
  int ok = rd(nc, addr, blk, bytes - 4);
  if (ok && memcmp(blk, "FHIB", 4) != 0)
    ok = h5fail(nc, "fractal heap indirect block without signature");
  size_t at = head;
  for (int r = 0; ok && r < nrows; r++) {
    const uint64_t bsize = r < 2 ? h->start_size : h->start_size << (r - 1);
    for (int c = 0; ok && c < h->width; c++) {
      const uint64_t child = la(blk + at, h->so);
      at += (size_t) h->so;
      if (r < ndirect_rows) {
        if (h->filtered)
          at += (size_t) h->sl + 4;
        ok = heap_direct(nc, h, child, bsize);
      } else {
        int sub = 0;   /* rows of the child: its blocks cover bsize bytes of heap space */
        for (uint64_t cover = 0; cover < bsize; sub++)
          cover += (uint64_t) h->width * (sub < 2 ? h->start_size : h->start_size << (sub - 1));
        ok = heap_indirect(nc, h, child, sub, depth + 1);
      }
    }
  }
  free(blk);
  return ok;
}

/* every managed object area of the heap at `addr` is handed to `visit` (one call per direct block) */
static int walk_heap(ncc_file *nc, uint64_t addr, heap_visit visit, void *user) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  unsigned char b[256];
  const size_t need = 5 + 2 + 2 + 1 + 4 + (size_t) sl + (size_t) so + (size_t) sl + (size_t) so + 8 * (size_t) sl + 2 + 2 * (size_t) sl + 2 + 2
    + (size_t) so + 2;
  if (need > sizeof(b) || !rd(nc, addr, b, need))
    return 0;
  if (memcmp(b, "FRHP", 4) != 0)
    return h5fail(nc, "fractal heap without signature");
  size_t at = 5;
  at += 2;   /* heap ID length */
  const int filter_len = (int) le(b + at, 2);
  at += 2;
  const int flags = b[at];
  at += 1;
  at += 4;                           /* max size of managed objects */
  at += (size_t) sl + (size_t) so;   /* next huge ID, huge B-tree */
  at += (size_t) sl + (size_t) so;   /* free space, free-space manager */
  at += 8 * (size_t) sl;             /* managed space ... number of tiny objects */
  frheap h;
  memset(&h, 0, sizeof(h));
  h.so = so;
  h.sl = sl;
  h.width = (int) le(b + at, 2);
  at += 2;
  h.start_size = le(b + at, sl);
  at += (size_t) sl;
  h.max_direct = le(b + at, sl);
  at += (size_t) sl;
  h.max_bits = (int) le(b + at, 2);
  at += 2;
  at += 2;   /* starting rows of the root indirect block */
  const uint64_t root = la(b + at, so);
  at += (size_t) so;
  const int cur_rows = (int) le(b + at, 2);
  h.has_checksum = (flags & 2) != 0;
  h.filtered = filter_len > 0;
  h.visit = visit;
  h.user = user;
  if (h.filtered)
    return h5fail(nc, "filtered fractal heaps are not read");
  if (h.width < 1 || h.width > 1024 || h.start_size < 16 || h.start_size > (1u << 24) || h.max_bits < 8 || h.max_bits > 64
      || h.max_direct < h.start_size || h.max_direct > ((uint64_t) 1 << 32) || cur_rows > 64)
    return h5fail(nc, "implausible fractal heap");
  if (cur_rows == 0)
    return heap_direct(nc, &h, root, h.start_size);
  return heap_indirect(nc, &h, root, cur_rows, 0);
}

/* ---- links ------------------------------------------------------------------------------------------------ */

typedef struct {
  int n, cap;
  char **name;
  uint64_t *addr;
} h5_links;

static int add_link(ncc_file *nc, h5_links *L, const char *name, size_t len, uint64_t addr) {
  if (L->n == L->cap) {
    const int want = L->cap ? 2 * L->cap : 64;
    char **names = realloc(L->name, (size_t) want * sizeof(char *));
    if (names)
      L->name = names;
    uint64_t *addrs = realloc(L->addr, (size_t) want * sizeof(uint64_t));
    if (addrs)
      L->addr = addrs;
    if (!names || !addrs)
      return h5fail(nc, "out of memory");
    L->cap = want;
  }
  L->name[L->n] = malloc(len + 1);
  if (!L->name[L->n])
    return h5fail(nc, "out of memory");
  memcpy(L->name[L->n], name, len);
  L->name[L->n][len] = '\0';
  L->addr[L->n++] = addr;
  return 1;
}

/* one link message; *used = its length (0: not a link message) */
static int parse_link(ncc_file *nc, h5_links *L, const unsigned char *p, size_t n, size_t *used) {
  *used = 0;
  if (n < 4 || p[0] != 1)
    return 1;
  const int flags = p[1];
  size_t at = 2;
  int type = 0;
  if (flags & 0x08)
    type = p[at++];
  if (flags & 0x04)
    at += 8;
  if (flags & 0x10)
    at += 1;
  const int lsz = 1 << (flags & 3);
  if (at + (size_t) lsz > n)
    return 1;
  const uint64_t len = le(p + at, lsz);
  at += (size_t) lsz;
  if (len == 0 || len > 4096 || at + len > n)
    return 1;
  const char *name = (const char *) p + at;
  at += (size_t) len;
  if (type == 0) {
    if (at + (size_t) nc->h5->so > n)
      return 1;
    const uint64_t addr = la(p + at, nc->h5->so);
    at += (size_t) nc->h5->so;
    if (addr != UNDEF && addr < nc->h5->eof && !add_link(nc, L, name, (size_t) len, addr))
      return 0;
  } else if (type == 1) {   /* soft link: length + target path; not followed */
    if (at + 2 > n)
      return 1;
    at += 2 + (size_t) le(p + at, 2);
  } else
    return 1;
  *used = at;
  return 1;
}

static int visit_links(ncc_file *nc, void *user, const unsigned char *p, size_t n) {
  size_t at = 0;
  while (at < n) {
    size_t used;
    if (!parse_link(nc, (h5_links *) user, p + at, n - at, &used))
      return 0;
    if (!used)
      break;   /* the unused tail of the block */
    at += used;
  }
  return 1;
}

/* old-style group: B-tree v1 of symbol table nodes, names in a local heap */
static int symbol_tree(ncc_file *nc, h5_links *L, uint64_t node, const unsigned char *heap, size_t heap_size, int depth) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  unsigned char head[8];
  if (depth > 32 || !rd(nc, node, head, 8))
    return depth > 32 ? h5fail(nc, "group B-tree nested too deeply") : 0;
  if (memcmp(head, "SNOD", 4) == 0) {
    const int nsym = (int) le(head + 6, 2);
    if (nsym > 8192)
      return h5fail(nc, "implausible symbol table node");
    const size_t esz = 2 * (size_t) so + 24;
    unsigned char *e = malloc((size_t) nsym * esz + 8);
    if (!e)
      return h5fail(nc, "out of memory");
    int ok = rd(nc, node + 8, e, (size_t) nsym * esz);
    for (int i = 0; ok && i < nsym; i++) {
      const uint64_t off = la(e + (size_t) i * esz, so), addr = la(e + (size_t) i * esz + (size_t) so, so);
      if (off < heap_size) {
        const char *name = (const char *) heap + off;
        ok = add_link(nc, L, name, strnlen(name, heap_size - (size_t) off), addr);
      }
    }
    free(e);
    return ok;
  }
  if (memcmp(head, "TREE", 4) != 0 || head[4] != 0)
    return h5fail(nc, "group B-tree node without signature");
  const int used = (int) le(head + 6, 2);
  if (used > 8192)
    return h5fail(nc, "implausible group B-tree node");
  const size_t bytes = 2 * (size_t) so + (size_t) (2 * used + 1) * (size_t) (so > sl ? so : sl) + 16;
  unsigned char *b = malloc(bytes);
  if (!b)
    return h5fail(nc, "out of memory");
  int ok = rd(nc, node + 8 + 2 * (uint64_t) so, b, (size_t) used * (size_t) (sl + so) + (size_t) sl);
  for (int i = 0; ok && i < used; i++)
    ok = symbol_tree(nc, L, la(b + (size_t) sl + (size_t) i * (size_t) (sl + so), so), heap, heap_size, depth + 1);
  free(b);
  return ok;
}

static int group_links(ncc_file *nc, const h5_msgs *ms, h5_links *L) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  for (int i = 0; i < ms->n; i++) {
    const h5_msg *m = &ms->m[i];
    if (m->type == 0x06) {
      size_t used;
      if (!parse_link(nc, L, m->data, m->size, &used))
        return 0;
    } else if (m->type == 0x02) {   /* link info: dense storage */
      size_t at = 2;
      if (m->data[1] & 1)
        at += 8;
      const uint64_t heap = la(m->data + at, so);
      if (heap != UNDEF && !walk_heap(nc, heap, visit_links, L))
        return 0;
    } else if (m->type == 0x11) {   /* symbol table */
      const uint64_t tree = la(m->data, so), lheap = la(m->data + so, so);
      unsigned char h[64];
      if (!rd(nc, lheap, h, 8 + 2 * (size_t) sl + (size_t) so))
        return 0;
      if (memcmp(h, "HEAP", 4) != 0)
        return h5fail(nc, "local heap without signature");
      const uint64_t dsize = le(h + 8, sl), daddr = la(h + 8 + 2 * (size_t) sl, so);
      if (dsize > (1u << 26))
        return h5fail(nc, "implausible local heap");
      unsigned char *data = malloc((size_t) dsize + 8);
      if (!data)
        return h5fail(nc, "out of memory");
      memset(data, 0, (size_t) dsize + 8);
      const int ok = rd(nc, daddr, data, (size_t) dsize) && symbol_tree(nc, L, tree, data, (size_t) dsize, 0);
      free(data);
      if (!ok)
        return 0;
    }
  }
  return 1;
}

/* ---- datatypes, dataspaces, attributes -------------------------------------------------------------------- */

typedef struct {
  int cls, size, little_endian, is_signed;
} h5_type;

static int parse_type(const unsigned char *p, size_t n, h5_type *t) {
  if (n < 8)
    return 0;
  t->cls = p[0] & 0x0f;
  t->little_endian = !(p[1] & 1);
  t->is_signed = (p[1] & 0x08) != 0;
  t->size = (int) le(p + 4, 4);
  return 1;
}

/* netCDF type of an HDF5 number type; 0: none */
static int nc_type_of(const h5_type *t) {
  if (t->cls == 1)
    return t->size == 4 ? T_FLOAT : (t->size == 8 ? T_DOUBLE : 0);
  if (t->cls == 0)
    switch (t->size) {
    case 1: return T_BYTE;
    case 2: return T_SHORT;
    case 4: return T_INT;
    case 8: return T_INT64;
    }
  return 0;
}

static int parse_space(ncc_file *nc, const unsigned char *p, size_t n, int *rank, uint64_t *dims) {
  const int sl = nc->h5->sl;
  if (n < 4)
    return h5fail(nc, "short dataspace message");
  const int version = p[0];
  *rank = p[1];
  if (*rank > 8)
    return h5fail(nc, "more than eight dimensions");
  const size_t at = version == 1 ? 8 : 4;
  if (version == 2 && p[3] == 2) {   /* null dataspace */
    *rank = 0;
    return 1;
  }
  if (at + (size_t) *rank * (size_t) sl > n)
    return h5fail(nc, "short dataspace message");
  uint64_t total = 1;
  for (int d = 0; d < *rank; d++) {
    dims[d] = le(p + at + (size_t) d * (size_t) sl, sl);
    if (dims[d] > ((uint64_t) 1 << 34) || (total *= (dims[d] ? dims[d] : 1)) > ((uint64_t) 1 << 34))
      return h5fail(nc, "implausible dataspace extent");
  }
  return 1;
}

static void to_big_endian(unsigned char *p, size_t count, int elsize, int little_endian) {
  if (!little_endian || elsize == 1)
    return;
  for (size_t i = 0; i < count; i++) {
    unsigned char *e = p + i * (size_t) elsize;
    for (int a = 0, b = elsize - 1; a < b; a++, b--) {
      const unsigned char t = e[a];
      e[a] = e[b];
      e[b] = t;
    }
  }
}

typedef struct {
  int n, cap;
  ncc_att *a;
} att_list;

/* one attribute message (body as in the object header); strings and numbers are kept, others skipped */
static int parse_attribute(ncc_file *nc, att_list *A, const unsigned char *p, size_t n, size_t *used) {
  *used = 0;
  if (n < 8)
    return 1;
  const int version = p[0];
  if (version < 1 || version > 3)
    return 1;
  const size_t name_size = (size_t) le(p + 2, 2), type_size_ = (size_t) le(p + 4, 2), space_size = (size_t) le(p + 6, 2);
  size_t at = version == 3 ? 9 : 8;
  const size_t pad = version == 1 ? 7 : 0;
#define PADDED(x) (((x) + pad) & ~pad)
  if (name_size == 0 || at + PADDED(name_size) + PADDED(type_size_) + PADDED(space_size) > n)
    return 1;
  const char *name = (const char *) p + at;
  at += PADDED(name_size);
  h5_type t;
  if (!parse_type(p + at, type_size_, &t))
    return 1;
  at += PADDED(type_size_);
  int rank;
  uint64_t dims[8];
  if (!parse_space(nc, p + at, space_size, &rank, dims))
    return 0;
  at += PADDED(space_size);
#undef PADDED
  uint64_t count = 1;
  for (int d = 0; d < rank; d++)
    count *= dims[d];
  const uint64_t bytes = count * (uint64_t) t.size;
  if (bytes > n - at)
    return 1;
  *used = at + (size_t) bytes;
  int type = nc_type_of(&t);
  if (t.cls == 3)
    type = T_CHAR;
  if (t.cls == 9 && name_size >= 14 && strncmp(name, "DIMENSION_LIST", 14) == 0)
    type = T_BYTE;   /* kept as stored: one (length, global heap address, index) entry per axis -- axis_scale() */
  if (!type)
    return 1;   /* variable-length, reference, compound ...: not an attribute the classic model has */
  if (A->n == A->cap) {
    const int want = A->cap ? 2 * A->cap : 16;
    ncc_att *grown = realloc(A->a, (size_t) want * sizeof(ncc_att));
    if (!grown)
      return h5fail(nc, "out of memory");
    A->a = grown;
    A->cap = want;
  }
  ncc_att *a = &A->a[A->n];
  memset(a, 0, sizeof(*a));
  a->name = malloc(name_size + 1);
  a->raw = malloc((size_t) bytes + 8);
  if (!a->name || !a->raw)
    return h5fail(nc, "out of memory");
  memcpy(a->name, name, name_size);
  a->name[name_size] = '\0';
  memcpy(a->raw, p + at, (size_t) bytes);
  memset(a->raw + bytes, 0, 8);
  a->type = type;
  a->n = (type == T_CHAR || t.cls == 9) ? (size_t) bytes : (size_t) count;
  if (type != T_CHAR && t.cls != 9)
    to_big_endian(a->raw, (size_t) count, t.size, t.little_endian);
  A->n++;
  return 1;
}

static int visit_attributes(ncc_file *nc, void *user, const unsigned char *p, size_t n) {
  size_t at = 0;
  while (at < n) {
    size_t used;
    if (!parse_attribute(nc, (att_list *) user, p + at, n - at, &used))
      return 0;
    if (!used)
      break;
    at += used;
  }
  return 1;
}

static int object_attributes(ncc_file *nc, const h5_msgs *ms, att_list *A) {
  for (int i = 0; i < ms->n; i++) {
    const h5_msg *m = &ms->m[i];
    size_t used;
    if (m->type == 0x0c && !parse_attribute(nc, A, m->data, m->size, &used))
      return 0;
    if (m->type == 0x15) {   /* attribute info: dense storage */
      size_t at = 2;
      if (m->data[1] & 1)
        at += 2;
      const uint64_t heap = la(m->data + at, nc->h5->so);
      if (heap != UNDEF && !walk_heap(nc, heap, visit_attributes, A))
        return 0;
    }
  }
  return 1;
}

static const ncc_att *find_att(const att_list *A, const char *name) {
  for (int i = 0; i < A->n; i++)
    if (strcmp(A->a[i].name, name) == 0)
      return &A->a[i];
  return NULL;
}

/* Object header address of the dimension scale attached to axis k of a variable: entry k of its DIMENSION_LIST
 * attribute names an object of a global heap collection that holds the references.  UNDEF: none / not readable. */
static uint64_t axis_scale(ncc_file *nc, const att_list *A, int k) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  const ncc_att *a = find_att(A, "DIMENSION_LIST");
  const size_t entry = 8 + (size_t) so;
  if (!a || a->type != T_BYTE || (size_t) (k + 1) * entry > a->n)
    return UNDEF;
  const unsigned char *e = a->raw + (size_t) k * entry;
  const uint64_t count = le(e, 4), heap = la(e + 4, so), index = le(e + 4 + so, 4);
  if (count < 1 || heap == UNDEF || index == 0)
    return UNDEF;
  unsigned char head[32];
  if (!rd(nc, heap, head, 8 + (size_t) sl) || memcmp(head, "GCOL", 4) != 0) {
    nc->err[0] = '\0';
    return UNDEF;
  }
  const uint64_t size = le(head + 8, sl);
  if (size < 16 || size > (1u << 26))
    return UNDEF;
  unsigned char *col = malloc((size_t) size + 16);
  uint64_t found = UNDEF;
  if (col && rd(nc, heap, col, (size_t) size)) {
    size_t at = 8 + (size_t) sl;
    while (at + 8 + (size_t) sl <= size) {
      const uint64_t id = le(col + at, 2), osize = le(col + at + 8, sl);
      const size_t data = at + 8 + (size_t) sl;
      if (id == 0 || osize > size - data)
        break;   /* object 0 is the free space at the end */
      if (id == index) {
        if (osize >= (uint64_t) so)
          found = la(col + data, so);
        break;
      }
      at = data + (((size_t) osize + 7) & ~(size_t) 7);
    }
  } else
    nc->err[0] = '\0';
  free(col);
  return found;
}

/* ---- datasets --------------------------------------------------------------------------------------------- */

static int parse_dataset(ncc_file *nc, const h5_msgs *ms, struct h5_dataset *d, h5_type *t, int *is_dataset) {
  const int so = nc->h5->so, sl = nc->h5->sl;
  int have_space = 0, have_type = 0, have_layout = 0;
  memset(d, 0, sizeof(*d));
  for (int i = 0; i < ms->n; i++) {
    const h5_msg *m = &ms->m[i];
    const unsigned char *p = m->data;
    if (m->type == 0x01) {
      if (!parse_space(nc, p, m->size, &d->rank, d->dims))
        return 0;
      have_space = 1;
    } else if (m->type == 0x03) {
      have_type = parse_type(p, m->size, t);
    } else if (m->type == 0x05 && m->size >= 4) {   /* fill value */
      const int version = p[0];
      if (version <= 2) {
        if (m->size >= 8 && (version == 1 || p[3])) {
          const uint64_t sz = le(p + 4, 4);
          if (sz > 0 && sz <= 8 && 8 + sz <= m->size) {
            memcpy(d->fill, p + 8, (size_t) sz);
            d->have_fill = (int) sz;
          }
        }
      } else if (version == 3 && (p[1] & 0x20) && m->size >= 6) {
        const uint64_t sz = le(p + 2, 4);
        if (sz > 0 && sz <= 8 && 6 + sz <= m->size) {
          memcpy(d->fill, p + 6, (size_t) sz);
          d->have_fill = (int) sz;
        }
      }
    } else if (m->type == 0x0b) {   /* filter pipeline */
      const int version = p[0], nf = p[1];
      size_t at = version == 1 ? 8 : 2;
      if (nf > 8)
        return h5fail(nc, "too many filters");
      for (int k = 0; k < nf; k++) {
        const int id = (int) le(p + at, 2);
        size_t name_len = 0;
        at += 2;
        if (version == 1 || id >= 256) {
          name_len = (size_t) le(p + at, 2);
          at += 2;
        }
        at += 2;   /* flags */
        const size_t nclient = (size_t) le(p + at, 2);
        at += 2;
        at += version == 1 ? ((name_len + 7) & ~(size_t) 7) : name_len;
        at += 4 * nclient;
        if (version == 1 && (nclient & 1))
          at += 4;
        if (at > m->size)
          return h5fail(nc, "malformed filter pipeline");
        d->filter[d->nfilter++] = id;
      }
    } else if (m->type == 0x08) {   /* data layout */
      /* every fixed-offset read below is checked against the message's length first */
#define NEED(n)                                                      \
  do {                                                               \
    if ((size_t) (n) > m->size)                                      \
      return h5fail(nc, "malformed data layout message");            \
  } while (0)
      NEED(2);
      const int version = p[0];
      have_layout = 1;
      if (version == 3 || version == 4) {
        d->layout = p[1];
        if (d->layout == 0) {
          NEED(4);
          d->size = le(p + 2, 2);
          if (4 + d->size > m->size)
            return h5fail(nc, "compact data run past their message");
          d->compact = malloc((size_t) d->size + 8);
          if (!d->compact)
            return h5fail(nc, "out of memory");
          memcpy(d->compact, p + 4, (size_t) d->size);
        } else if (d->layout == 1) {
          NEED(2 + so + sl);
          d->addr = la(p + 2, so);
          d->size = le(p + 2 + so, sl);
        } else if (d->layout == 2 && version == 3) {
          NEED(3);
          const int nd = p[2];
          if (nd < 2 || nd > 9)
            return h5fail(nc, "chunk dimensionality out of range");
          NEED(3 + so + 4 * nd);
          d->chunk_nd = nd;
          d->index_addr = la(p + 3, so);
          for (int k = 0; k < nd; k++)
            d->chunk[k] = le(p + 3 + (size_t) so + 4 * (size_t) k, 4);
          d->chunk_index = 0;
        } else if (d->layout == 2) {
          NEED(5);
          const int flags = p[2], nd = p[3], enc = p[4];
          if (nd < 2 || nd > 9 || enc < 1 || enc > 8)
            return h5fail(nc, "chunk dimensionality out of range");
          NEED(5 + nd * enc + 1);
          d->chunk_nd = nd;
          size_t at = 5;
          for (int k = 0; k < nd; k++, at += (size_t) enc)
            d->chunk[k] = le(p + at, enc);
          const int index = p[at++];
          if (index == 1) {   /* single chunk */
            if (flags & 2) {
              NEED(at + (size_t) sl + 4);
              d->single_size = le(p + at, sl);
              at += (size_t) sl;
              d->single_mask = (uint32_t) le(p + at, 4);
              at += 4;
            }
            d->chunk_index = 1;
          } else if (index == 2)
            d->chunk_index = 2;
          else if (index == 3) {
            at += 1;   /* page bits */
            d->chunk_index = 3;
          } else
            return h5fail(nc, "chunk index of this kind (extensible array / B-tree v2: datasets with unlimited dimensions "
                              "written with the latest file format) is not read");
          NEED(at + (size_t) so);
          d->index_addr = la(p + at, so);
        } else
          return h5fail(nc, "virtual or unknown data layout");
      } else if (version == 1 || version == 2) {
        NEED(8);
        const int nd = p[1];
        d->layout = p[2];
        if (nd < 1 || nd > 8)
          return h5fail(nc, "data layout dimensionality out of range");
        size_t at = 8;
        if (d->layout != 0) {
          NEED(at + (size_t) so);
          d->addr = d->index_addr = la(p + at, so);
          at += (size_t) so;
        }
        NEED(at + 4 * (size_t) nd + (d->layout == 1 ? 0 : 4));
        uint64_t total = 1;
        for (int k = 0; k < nd; k++) {
          d->chunk[k] = le(p + at + 4 * (size_t) k, 4);
          total *= d->chunk[k];
        }
        at += 4 * (size_t) nd;
        if (d->layout == 2) {
          d->chunk[nd] = le(p + at, 4);
          d->chunk_nd = nd + 1;
        } else if (d->layout == 0) {
          d->size = le(p + at, 4);
          if (at + 4 + d->size > m->size)
            return h5fail(nc, "compact data run past their message");
          d->compact = malloc((size_t) d->size + 8);
          if (!d->compact)
            return h5fail(nc, "out of memory");
          memcpy(d->compact, p + at + 4, (size_t) d->size);
        } else
          d->size = total;   /* (element size applied by the caller) */
        d->chunk_index = 0;
      } else
        return h5fail(nc, "unknown data layout version");
#undef NEED
    }
  }
  if (have_layout && have_space && d->layout == 2 && d->chunk_nd != d->rank + 1)
    return h5fail(nc, "chunk dimensionality does not match the dataspace");
  *is_dataset = have_space && have_type && have_layout;
  if (*is_dataset && (t->size < 1 || t->size > 8) && nc_type_of(t))
    return h5fail(nc, "implausible element size");
  if (*is_dataset) {
    d->elsize = t->size;
    d->little_endian = t->little_endian;
  }
  return 1;
}

void h5_free_dataset(struct h5_dataset *d) {
  if (d) {
    free(d->compact);
    free(d);
  }
}

/* ---- chunk assembly --------------------------------------------------------------------------------------- */

static int unfilter(ncc_file *nc, const struct h5_dataset *d, unsigned char **buf, size_t *len, uint32_t mask, size_t chunk_bytes) {
  for (int k = d->nfilter - 1; k >= 0; k--) {
    if (mask & (1u << k))
      continue;
    const int id = d->filter[k];
    if (id == 1) {   /* deflate */
      unsigned char *out = malloc(chunk_bytes + 8);
      if (!out)
        return h5fail(nc, "out of memory");
      uLongf outlen = (uLongf) chunk_bytes + 4;   /* (+ fletcher32 trailer if that filter comes first) */
      z_stream z;
      memset(&z, 0, sizeof(z));
      if (inflateInit(&z) != Z_OK) {
        free(out);
        return h5fail(nc, "zlib initialisation failed");
      }
      z.next_in = *buf;
      z.avail_in = (uInt) *len;
      z.next_out = out;
      z.avail_out = (uInt) outlen;
      const int rc = inflate(&z, Z_FINISH);
      const size_t got = (size_t) z.total_out;
      inflateEnd(&z);
      if (rc != Z_STREAM_END) {
        free(out);
        return h5fail(nc, "a compressed chunk could not be inflated");
      }
      free(*buf);
      *buf = out;
      *len = got;
    } else if (id == 2) {   /* shuffle: byte k of every element stored together */
      const size_t es = (size_t) d->elsize, ne = *len / es;
      if (es > 1 && ne > 0) {
        unsigned char *out = malloc(*len + 8);
        if (!out)
          return h5fail(nc, "out of memory");
        for (size_t b = 0; b < es; b++)
          for (size_t e = 0; e < ne; e++)
            out[e * es + b] = (*buf)[b * ne + e];
        memcpy(out + ne * es, *buf + ne * es, *len - ne * es);
        free(*buf);
        *buf = out;
      }
    } else if (id == 3) {   /* fletcher32: checksum behind the data */
      if (*len >= 4)
        *len -= 4;
    } else
      return h5fail(nc, "a filter other than deflate, shuffle and fletcher32 is applied to the data");
  }
  return 1;
}

/* one stored chunk (at `addr`, `size` bytes, first element at `off`) copied into the whole-variable array */
static int place_chunk(ncc_file *nc, const struct h5_dataset *d, unsigned char *whole, uint64_t addr, uint64_t size,
                       uint32_t mask, const uint64_t *off) {
  const int r = d->rank;
  size_t chunk_elems = 1;
  for (int k = 0; k < r; k++)
    chunk_elems *= (size_t) d->chunk[k];
  const size_t chunk_bytes = chunk_elems * (size_t) d->elsize;
  if (size > (1u << 30) || chunk_bytes > ((size_t) 1 << 31))
    return h5fail(nc, "implausible chunk size");
  unsigned char *buf = malloc((size_t) size + 8);
  if (!buf)
    return h5fail(nc, "out of memory");
  size_t len = (size_t) size;
  int ok = rd(nc, addr, buf, len) && unfilter(nc, d, &buf, &len, mask, chunk_bytes);
  if (ok && len < chunk_bytes)
    ok = h5fail(nc, "a chunk is shorter than its extent");
  if (ok) {
    /* rows along the last axis */
    uint64_t idx[8] = { 0 };
    const size_t es = (size_t) d->elsize;
    const uint64_t last = r ? d->chunk[r - 1] : 1;
    size_t nrows = chunk_elems / (size_t) (last ? last : 1);
    for (size_t row = 0; row < nrows; row++) {
      int inside = 1;
      uint64_t flat = 0;
      for (int k = 0; k < r - 1; k++) {
        const uint64_t g = off[k] + idx[k];
        if (g >= d->dims[k])
          inside = 0;
        flat = flat * d->dims[k] + g;
      }
      if (inside && r >= 1 && off[r - 1] < d->dims[r - 1]) {
        uint64_t ncopy = d->dims[r - 1] - off[r - 1];
        if (ncopy > last)
          ncopy = last;
        flat = flat * d->dims[r - 1] + off[r - 1];
        memcpy(whole + (size_t) flat * es, buf + row * (size_t) last * es, (size_t) ncopy * es);
      }
      for (int k = r - 2; k >= 0; k--) {
        if (++idx[k] < d->chunk[k])
          break;
        idx[k] = 0;
      }
    }
  }
  free(buf);
  return ok;
}

static int chunk_tree(ncc_file *nc, const struct h5_dataset *d, unsigned char *whole, uint64_t node, int depth) {
  const int so = nc->h5->so, nd = d->rank + 1;
  unsigned char head[8];
  if (depth > 32)
    return h5fail(nc, "chunk B-tree nested too deeply");
  if (!rd(nc, node, head, 8))
    return 0;
  if (memcmp(head, "TREE", 4) != 0 || head[4] != 1)
    return h5fail(nc, "chunk B-tree node without signature");
  const int level = head[5], used = (int) le(head + 6, 2);
  if (used > 8192)
    return h5fail(nc, "implausible chunk B-tree node");
  const size_t key = 8 + 8 * (size_t) nd, entry = key + (size_t) so;
  unsigned char *b = malloc((size_t) used * entry + key + 8);
  if (!b)
    return h5fail(nc, "out of memory");
  int ok = rd(nc, node + 8 + 2 * (uint64_t) so, b, (size_t) used * entry + key);
  for (int i = 0; ok && i < used; i++) {
    const unsigned char *k = b + (size_t) i * entry;
    const uint64_t child = la(k + key, so);
    if (level > 0)
      ok = chunk_tree(nc, d, whole, child, depth + 1);
    else {
      uint64_t off[8];
      for (int a = 0; a < d->rank; a++)
        off[a] = le(k + 8 + 8 * (size_t) a, 8);
      ok = place_chunk(nc, d, whole, child, le(k, 4), (uint32_t) le(k + 4, 4), off);
    }
  }
  free(b);
  return ok;
}

/* the whole variable, native byte order */
static int assemble(ncc_file *nc, const struct h5_dataset *d, unsigned char *whole, size_t total_bytes) {
  /* unwritten chunks read as the fill value */
  if (d->have_fill == d->elsize)
    for (size_t i = 0; i + (size_t) d->elsize <= total_bytes; i += (size_t) d->elsize)
      memcpy(whole + i, d->fill, (size_t) d->elsize);
  else
    memset(whole, 0, total_bytes);
  if (d->index_addr == UNDEF)
    return 1;   /* nothing was ever written */
  size_t chunk_elems = 1;
  uint64_t nchunk[8], total_chunks = 1;
  for (int k = 0; k < d->rank; k++) {
    if (d->chunk[k] == 0 || d->chunk[k] > ((uint64_t) 1 << 31) || chunk_elems > ((size_t) 1 << 31) / (size_t) d->chunk[k])
      return h5fail(nc, "implausible chunk extent");
    chunk_elems *= (size_t) d->chunk[k];
    nchunk[k] = (d->dims[k] + d->chunk[k] - 1) / d->chunk[k];
    total_chunks *= nchunk[k];
  }
  const uint64_t chunk_bytes = (uint64_t) chunk_elems * (uint64_t) d->elsize;
  if (d->chunk_index == 0)
    return chunk_tree(nc, d, whole, d->index_addr, 0);
  if (d->chunk_index == 1) {
    const uint64_t off[8] = { 0 };
    return place_chunk(nc, d, whole, d->index_addr, d->nfilter ? d->single_size : chunk_bytes, d->single_mask, off);
  }
  /* implicit index: the chunks follow each other; fixed array: a table of their addresses */
  unsigned char *table = NULL;
  size_t entry = 0;
  if (d->chunk_index == 3) {
    const int so = nc->h5->so;
    unsigned char h[64];
    if (!rd(nc, d->index_addr, h, 12 + (size_t) nc->h5->sl + (size_t) so))
      return 0;
    if (memcmp(h, "FAHD", 4) != 0)
      return h5fail(nc, "fixed array chunk index without signature");
    entry = h[6];
    const int page_bits = h[7];
    const uint64_t nent = le(h + 8, nc->h5->sl), data = la(h + 8 + (size_t) nc->h5->sl, so);
    if (nent < total_chunks)
      return h5fail(nc, "fixed array chunk index is too short");
    if (page_bits < 1 || page_bits > 63)
      return h5fail(nc, "implausible fixed array page size");
    if (nent > ((uint64_t) 1 << page_bits))
      return h5fail(nc, "paged fixed array chunk indices are not read");
    if (data == UNDEF)
      return 1;
    table = malloc((size_t) total_chunks * entry + 8);
    if (!table)
      return h5fail(nc, "out of memory");
    if (!rd(nc, data + 6 + (uint64_t) so, table, (size_t) total_chunks * entry)) {   /* "FADB", version, client, header address */
      free(table);
      return 0;
    }
  }
  int ok = 1;
  for (uint64_t c = 0; ok && c < total_chunks; c++) {
    uint64_t off[8], rem = c;
    for (int k = d->rank - 1; k >= 0; k--) {
      off[k] = (rem % nchunk[k]) * d->chunk[k];
      rem /= nchunk[k];
    }
    if (d->chunk_index == 2)
      ok = place_chunk(nc, d, whole, d->index_addr + c * chunk_bytes, chunk_bytes, 0, off);
    else {
      const unsigned char *e = table + (size_t) c * entry;
      const int so = nc->h5->so;
      const uint64_t addr = la(e, so);
      if (addr == UNDEF)
        continue;
      if (d->nfilter) {   /* address, chunk size (entry - so - 4 bytes), filter mask */
        const int csz = (int) entry - so - 4;
        ok = place_chunk(nc, d, whole, addr, le(e + so, csz), (uint32_t) le(e + so + csz, 4), off);
      } else
        ok = place_chunk(nc, d, whole, addr, chunk_bytes, 0, off);
    }
  }
  free(table);
  return ok;
}

int h5_read_raw(ncc_file *nc, int var, long long first, long long count, unsigned char **buf) {
  const ncc_var *v = &nc->var[var];
  const struct h5_dataset *d = v->h5;
  const size_t es = (size_t) d->elsize;
  *buf = malloc((size_t) count * es + 8);
  if (!*buf)
    return h5fail(nc, "out of memory");
  int ok = 1;
  if (d->layout == 0) {
    if ((uint64_t) (first + count) * es > d->size)
      ok = h5fail(nc, "compact data shorter than the variable");
    else
      memcpy(*buf, d->compact + (size_t) first * es, (size_t) count * es);
    if (ok)
      to_big_endian(*buf, (size_t) count, d->elsize, d->little_endian);
  } else if (d->layout == 1) {
    if (d->addr == UNDEF) {   /* never written: fill value */
      for (long long i = 0; i < count; i++)
        if (d->have_fill == d->elsize)
          memcpy(*buf + (size_t) i * es, d->fill, es);
        else
          memset(*buf + (size_t) i * es, 0, es);
    } else
      ok = rd(nc, d->addr + (uint64_t) first * es, *buf, (size_t) count * es);
    if (ok)
      to_big_endian(*buf, (size_t) count, d->elsize, d->little_endian);
  } else {
    struct h5_file *h = nc->h5;
    if (h->cached_var != var || !h->cache) {
      free(h->cache);
      h->cache = NULL;
      h->cached_var = -1;
      const size_t total = (size_t) v->nelem * es;
      unsigned char *whole = malloc(total + 8);
      if (!whole)
        ok = h5fail(nc, "out of memory");
      else if (!assemble(nc, d, whole, total)) {
        free(whole);
        ok = 0;
      } else {
        to_big_endian(whole, (size_t) v->nelem, d->elsize, d->little_endian);
        h->cache = whole;
        h->cached_var = var;
      }
    }
    if (ok)
      memcpy(*buf, h->cache + (size_t) first * es, (size_t) count * es);
  }
  if (!ok) {
    free(*buf);
    *buf = NULL;
  }
  return ok;
}

/* ---- the file --------------------------------------------------------------------------------------------- */

void h5_free(ncc_file *nc) {
  if (nc->h5) {
    free(nc->h5->cache);
    free(nc->h5);
    nc->h5 = NULL;
  }
}

static int find_or_add_dim(ncc_file *nc, const char *name, uint64_t len, int *cap) {
  if (nc->ndim == *cap) {
    const int want = *cap ? 2 * *cap : 32;
    char **names = realloc(nc->dim_name, (size_t) want * sizeof(char *));
    if (names)
      nc->dim_name = names;
    long long *lens = realloc(nc->dim_len, (size_t) want * sizeof(long long));
    if (lens)
      nc->dim_len = lens;
    if (!names || !lens)
      return -1;
    *cap = want;
  }
  char *copy = strdup(name);
  if (!copy)
    return -1;
  nc->dim_name[nc->ndim] = copy;
  nc->dim_len[nc->ndim] = (long long) len;
  return nc->ndim++;
}

int h5_load(ncc_file *nc) {
  struct h5_file *h = calloc(1, sizeof(*h));
  if (!h)
    return h5fail(nc, "out of memory");
  nc->h5 = h;
  h->cached_var = -1;
  h->so = h->sl = 8;
  nc->numrecs = 0;

  /* superblock (at offset 0; a user block in front of it is not supported) */
  unsigned char sb[128];
  if (fseeko(nc->f, 0, SEEK_SET) != 0 || fread(sb, 1, sizeof(sb), nc->f) < 48)
    return h5fail(nc, "file too short");
  const int version = sb[8];
  uint64_t root = UNDEF;
  if (version == 0 || version == 1) {
    h->so = sb[13];
    h->sl = sb[14];
    if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8))
      return h5fail(nc, "unsupported size of offsets / lengths");
    size_t at = 24 + (version == 1 ? 4 : 0);
    h->base = la(sb + at, h->so);
    if (h->base == UNDEF)
      h->base = 0;
    h->eof = la(sb + at + 2 * (size_t) h->so, h->so);
    at += 4 * (size_t) h->so;
    root = la(sb + at + (size_t) h->so, h->so);   /* object header address of the root symbol table entry */
  } else if (version == 2 || version == 3) {
    h->so = sb[9];
    h->sl = sb[10];
    if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8))
      return h5fail(nc, "unsupported size of offsets / lengths");
    h->base = la(sb + 12, h->so);
    if (h->base == UNDEF)
      h->base = 0;
    h->eof = la(sb + 12 + 2 * (size_t) h->so, h->so);
    root = la(sb + 12 + 3 * (size_t) h->so, h->so);
  } else
    return h5fail(nc, "unknown superblock version");
  if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8))
    return h5fail(nc, "unsupported size of offsets / lengths");

  /* the root group: its attributes are the global attributes, its links the variables and dimensions */
  h5_msgs ms;
  h5_links L;
  memset(&L, 0, sizeof(L));
  if (!read_object_header(nc, root, &ms))
    return 0;
  att_list G;
  memset(&G, 0, sizeof(G));
  int ok = object_attributes(nc, &ms, &G) && group_links(nc, &ms, &L);
  free_msgs(&ms);
  nc->att = G.a;
  nc->natt = G.n;

  /* first pass: every dataset; dimension scales become dimensions */
  typedef struct {
    struct h5_dataset d;
    h5_type t;
    att_list A;
    int is_dataset, is_scale, no_variable, dim;
  } obj;
  obj *o = ok ? calloc((size_t) (L.n ? L.n : 1), sizeof(obj)) : NULL;
  if (ok && !o)
    ok = h5fail(nc, "out of memory");
  int dim_cap = 0;
  for (int i = 0; ok && i < L.n; i++) {
    ok = read_object_header(nc, L.addr[i], &ms) && parse_dataset(nc, &ms, &o[i].d, &o[i].t, &o[i].is_dataset)
      && (!o[i].is_dataset || object_attributes(nc, &ms, &o[i].A));
    free_msgs(&ms);
    if (!ok || !o[i].is_dataset)
      continue;
    const ncc_att *cls = find_att(&o[i].A, "CLASS"), *nm = find_att(&o[i].A, "NAME");
    if (cls && cls->type == T_CHAR && strncmp((const char *) cls->raw, "DIMENSION_SCALE", 15) == 0 && o[i].d.rank == 1) {
      o[i].is_scale = 1;
      o[i].no_variable = nm && nm->type == T_CHAR && strncmp((const char *) nm->raw, "This is a netCDF dimension but not a netCDF variable", 52) == 0;
      o[i].dim = find_or_add_dim(nc, L.name[i], o[i].d.dims[0], &dim_cap);
      if (o[i].dim < 0)
        ok = h5fail(nc, "out of memory");
    }
  }
  /* second pass: the variables */
  nc->var = ok ? calloc((size_t) (L.n ? L.n : 1), sizeof(ncc_var)) : NULL;
  if (ok && !nc->var)
    ok = h5fail(nc, "out of memory");
  for (int i = 0; ok && i < L.n; i++) {
    if (!o[i].is_dataset || o[i].no_variable)
      continue;
    const int type = nc_type_of(&o[i].t);
    if (!type)
      continue;   /* strings, compounds ...: not in the classic data model the host layer reads */
    ncc_var *v = &nc->var[nc->nvar++];   /* (counted at once: ncc_close releases it if what follows fails) */
    v->name = strdup(L.name[i]);
    if (!v->name) {
      ok = h5fail(nc, "out of memory");
      break;
    }
    v->ndims = o[i].d.rank;
    v->type = type;
    v->nelem = 1;
    for (int k = 0; ok && k < v->ndims; k++) {
      v->nelem *= (long long) o[i].d.dims[k];
      /* the axis belongs to the dimension scale its DIMENSION_LIST entry refers to; a coordinate variable is its
       * own scale; without the list (files not written by netCDF-4) the dimension of that length gives the name */
      int id = -1;
      if (o[i].is_scale && v->ndims == 1)
        id = o[i].dim;
      if (id < 0) {
        const uint64_t scale = axis_scale(nc, &o[i].A, k);
        for (int j = 0; scale != UNDEF && id < 0 && j < L.n; j++)
          if (L.addr[j] == scale && o[j].is_scale && o[j].d.dims[0] == o[i].d.dims[k])
            id = o[j].dim;
      }
#ifndef NC_HDF5_NO_LENGTH_FALLBACK   /* (defined by a test that wants to see the DIMENSION_LIST path alone) */
      for (int j = 0; id < 0 && j < nc->ndim; j++)
        if (nc->dim_len[j] == (long long) o[i].d.dims[k]) {
          int taken = 0;   /* (an axis of equal length earlier in this variable took that name) */
          for (int e = 0; e < k; e++)
            taken |= v->dimid[e] == j;
          if (!taken)
            id = j;
        }
#endif
      if (id < 0) {
        char anon[64];
        snprintf(anon, sizeof(anon), "phony_dim_%d", nc->ndim);
        id = find_or_add_dim(nc, anon, o[i].d.dims[k], &dim_cap);
        if (id < 0)
          ok = h5fail(nc, "out of memory");
      }
      v->dimid[k] = id;
    }
    if (!ok)
      break;
    if (o[i].d.layout == 1 && o[i].d.size == 0)   /* layout versions 1 / 2 give no byte size */
      o[i].d.size = (uint64_t) v->nelem * (uint64_t) o[i].d.elsize;
    v->h5 = malloc(sizeof(struct h5_dataset));
    if (!v->h5) {
      ok = h5fail(nc, "out of memory");
      break;
    }
    *v->h5 = o[i].d;
    o[i].d.compact = NULL;
    v->att = o[i].A.a;
    v->natt = o[i].A.n;
    o[i].A.a = NULL;
    o[i].A.n = 0;
  }
  for (int i = 0; o && i < L.n; i++) {
    free(o[i].d.compact);
    for (int k = 0; k < o[i].A.n; k++) {
      free(o[i].A.a[k].name);
      free(o[i].A.a[k].raw);
    }
    free(o[i].A.a);
  }
  free(o);
  for (int i = 0; i < L.n; i++)
    free(L.name[i]);
  free(L.name);
  free(L.addr);
  return ok;
}
