/*
 * ctlfile.c -- control parameters and calendar arithmetic of the host layer.
 *
 * Behaviour follows the reference's documentation of its control files
 * (docs/manual/control-parameters.md) and of scan_ctl / jsec2time / time2jsec
 * (src/mptrac.h); the implementation is this repository's own:
 *
 *  - a control file is parsed once into a table (the reference re-reads the
 *    file for every key); look-ups then follow the documented rules;
 *  - dates are converted with the civil-calendar day-number algorithm
 *    (proleptic Gregorian), not through the C library's struct tm.
 */
#define _GNU_SOURCE
#include "mptrac.h"

#include <ctype.h>
#include <strings.h>
#include <sys/stat.h>

/* ---- control file table ---------------------------------------------------- */

/* "NAME <any token> VALUE" lines of one control file, in file order */
static struct {
  char *path;       /* the file the table belongs to (NULL: none loaded) */
  struct timespec mtime;   /* what the file looked like when it was parsed: a file rewritten under the */
  off_t size;              /* same name (an ensemble driver, a test) is parsed again */
  ino_t inode;
  char **name;
  char **value;
  size_t n, cap;
} g_ctlfile;

static char *dup_token(const char *s, size_t len) {
  char *d = malloc(len + 1);
  if (!d)
    ERRMSG("Out of memory!");
  memcpy(d, s, len);
  d[len] = '\0';
  return d;
}

/* next blank-separated token of *s (at most LEN - 1 characters are kept, like the reference's "%4999s") */
static int next_token(const char **s, const char **tok, size_t *len) {
  const char *p = *s;
  while (*p && isspace((unsigned char) *p))
    p++;
  if (!*p)
    return 0;
  const char *q = p;
  while (*q && !isspace((unsigned char) *q))
    q++;
  *tok = p;
  *len = (size_t) (q - p) < (size_t) (LEN - 1) ? (size_t) (q - p) : (size_t) (LEN - 1);
  *s = q;
  return 1;
}

static void ctlfile_drop(void) {
  for (size_t i = 0; i < g_ctlfile.n; i++) {
    free(g_ctlfile.name[i]);
    free(g_ctlfile.value[i]);
  }
  free(g_ctlfile.name);
  free(g_ctlfile.value);
  free(g_ctlfile.path);
  memset(&g_ctlfile, 0, sizeof(g_ctlfile));
}

static void ctlfile_load(const char *path) {
  struct stat st;
  const int have_stat = stat(path, &st) == 0;
  if (g_ctlfile.path && strcmp(g_ctlfile.path, path) == 0 && have_stat && st.st_size == g_ctlfile.size
      && st.st_ino == g_ctlfile.inode && st.st_mtim.tv_sec == g_ctlfile.mtime.tv_sec
      && st.st_mtim.tv_nsec == g_ctlfile.mtime.tv_nsec)
    return;
  ctlfile_drop();
  FILE *in = fopen(path, "r");
  if (!in)
    ERRMSG("Cannot open file!");
  if (fstat(fileno(in), &st) != 0)
    memset(&st, 0, sizeof(st));
  char line[LEN];
  while (fgets(line, LEN, in)) {
    /* a setting needs three tokens: the name, a separator (conventionally "="), the value */
    const char *s = line, *t[3];
    size_t l[3];
    if (!next_token(&s, &t[0], &l[0]) || !next_token(&s, &t[1], &l[1]) || !next_token(&s, &t[2], &l[2]))
      continue;
    if (g_ctlfile.n == g_ctlfile.cap) {
      g_ctlfile.cap = g_ctlfile.cap ? 2 * g_ctlfile.cap : 64;
      g_ctlfile.name = realloc(g_ctlfile.name, g_ctlfile.cap * sizeof(char *));
      g_ctlfile.value = realloc(g_ctlfile.value, g_ctlfile.cap * sizeof(char *));
      if (!g_ctlfile.name || !g_ctlfile.value)
        ERRMSG("Out of memory!");
    }
    g_ctlfile.name[g_ctlfile.n] = dup_token(t[0], l[0]);
    g_ctlfile.value[g_ctlfile.n] = dup_token(t[2], l[2]);
    g_ctlfile.n++;
  }
  fclose(in);
  g_ctlfile.path = dup_token(path, strlen(path));
  g_ctlfile.mtime = st.st_mtim;
  g_ctlfile.size = st.st_size;
  g_ctlfile.inode = st.st_ino;
}

/* Forget the parsed control file: the next look-up reads the file again.  mptrac_read_ctl starts with this, so
 * that every call sees the file as it is now (the reference re-reads it for every key). */
void ctlfile_invalidate(void) {
  ctlfile_drop();
}

/* Value of a control parameter (reference interface: src/mptrac.h scan_ctl).
 *   - settings come from the control file ("NAME = VALUE" lines, first match wins) and are overridden by
 *     "NAME VALUE" pairs among the trailing command-line arguments (first match wins);
 *   - names are case-insensitive; element i of an array parameter is NAME[i], NAME[*] sets all elements;
 *   - a file name that ends in '-' means: arguments only;
 *   - a parameter that is set nowhere takes `defvalue`; an empty `defvalue` makes it mandatory.
 * Returns the value as a number and, if `value` is given, copies its text. */
double scan_ctl(const char *filename, int argc, char *argv[], const char *varname, const int arridx,
                const char *defvalue, char *value) {
  char exact[LEN], wild[LEN];
  if (arridx >= 0) {
    snprintf(exact, LEN, "%s[%d]", varname, arridx);
    snprintf(wild, LEN, "%s[*]", varname);
  } else {
    snprintf(exact, LEN, "%s", varname);
    snprintf(wild, LEN, "%s", varname);
  }
  const char *found = NULL;
  const size_t flen = strlen(filename);
  if (flen == 0 || filename[flen - 1] != '-') {
    ctlfile_load(filename);
    for (size_t i = 0; i < g_ctlfile.n && !found; i++)
      if (strcasecmp(g_ctlfile.name[i], exact) == 0 || strcasecmp(g_ctlfile.name[i], wild) == 0)
        found = g_ctlfile.value[i];
  }
  for (int i = 1; i + 1 < argc; i++)
    if (strcasecmp(argv[i], exact) == 0 || strcasecmp(argv[i], wild) == 0) {
      found = argv[i + 1];
      break;
    }
  if (!found) {
    if (defvalue[0] == '\0')
      ERRMSG("Missing variable %s!\n", exact);
    found = defvalue;
  }
  char text[LEN];
  snprintf(text, LEN, "%s", found);
  LOG(1, "%s = %s", exact, text);
  if (value != NULL)
    strcpy(value, text);
  return atof(text);
}

/* ---- calendar ---------------------------------------------------------------- */

/* days from 2000-01-01 to year-month-day of the proleptic Gregorian calendar */
static long long days_from_2000(long long y, int m, int d) {
  /* shift the year so that it starts on 1 March: leap days fall at the end */
  y -= m <= 2;
  const long long era = (y >= 0 ? y : y - 399) / 400;
  const long long yoe = y - era * 400;                                   /* [0, 399] */
  const long long doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;  /* [0, 365] */
  const long long doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;           /* [0, 146096] */
  return era * 146097 + doe - 730425;   /* 730425 days from 0000-03-01 to 2000-01-01 */
}

static void civil_from_days(long long z, int *y, int *m, int *d) {
  z += 730425;
  const long long era = (z >= 0 ? z : z - 146096) / 146097;
  const long long doe = z - era * 146097;
  const long long yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const long long doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const long long mp = (5 * doy + 2) / 153;
  *d = (int) (doy - (153 * mp + 2) / 5 + 1);
  *m = (int) (mp < 10 ? mp + 3 : mp - 9);
  *y = (int) (yoe + era * 400 + (*m <= 2));
}

/* Seconds since 2000-01-01T00:00Z -> calendar date (reference interface: src/mptrac.h jsec2time).  The whole
 * seconds are those of the value truncated towards zero, `remain` is its distance to the next lower whole
 * number -- the reference's conversion through time_t behaves that way for negative fractions too. */
void jsec2time(const double jsec, int *year, int *mon, int *day, int *hour, int *min, int *sec,
               double *remain) {
  const long long whole = (long long) jsec;
  long long days = whole / 86400, rest = whole % 86400;
  if (rest < 0) {
    rest += 86400;
    days--;
  }
  civil_from_days(days, year, mon, day);
  *hour = (int) (rest / 3600);
  *min = (int) (rest / 60 % 60);
  *sec = (int) (rest % 60);
  *remain = jsec - floor(jsec);
}

/* calendar date -> seconds since 2000-01-01T00:00Z (reference interface: src/mptrac.h time2jsec); fields
 * outside their usual range carry over (month 13 = January of the next year, ...) as timegm does */
void time2jsec(const int year, const int mon, const int day, const int hour, const int min,
               const int sec, const double remain, double *jsec) {
  long long y = year, m0 = (long long) mon - 1;
  y += m0 / 12;
  m0 %= 12;
  if (m0 < 0) {
    m0 += 12;
    y--;
  }
  const long long days = days_from_2000(y, (int) m0 + 1, 1) + (day - 1);
  *jsec = (double) (days * 86400 + (long long) hour * 3600 + (long long) min * 60 + sec) + remain;
}
