/*
 * rendezvous.c -- hands a small blob from rank 0 to the other ranks of a single-node job over TCP.
 *
 * The multi-GPU driver needs exactly one exchange outside RCCL: the 128-byte communicator identifier that
 * rank 0 creates (mphip_comm_unique_id) has to reach the other processes before ncclCommInitRank can run.
 * The reference's driver would use MPI for this (src/trac.c:70-81); the image has none, so the ranks meet at
 * MASTER_ADDR : MASTER_PORT + 1 (the environment a launcher such as torch.distributed.run exports; + 1
 * keeps clear of a launcher's own store on MASTER_PORT).
 */
#define _GNU_SOURCE
#include "mptrac.h"

#include <arpa/inet.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

static int send_all(int fd, const char *p, size_t n) {
  while (n) {
    const ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0)
      return 0;
    p += k;
    n -= (size_t) k;
  }
  return 1;
}

static int recv_all(int fd, char *p, size_t n) {
  while (n) {
    const ssize_t k = recv(fd, p, n, 0);
    if (k <= 0)
      return 0;
    p += k;
    n -= (size_t) k;
  }
  return 1;
}

/* MASTER_ADDR may be a dotted address or a host name ("localhost", the node's name as a launcher exports it) */
static int resolve_ipv4(const char *addr, int port, struct sockaddr_in *sa) {
  memset(sa, 0, sizeof(*sa));
  sa->sin_family = AF_INET;
  sa->sin_port = htons((unsigned short) port);
  if (inet_pton(AF_INET, addr, &sa->sin_addr) == 1)
    return 1;
  struct addrinfo hints, *res = NULL;
  memset(&hints, 0, sizeof(hints));
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(addr, NULL, &hints, &res) != 0 || !res)
    return 0;
  sa->sin_addr = ((struct sockaddr_in *) res->ai_addr)->sin_addr;
  freeaddrinfo(res);
  return 1;
}

/* seconds rank 0 waits for all peers, and a peer for rank 0 (MPTRAC_RENDEZVOUS_TIMEOUT, default 120) */
static int rendezvous_timeout(void) {
  const char *e = getenv("MPTRAC_RENDEZVOUS_TIMEOUT");
  const int s = e ? atoi(e) : 0;
  return s > 0 ? s : 120;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* what a peer sends before it is given the buffer */
typedef struct {
  unsigned magic;
  int rank, world;
  unsigned long long bytes;
} hello_t;
#define HELLO_MAGIC 0x4d505452u   /* "MPTR" */

/* rank 0 sends buf[0 .. n) to each of the world - 1 other ranks; they receive it.  1 = ok, 0 = failed or timed
 * out (a peer that died before it connected must not leave rank 0 waiting for ever). */
int mptrac_amd_bcast(void *buf, size_t n, int rank, int world, const char *addr, int port) {
  if (world <= 1)
    return 1;
  const double deadline = now_s() + rendezvous_timeout();
  if (rank == 0) {
    /* listen on every interface: the peers reach this process through whatever MASTER_ADDR resolves to for them */
    struct sockaddr_in any;
    memset(&any, 0, sizeof(any));
    any.sin_family = AF_INET;
    any.sin_port = htons((unsigned short) port);
    any.sin_addr.s_addr = htonl(INADDR_ANY);
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    if (ls < 0)
      return 0;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (bind(ls, (struct sockaddr *) &any, sizeof(any)) != 0 || listen(ls, world) != 0) {
      close(ls);
      return 0;
    }
    /* every peer says who it is first (hello_t) and acknowledges its copy with one byte; a connection that does
     * neither is dropped without using up a place, and a rank whose copy got lost on the way is served again when it
     * comes back (the buffer is the same every time) -- until world - 1 different ranks have acknowledged */
    int ok = 1, served = 0;
    unsigned char *seen = calloc((size_t) world, 1);
    if (!seen) {
      close(ls);
      return 0;
    }
    while (served < world - 1 && ok) {
      struct pollfd pfd = { ls, POLLIN, 0 };
      const double left = deadline - now_s();
      if (left <= 0 || poll(&pfd, 1, (int) (left * 1000.0) + 1) <= 0) {
        ok = 0;
        break;
      }
      const int fd = accept(ls, NULL, NULL);
      if (fd < 0)
        continue;
      struct timeval tv = { 1, 0 };      /* (a peer sends its hello at once: a silent connection holds the server for a second at most) */
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      hello_t hello;
      char ack = 0;
      if (recv_all(fd, (char *) &hello, sizeof(hello)) && hello.magic == HELLO_MAGIC && hello.world == world && hello.rank >= 1
          && hello.rank < world && hello.bytes == (unsigned long long) n) {
        struct timeval tv_ack = { 5, 0 };
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv_ack, sizeof(tv_ack));
        /* a rank counts as served once the blob went out: the peer returns as soon as it has the blob (its
         * acknowledgement is advisory -- it only keeps the connection open until the copy is complete), so a lost or
         * late acknowledgement must not leave rank 0 waiting for a peer that has moved on */
        if (send_all(fd, buf, n)) {
          (void) recv_all(fd, &ack, 1);
          if (!seen[hello.rank]) {
            seen[hello.rank] = 1;
            served++;
          }
        }
      }
      close(fd);
    }
    free(seen);
    close(ls);
    return ok;
  }
  struct sockaddr_in sa;
  if (!resolve_ipv4(addr, port, &sa))
    return 0;
  /* the other ranks: rank 0 may not listen yet */
  while (now_s() < deadline) {
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0)
      return 0;
    if (connect(fd, (struct sockaddr *) &sa, sizeof(sa)) == 0) {
      struct timeval tv = { rendezvous_timeout(), 0 };
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      hello_t hello;
      memset(&hello, 0, sizeof(hello));      /* (padding bytes included: nothing uninitialised goes over the wire) */
      hello.magic = HELLO_MAGIC;
      hello.rank = rank;
      hello.world = world;
      hello.bytes = (unsigned long long) n;
      const char ack = 1;
      const int ok = send_all(fd, (const char *) &hello, sizeof(hello)) && recv_all(fd, buf, n) && send_all(fd, &ack, 1);
      close(fd);
      if (ok)
        return 1;
      /* (rank 0 dropped the connection -- e.g. it was still serving a run before this one -- : once more) */
    } else
      close(fd);
    struct timespec ts = { 0, 100 * 1000 * 1000 };
    nanosleep(&ts, NULL);
  }
  return 0;
}
