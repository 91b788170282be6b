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
#include <netinet/in.h>
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

/* rank 0 sends buf[0 .. n) to each of the world - 1 other ranks; they receive it.  1 = ok. */
int mptrac_amd_bcast(void *buf, size_t n, int rank, int world, const char *addr, int port) {
  if (world <= 1)
    return 1;
  struct sockaddr_in sa;
  memset(&sa, 0, sizeof(sa));
  sa.sin_family = AF_INET;
  sa.sin_port = htons((unsigned short) port);
  if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1)
    return 0;
  if (rank == 0) {
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    if (ls < 0)
      return 0;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (bind(ls, (struct sockaddr *) &sa, sizeof(sa)) != 0 || listen(ls, world) != 0) {
      close(ls);
      return 0;
    }
    int ok = 1;
    for (int k = 1; k < world && ok; k++) {
      const int fd = accept(ls, NULL, NULL);
      ok = fd >= 0 && send_all(fd, buf, n);
      if (fd >= 0)
        close(fd);
    }
    close(ls);
    return ok;
  }
  /* the other ranks: rank 0 may not listen yet */
  for (int attempt = 0; attempt < 600; attempt++) {
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0)
      return 0;
    if (connect(fd, (struct sockaddr *) &sa, sizeof(sa)) == 0) {
      const int ok = recv_all(fd, buf, n);
      close(fd);
      return ok;
    }
    close(fd);
    struct timespec ts = { 0, 100 * 1000 * 1000 };
    nanosleep(&ts, NULL);
  }
  return 0;
}
