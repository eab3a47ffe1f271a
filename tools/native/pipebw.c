// Bandwidth of one pipe with 1 MB buffer between a writer and a reader thread (1 MB writes / reads): what a stage of the
// four-process chain can hand to the next.  gcc -O2 -pthread -o /tmp/pipebw tools/native/pipebw.c && /tmp/pipebw
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static int fds[2];
static size_t total = (size_t)4 << 30;
static void* reader(void* a) { char* b = malloc(1 << 20); size_t got = 0; while (got < total) { ssize_t r = read(fds[0], b, 1 << 20); if (r <= 0) break; got += r; } return 0; }
int main() {
    if (pipe(fds)) return 1;
    fcntl(fds[1], F_SETPIPE_SZ, 1 << 20);
    pthread_t t; pthread_create(&t, 0, reader, 0);
    char* b = malloc(1 << 20); memset(b, 'x', 1 << 20);
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t w = 0; w < total; w += 1 << 20) { size_t off = 0; while (off < (1 << 20)) { ssize_t r = write(fds[1], b + off, (1 << 20) - off); if (r < 0) return 1; off += r; } }
    pthread_join(t, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    printf("%.2f GB/s\n", total / s / 1e9);
    return 0;
}
