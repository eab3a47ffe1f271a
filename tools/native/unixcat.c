/* unixcat.c — what `nc -NU <socket>` does in scripts/umgap-analyse.sh:299 (netcat is not in this image): stdin -> the Unix
 * socket, the sending side shut at EOF (-N), the socket -> stdout.  Measurement harness of tools/end_to_end.py (the Python loop
 * that stood in for it before topped out near 1.8 GB/s, below what the chain moves).   cc -O2 -pthread -o unixcat unixcat.c */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

static int sock_fd;

static int write_all(int fd, const char* p, size_t n) {
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return -1; }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}

static void* feed(void* arg) {
    (void)arg;
    char* buf = malloc(1 << 20);
    for (;;) {
        ssize_t got = read(0, buf, 1 << 20);
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) break;
        if (write_all(sock_fd, buf, (size_t)got)) break;
    }
    shutdown(sock_fd, SHUT_WR);
    free(buf);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc != 2) { fprintf(stderr, "usage: unixcat <socket path>\n"); return 2; }
    struct sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    strncpy(a.sun_path, argv[1], sizeof a.sun_path - 1);
    sock_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (sock_fd < 0 || connect(sock_fd, (struct sockaddr*)&a, sizeof a) != 0) { perror("unixcat: connect"); return 1; }
#ifdef F_SETPIPE_SZ
    (void)fcntl(0, F_SETPIPE_SZ, 1 << 20);
    (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);
#endif
    pthread_t t;
    if (pthread_create(&t, NULL, feed, NULL) != 0) { perror("unixcat: thread"); return 1; }
    char* buf = malloc(1 << 20);
    int rc = 0;
    /* UNIXCAT_SPLICE=1: the socket's pages go into the pipe without passing through this process (splice); off by default -- netcat
     * copies, and with a copying relay this hop, not the stages, sets the chain's pace (about 2 GB/s of hit text) */
    const char* sp = getenv("UNIXCAT_SPLICE");
    if (sp && sp[0] == '1') {
        for (;;) {
            ssize_t got = splice(sock_fd, NULL, 1, NULL, 1 << 20, SPLICE_F_MOVE | SPLICE_F_MORE);
            if (got < 0 && errno == EINTR) continue;
            if (got < 0 && (errno == EINVAL || errno == ENOSYS)) break;   /* stdout is no pipe: copy below */
            if (got <= 0) { pthread_join(t, NULL); close(sock_fd); return got < 0; }
        }
    }
    for (;;) {
        ssize_t got = read(sock_fd, buf, 1 << 20);
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) break;
        if (write_all(1, buf, (size_t)got)) { rc = 1; break; }
    }
    pthread_join(t, NULL);
    close(sock_fd);
    return rc;
}
