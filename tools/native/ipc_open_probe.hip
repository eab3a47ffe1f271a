// ipc_open_probe — how long hipIpcOpenMemHandle takes for allocations of given sizes between two processes on one GPU (measurement;
// found: see DESIGN.md section 7).  For each size on the command line: the exporter hipMallocs it and hands the handle over a pipe, the
// importer opens it (15 s alarm), touches it with a memset and closes it.   hipcc -o ipc_open_probe ipc_open_probe.hip
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static void on_alarm(int) { const char m[] = "  -> no answer within 15 s\n"; (void)!write(2, m, sizeof(m) - 1); _exit(3); }

int main(int argc, char** argv) {
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(strtoull(argv[i], nullptr, 0));
    for (size_t n : sizes) {
        int to_imp[2], to_exp[2];
        if (pipe(to_imp) || pipe(to_exp)) return 2;
        const pid_t exporter = fork();
        if (exporter == 0) {
            void* p = nullptr;
            if (hipMalloc(&p, n) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu) failed\n", n); _exit(4); }
            hipIpcMemHandle_t h;
            const hipError_t e = hipIpcGetMemHandle(&h, p);
            if (e != hipSuccess) { fprintf(stderr, "hipIpcGetMemHandle: %s\n", hipGetErrorString(e)); _exit(5); }
            (void)!write(to_imp[1], &h, sizeof(h));
            char done;
            (void)!read(to_exp[0], &done, 1);
            (void)hipFree(p);
            _exit(0);
        }
        const pid_t importer = fork();
        if (importer == 0) {
            hipIpcMemHandle_t h;
            if (read(to_imp[0], &h, sizeof(h)) != (ssize_t)sizeof(h)) _exit(6);
            (void)hipFree(nullptr);
            signal(SIGALRM, on_alarm);
            alarm(15);
            const auto t0 = std::chrono::steady_clock::now();
            void* q = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            alarm(0);
            if (e != hipSuccess) { fprintf(stderr, "  hipIpcOpenMemHandle: %s\n", hipGetErrorString(e)); _exit(7); }
            const hipError_t m = hipMemset((char*)q + n - 4096, 1, 4096);
            fprintf(stderr, "  opened in %.3f s, memset of the last page: %s\n", dt, hipGetErrorString(m));
            (void)hipIpcCloseMemHandle(q);
            _exit(0);
        }
        fprintf(stderr, "size %zu (%.1f MB; mod 2 MB = %zu):\n", n, n / 1048576.0, n % (2u << 20));
        int st = 0;
        waitpid(importer, &st, 0);
        (void)!write(to_exp[1], "x", 1);
        int st2 = 0;
        waitpid(exporter, &st2, 0);
        for (int fd : {to_imp[0], to_imp[1], to_exp[0], to_exp[1]}) close(fd);
    }
    return 0;
}
