// queue_probe.hip -- which pairs of HIP streams of one process overlap a host-to-device copy with a kernel?  The runtime maps a
// process's streams onto a few hardware queues; two streams on the same queue run their work back to back.  Creates N streams one
// after the other and, for every ordered pair (a, b), times a 64 MB pinned upload on a together with a ~1.5 ms kernel on b.
//   hipcc --offload-arch=gfx950 -O2 -o tools/native/queue_probe tools/native/queue_probe.hip && tools/native/queue_probe [N]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long cycles, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<hipStream_t> st(N);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t bytes = 64u << 20;
    void *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, bytes, hipHostMallocDefault);
    hipMalloc(&d, bytes);
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto run = [&](int a, int b, bool copy, bool kern) {
        hipDeviceSynchronize();
        const double t0 = now();
        if (kern) hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, st[b], 150000ull, (int*)nullptr);   // 100 MHz wall clock: 1.5 ms
        if (copy) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[a]);
        hipDeviceSynchronize();
        return now() - t0;
    };
    run(0, 1, true, true);
    const double tc = run(0, 0, true, false), tk = run(0, 0, false, true);
    printf("copy alone %.2f ms, kernel alone %.2f ms\n   ", tc, tk);
    for (int b = 0; b < N; b++) printf(" k%-5d", b);
    printf("\n");
    for (int a = 0; a < N; a++) {
        printf("c%-2d", a);
        for (int b = 0; b < N; b++) {
            double best = 1e9;
            for (int r = 0; r < 3; r++) best = std::min(best, run(a, b, true, true));
            printf(" %6.2f", best);
        }
        printf("\n");
    }
    // kernel on a + kernel on b (64 workgroups each: both fit the chip at once): 1.5 ms when the two streams sit on different hardware
    // queues, 3 ms when they share one
    auto run2 = [&](int a, int b) {
        hipDeviceSynchronize();
        const double t0 = now();
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[a], 150000ull, (int*)nullptr);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[b], 150000ull, (int*)nullptr);
        hipDeviceSynchronize();
        return now() - t0;
    };
    printf("kernel + kernel\n   ");
    for (int b = 0; b < N; b++) printf(" k%-5d", b);
    printf("\n");
    for (int a = 0; a < N; a++) {
        printf("k%-2d", a);
        for (int b = 0; b < N; b++) {
            double best = 1e9;
            for (int r = 0; r < 3; r++) best = std::min(best, run2(a, b));
            printf(" %6.2f", best);
        }
        printf("\n");
    }
    // a fill (memset kernel) + an upload on a, while a kernel runs on b: the upload lane of records.hip as it was
    printf("memset + copy on a, kernel on b\n");
    for (int a = 0; a < N; a++) {
        printf("c%-2d", a);
        for (int b = 0; b < N; b++) {
            double best = 1e9;
            for (int r = 0; r < 3; r++) {
                hipDeviceSynchronize();
                const double t0 = now();
                hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, st[b], 150000ull, (int*)nullptr);
                hipMemsetAsync(d, 0, 256, st[a]);
                hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[a]);
                hipDeviceSynchronize();
                best = std::min(best, now() - t0);
            }
            printf(" %6.2f", best);
        }
        printf("\n");
    }
    // the pipeline of umgap_index_lookup9: 16 chunks of {32 MB up on a; wait, 0.1 ms kernel, 16 MB down on b}, three slots, events
    // per slot -- and the same with the download on a third stream c = (b + 1) % N
    {
        const size_t up_b = 32u << 20, dn_b = 16u << 20;
        hipEvent_t upl[3], done[3], kdone[3];
        for (int i = 0; i < 3; i++) { hipEventCreateWithFlags(&upl[i], hipEventDisableTiming); hipEventCreateWithFlags(&done[i], hipEventDisableTiming); hipEventCreateWithFlags(&kdone[i], hipEventDisableTiming); }
        for (int three = 0; three < 2; three++) {
            printf(three ? "pipeline, download on a third stream\n" : "pipeline, download on the kernel's stream\n");
            for (int a = 0; a < N; a++) {
                printf("a%-2d", a);
                for (int b = 0; b < N; b++) {
                    if (a == b) { printf("      -"); continue; }
                    int c3 = (b + 1) % N;
                    if (c3 == a) c3 = (c3 + 1) % N;
                    double best = 1e9;
                    for (int r = 0; r < 2; r++) {
                        hipDeviceSynchronize();
                        const double t0 = now();
                        for (int c = 0; c < 16; c++) {
                            const int sl = c % 3;
                            if (c >= 3) hipEventSynchronize(done[sl]);
                            hipMemcpyAsync((char*)d + (size_t)sl * (16u << 20), h, up_b / 2, hipMemcpyHostToDevice, st[a]);
                            hipMemcpyAsync((char*)d + (size_t)sl * (16u << 20), h, up_b / 2, hipMemcpyHostToDevice, st[a]);
                            hipEventRecord(upl[sl], st[a]);
                            hipStreamWaitEvent(st[b], upl[sl], 0);
                            hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, st[b], 10000ull, (int*)nullptr);
                            hipStream_t ds = st[b];
                            if (three) { hipEventRecord(kdone[sl], st[b]); hipStreamWaitEvent(st[c3], kdone[sl], 0); ds = st[c3]; }
                            hipMemcpyAsync((char*)h + (32u << 20), (char*)d + (size_t)sl * (16u << 20), dn_b, hipMemcpyDeviceToHost, ds);
                            hipEventRecord(done[sl], ds);
                        }
                        hipDeviceSynchronize();
                        best = std::min(best, now() - t0);
                    }
                    printf(" %6.2f", best);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
