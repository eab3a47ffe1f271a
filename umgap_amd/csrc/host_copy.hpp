// host_copy.hpp -- memcpy between pageable memory and the pinned staging buffers of the host-pointer entry points, on a few threads:
// one thread moves ~10-18 GB/s, the link takes 56.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <thread>

namespace umgap {
inline void par_memcpy(void* dst, const void* src, size_t n) {
    constexpr size_t MIN_PER_THREAD = 4u << 20;
    unsigned t = (unsigned)std::min<size_t>(4, n / MIN_PER_THREAD);
    if (t <= 1) { memcpy(dst, src, n); return; }
    const size_t per = ((n / t) + 4095) & ~(size_t)4095;
    std::thread th[3];
    unsigned started = 0;
    for (unsigned i = 1; i < t; i++) {
        const size_t o = (size_t)i * per;
        if (o >= n) break;
        try {
            th[started] = std::thread([=] { memcpy((char*)dst + o, (const char*)src + o, std::min(per, n - o)); });
            started++;
        } catch (...) {       // no thread to be had: this one copies the rest (nothing throws across the C ABI)
            memcpy((char*)dst + o, (const char*)src + o, n - o);
            break;
        }
    }
    memcpy(dst, src, std::min(per, n));
    for (unsigned i = 0; i < started; i++) th[i].join();
}
}  // namespace umgap
