// async_io.hpp — stdin / stdout (and the socket server's connections) decoupled from the stage's own work.
//
// The text stages of scripts/umgap-analyse.sh:299-304 are chained by pipes.  A stage that reads a block, works on it and writes
// the result in turn spends most of its wall time blocked in read() or write() on a 64 KB pipe -- each stage alone runs at
// 2.3-6 M pairs/s (tools/stage_times.py), the chain at under 1 M.  Here a reader thread keeps pulling the input into a queue of
// 1 MB chunks (up to a few blocks ahead) and a writer thread pushes finished output into the pipe, so a stage's pace is the
// slowest of read, work and write, not their sum.  Both sit behind ordinary FILE* handles (glibc fopencookie), so every
// fread / fwrite / printf of the command code is unchanged; pipes are also grown to 1 MB (F_SETPIPE_SZ) where the kernel allows.
// UMGAP_ASYNC_IO=0 switches it off (A/B measurements).
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace umgap_host {

class AsyncIo {
  public:
    static constexpr size_t CHUNK = 1u << 20, READ_AHEAD = 192u << 20, WRITE_BEHIND = 192u << 20;

    static bool enabled() {
        const char* e = getenv("UMGAP_ASYNC_IO");
        return !(e && e[0] == '0');
    }
    static void grow_pipe(int fd) {
#ifdef F_SETPIPE_SZ
        (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);   // fails harmlessly on anything that is not a pipe
#endif
    }

    // A FILE* that reads what a background thread pulls from `fd`.  The state is shared with the (detached) thread: a reader that
    // is still blocked in read() when the command ends -- input it never asked for -- simply dies with the process.
    // The thread starts with the first read (a command that never reads its input leaves it alone); own_fd: close it when done.
    static FILE* reader(int fd, bool own_fd = false) {
        auto st = std::make_shared<In>();
        st->fd = fd;
        st->own_fd = own_fd;
        grow_pipe(fd);
        cookie_io_functions_t io{};
        io.read = [](void* c, char* buf, size_t n) -> ssize_t {
            auto& sp = *(std::shared_ptr<In>*)c;
            if (!sp->started) { sp->started = true; auto keep = sp; std::thread([keep]() { keep->pump(); }).detach(); }
            return sp->read(buf, n);
        };
        io.close = [](void* c) -> int { auto* p = (std::shared_ptr<In>*)c; (*p)->stop(); delete p; return 0; };
        FILE* f = fopencookie(new std::shared_ptr<In>(st), "rb", io);
        if (f) setvbuf(f, nullptr, _IOFBF, 1 << 16);
        return f;
    }
    // A FILE* whose bytes a background thread writes to `fd`.  fclose (or drain_all at exit) waits until everything is out.
    static FILE* writer(int fd, bool own_fd = false) {
        auto st = std::make_shared<Out>();
        st->fd = fd;
        st->own_fd = own_fd;
        grow_pipe(fd);
        st->th = std::thread([st]() { st->pump(); });
        {
            std::lock_guard<std::mutex> g(registry_mutex());
            registry().push_back(st);
        }
        cookie_io_functions_t io{};
        io.write = [](void* c, const char* buf, size_t n) -> ssize_t { return (*(std::shared_ptr<Out>*)c)->write(buf, n); };
        io.close = [](void* c) -> int {
            auto* p = (std::shared_ptr<Out>*)c;
            (*p)->finish();
            {
                std::lock_guard<std::mutex> g(registry_mutex());
                auto& r = registry();
                for (size_t i = 0; i < r.size(); i++)
                    if (r[i] == *p) { r.erase(r.begin() + (long)i); break; }
            }
            delete p;
            return 0;
        };
        FILE* f = fopencookie(new std::shared_ptr<Out>(st), "wb", io);
        if (f) setvbuf(f, nullptr, _IOFBF, 1 << 16);
        return f;
    }
    // everything handed to any writer so far has reached its descriptor (call after fflush)
    static void drain_all() {
        std::vector<std::shared_ptr<Out>> all;
        {
            std::lock_guard<std::mutex> g(registry_mutex());
            all = registry();
        }
        for (auto& o : all) o->drain();
    }

  private:
    struct Chunk {
        std::unique_ptr<char[]> p;
        size_t n = 0, off = 0;
    };
    struct In {
        int fd = -1;
        std::mutex m;
        std::condition_variable data, space;
        std::deque<Chunk> q;
        size_t queued = 0;
        bool eof = false, stopped = false, started = false, own_fd = false;
        int err = 0;
        void pump() {
            struct Closer { In* in; ~Closer() { if (in->own_fd) ::close(in->fd); } } closer{this};
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(m);
                    space.wait(lk, [&] { return queued < READ_AHEAD || stopped; });
                    if (stopped) return;
                }
                Chunk c;
                c.p.reset(new char[CHUNK]);
                ssize_t got;
                do got = ::read(fd, c.p.get(), CHUNK); while (got < 0 && errno == EINTR);
                std::lock_guard<std::mutex> lk(m);
                if (got <= 0) { if (got < 0) err = errno; eof = true; data.notify_all(); return; }
                c.n = (size_t)got;
                queued += c.n;
                q.push_back(std::move(c));
                data.notify_all();
            }
        }
        ssize_t read(char* buf, size_t n) {
            std::unique_lock<std::mutex> lk(m);
            data.wait(lk, [&] { return !q.empty() || eof; });
            if (q.empty()) { if (err) { errno = err; return -1; } return 0; }
            size_t done = 0;
            while (done < n && !q.empty()) {          // whatever is there, up to n: never waits for more than one byte
                Chunk& c = q.front();
                const size_t take = std::min(n - done, c.n - c.off);
                memcpy(buf + done, c.p.get() + c.off, take);
                c.off += take;
                done += take;
                if (c.off == c.n) { queued -= c.n; q.pop_front(); }
            }
            space.notify_all();
            return (ssize_t)done;
        }
        void stop() {
            bool never_ran;
            {
                std::lock_guard<std::mutex> lk(m);
                stopped = true;
                never_ran = !started;
                space.notify_all();
            }
            if (own_fd) {
                if (never_ran) ::close(fd);
                else (void)::shutdown(fd, SHUT_RD);      // wakes a pump blocked in read(); it closes the descriptor on its way out
            }
        }
    };
    struct Out {
        int fd = -1;
        std::thread th;
        std::mutex m;
        std::condition_variable data, space, idle;
        std::deque<Chunk> q;
        size_t queued = 0;
        bool done = false, busy = false, own_fd = false;
        int err = 0;
        void pump() {
            for (;;) {
                Chunk c;
                {
                    std::unique_lock<std::mutex> lk(m);
                    data.wait(lk, [&] { return !q.empty() || done; });
                    if (q.empty()) { idle.notify_all(); return; }
                    c = std::move(q.front());
                    q.pop_front();
                    busy = true;
                }
                size_t off = 0;
                while (off < c.n && !err) {
                    ssize_t w = ::write(fd, c.p.get() + off, c.n - off);
                    if (w < 0) { if (errno == EINTR) continue; err = errno; break; }
                    off += (size_t)w;
                }
                std::lock_guard<std::mutex> lk(m);
                queued -= c.n;
                busy = false;
                space.notify_all();
                if (q.empty()) idle.notify_all();
            }
        }
        ssize_t write(const char* buf, size_t n) {
            std::unique_lock<std::mutex> lk(m);
            if (err) { errno = err; return 0; }
            space.wait(lk, [&] { return queued < WRITE_BEHIND; });
            size_t done_b = 0;
            // small writes join the last chunk while it has room; large ones get chunks of their own size
            if (!q.empty() && q.back().off == 0 && q.back().n < CHUNK) {
                Chunk& c = q.back();
                const size_t take = std::min(n, CHUNK - c.n);
                memcpy(c.p.get() + c.n, buf, take);
                c.n += take;
                done_b = take;
            }
            while (done_b < n) {
                Chunk c;
                const size_t take = n - done_b;
                c.p.reset(new char[std::max(take, CHUNK)]);
                memcpy(c.p.get(), buf + done_b, take);
                c.n = take;
                done_b += take;
                q.push_back(std::move(c));
            }
            queued += n;
            data.notify_all();
            return (ssize_t)n;
        }
        void drain() {
            std::unique_lock<std::mutex> lk(m);
            idle.wait(lk, [&] { return (q.empty() && !busy) || done; });
        }
        void finish() {
            {
                std::unique_lock<std::mutex> lk(m);
                idle.wait(lk, [&] { return q.empty() && !busy; });
                done = true;
                data.notify_all();
            }
            if (th.joinable()) th.join();
            if (own_fd) ::close(fd);
        }
    };
    static std::mutex& registry_mutex() { static std::mutex m; return m; }
    static std::vector<std::shared_ptr<Out>>& registry() { static std::vector<std::shared_ptr<Out>> r; return r; }
};

}  // namespace umgap_host
