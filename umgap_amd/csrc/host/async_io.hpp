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

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
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
        io.close = [](void* c) -> int {
            auto* p = (std::shared_ptr<In>*)c;
            (*p)->stop();
            {
                std::lock_guard<std::mutex> g(registry_mutex());
                auto& r = in_by_file();
                for (size_t i = 0; i < r.size(); i++)
                    if (r[i].second == *p) { r.erase(r.begin() + (long)i); break; }
            }
            delete p;
            return 0;
        };
        FILE* f = fopencookie(new std::shared_ptr<In>(st), "rb", io);
        if (f) {
            setvbuf(f, nullptr, _IOFBF, 1 << 16);
            std::lock_guard<std::mutex> g(registry_mutex());
            in_by_file().emplace_back(f, st);
        }
        return f;
    }
    // The descriptor behind a reader FILE* nothing has been read from yet: the caller reads it itself from now on (BlockReader's
    // filler reads straight into its block buffers: no pump thread, no 1 MB chunks, no copy); -1: not such a FILE*, or already in use.
    static int take_fd(FILE* f) {
        std::lock_guard<std::mutex> g(registry_mutex());
        for (auto& e : in_by_file())
            if (e.first == f) {
                std::lock_guard<std::mutex> lk(e.second->m);
                if (e.second->started) return -1;
                e.second->started = true;           // the cookie never starts its pump ...
                e.second->eof = true;               // ... and a stray fread on the FILE* sees the end of the stream
                e.second->taken = true;             // (an owned descriptor is then closed by fclose itself: no pump will)
                return e.second->fd;
            }
        return -1;
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
                auto& bf = by_file();
                for (size_t i = 0; i < bf.size(); i++)
                    if (bf[i].second == *p) { bf.erase(bf.begin() + (long)i); break; }
            }
            delete p;
            return 0;
        };
        FILE* f = fopencookie(new std::shared_ptr<Out>(st), "wb", io);
        if (f) {
            setvbuf(f, nullptr, _IOFBF, 1 << 16);
            std::lock_guard<std::mutex> g(registry_mutex());
            by_file().emplace_back(f, st);
        }
        return f;
    }
    // [p, p + n) written to `f` without a copy, behind everything written to f so far: true if f is one of these writers -- on_done then
    // runs (on the writer thread) when the bytes are out and the caller may touch them again; false: nothing was done
    static bool write_owned(FILE* f, const char* p, size_t n, std::function<void()> on_done) {
        std::shared_ptr<Out> o;
        {
            std::lock_guard<std::mutex> g(registry_mutex());
            for (auto& e : by_file()) if (e.first == f) { o = e.second; break; }
        }
        if (!o) return false;
        fflush(f);                      // what stdio still holds goes first
        o->write_owned(p, n, std::move(on_done));
        return true;
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

    // errno of the first write() that failed on any writer (0: none): the reference's commands end with `Error: <io error>`, exit 1, on
    // a failed write (writer.write_record(..)? in every command); here the writer thread notes it, the block loops stop at their next
    // block (BlockReader::next) and main reports it
    static int first_write_error() {
        std::lock_guard<std::mutex> g(registry_mutex());
        for (auto& o : registry()) if (const int e = o->err.load(std::memory_order_relaxed)) return e;
        return 0;
    }

  private:
    // 1 MB buffers are recycled: a fresh `new char[1 << 20]` per chunk is an mmap + 256 page faults + an munmap each time (glibc hands
    // blocks of that size straight to the kernel), i.e. every byte a stage reads or writes cost a zeroed page first -- a quarter of the
    // stage's time at 2 GB/s
    struct Pool {
        std::mutex m;
        std::vector<char*> free_list;
        ~Pool() { for (char* b : free_list) delete[] b; }
        char* get() {
            { std::lock_guard<std::mutex> g(m); if (!free_list.empty()) { char* b = free_list.back(); free_list.pop_back(); return b; } }
            return new char[CHUNK];
        }
        void put(char* b) {
            std::lock_guard<std::mutex> g(m);
            if (free_list.size() < 512) free_list.push_back(b); else delete[] b;
        }
    };
    static Pool& pool() { static Pool* p = new Pool(); return *p; }   // (never destroyed: detached reader threads may outlive main)
    struct Chunk {
        char* p = nullptr;
        size_t n = 0, off = 0, cap = 0;   // cap == CHUNK: the buffer came from the pool and goes back to it; cap == 0 with p set: the caller's
        std::function<void()> on_done;    // memory (write_owned), not freed here -- on_done tells the caller when the writer is through with it
        Chunk() = default;
        Chunk(const Chunk&) = delete;
        Chunk& operator=(const Chunk&) = delete;
        Chunk(Chunk&& o) noexcept : p(o.p), n(o.n), off(o.off), cap(o.cap), on_done(std::move(o.on_done)) { o.p = nullptr; o.on_done = nullptr; }
        Chunk& operator=(Chunk&& o) noexcept {
            if (this != &o) { release(); p = o.p; n = o.n; off = o.off; cap = o.cap; on_done = std::move(o.on_done); o.p = nullptr; o.on_done = nullptr; }
            return *this;
        }
        ~Chunk() { release(); }
        void alloc(size_t bytes) { release(); if (bytes <= CHUNK) { p = pool().get(); cap = CHUNK; } else { p = new char[bytes]; cap = bytes; } }
        void release() {
            if (p && cap) { if (cap == CHUNK) pool().put(p); else delete[] p; }
            p = nullptr;
            if (on_done) { auto f = std::move(on_done); on_done = nullptr; f(); }
        }
        char* get() const { return p; }
    };
    struct In {
        int fd = -1;
        std::mutex m;
        std::condition_variable data, space;
        std::deque<Chunk> q;
        size_t queued = 0;
        bool eof = false, stopped = false, started = false, own_fd = false, taken = false;
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
                c.alloc(CHUNK);
                ssize_t got;
                do got = ::read(fd, c.p, CHUNK); while (got < 0 && errno == EINTR);
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
                memcpy(buf + done, c.p + c.off, take);
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
                never_ran = !started || taken;       // (taken: the caller read the descriptor itself and is through with it by now)
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
        std::atomic<int> err{0};          // errno of the first failed write(): set by the writer thread, read by everybody
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
                while (off < c.n && !err.load(std::memory_order_relaxed)) {
                    ssize_t w = ::write(fd, c.p + off, c.n - off);
                    if (w < 0) { if (errno == EINTR) continue; err.store(errno, std::memory_order_relaxed); break; }
                    off += (size_t)w;
                }
                const size_t n_done = c.n;
                c.release();                      // (the caller's memory is its own again: before the queue reads as idle)
                std::lock_guard<std::mutex> lk(m);
                queued -= n_done;
                busy = false;
                space.notify_all();
                if (q.empty()) idle.notify_all();
            }
        }
        // the caller's bytes [p, p + n) go out as they lie (no copy); on_done runs on the writer thread when they are out (or dropped
        // after a write error) -- until then the caller must leave them alone
        void write_owned(const char* p, size_t n, std::function<void()> on_done) {
            if (err.load(std::memory_order_relaxed)) { if (on_done) on_done(); return; }   // the descriptor is dead: nothing more goes out (first_write_error)
            std::unique_lock<std::mutex> lk(m);
            space.wait(lk, [&] { return queued < WRITE_BEHIND; });
            Chunk c;
            c.p = const_cast<char*>(p);
            c.n = n;
            c.cap = 0;
            c.on_done = std::move(on_done);
            q.push_back(std::move(c));
            queued += n;
            data.notify_all();
        }
        ssize_t write(const char* buf, size_t n) {
            std::unique_lock<std::mutex> lk(m);
            if (const int e = err.load(std::memory_order_relaxed)) { errno = e; return 0; }
            space.wait(lk, [&] { return queued < WRITE_BEHIND; });
            size_t done_b = 0;
            // small writes join the last chunk while it has room
            if (!q.empty() && q.back().cap == CHUNK && q.back().off == 0 && q.back().n < CHUNK) {
                Chunk& c = q.back();
                const size_t take = std::min(n, CHUNK - c.n);
                memcpy(c.p + c.n, buf, take);
                c.n += take;
                done_b = take;
            }
            while (done_b < n) {
                Chunk c;
                const size_t take = std::min(n - done_b, CHUNK);   // (pool-sized pieces: a block of its own size would be fresh pages every time)
                c.alloc(CHUNK);
                memcpy(c.p, buf + done_b, take);
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
    static std::vector<std::pair<FILE*, std::shared_ptr<In>>>& in_by_file() { static std::vector<std::pair<FILE*, std::shared_ptr<In>>> r; return r; }
    static std::vector<std::pair<FILE*, std::shared_ptr<Out>>>& by_file() { static std::vector<std::pair<FILE*, std::shared_ptr<Out>>> r; return r; }
};

}  // namespace umgap_host
