// fastpath.hpp — block-parallel FASTA tokenizer for the `umgap` command (host side, no GPU).
//
// The reference reads stdin through a 10 MB BufReader and allocates a String per line
// (src/io/fasta.rs:14,31,38-67); at GPU lookup rates that is the bottleneck (SURVEY.md §7 "Host
// text I/O is the real ceiling").  Here stdin is read in large blocks cut at record boundaries
// (a line that starts with '>'), each block is split into record-aligned ranges that worker threads
// tokenize in place (no per-line allocation), one device call serves the whole block, and the
// workers format their ranges' output, which is written in input order.
//
// Semantics kept from src/io/fasta.rs: lines end at '\n', one trailing '\r' is stripped, a last
// line without '\n' counts, the first line of the stream must start with '>', every later line
// that does not start with '>' is an item of the current record (empty lines included), invalid
// UTF-8 ends the stream with an error AFTER the record that was being read has been emitted
// truncated at the bad line (the Peekable stops at the Err, the next read_record returns it).
#pragma once
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include <sys/socket.h>

#include "async_io.hpp"
#include "text.hpp"

namespace umgap_host {

inline unsigned worker_count() {
    const char* env = getenv("UMGAP_THREADS");
    if (env && *env) {
        int v = atoi(env);
        if (v >= 1) return (unsigned)std::min(v, 64);
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    return std::min(hw, 16u);
}

// Reads `f` in blocks that end right before a record start (or at EOF).
class BlockReader {
  public:
    explicit BlockReader(FILE* f, size_t block = default_block()) : f_(f), block_(block) {}
    // The view handed out by next() then stays valid until the call AFTER the next one (two buffers take turns): a caller can
    // still format block n from its bytes while block n + 1 is being read and parsed.
    void keep_previous_block() { keep_previous_ = true; }
    static size_t default_block() {
        const char* env = getenv("UMGAP_BLOCK_BYTES");  // tests shrink it to exercise the block seams
        if (env && *env) { long v = atol(env); if (v >= 16) return (size_t)v; }
        return 64u << 20;
    }
    // a pipe or a socket: a stage starts on a block only when it holds all of it, so the four stages of the script's chain keep
    // eight blocks of stream between them before the first read comes out -- 32 MB blocks measured best there (4.6 against 4.1 M
    // pairs/s at 64 MB, 4.4 at 16); a mapped file has no such wait and keeps 64 MB
    static size_t piped_block(size_t block) {
        const char* env = getenv("UMGAP_BLOCK_BYTES");
        return (env && *env) ? block : std::min<size_t>(block, 32u << 20);
    }
    ~BlockReader() {
        if (map_) munmap(map_, map_len_);
        if (ahead_) { std::lock_guard<std::mutex> g(ahead_->m); ahead_->stop = true; ahead_->cv.notify_all(); }   // (stdin's filler is detached: see next_ahead)
        if (filler_.joinable()) {        // a connection's filler: woken out of its read() and waited for -- the descriptor is closed right after us
            if (conn_fd_ >= 0) (void)::shutdown(conn_fd_, SHUT_RD);
            filler_.join();
        }
    }
    // The stream is a connection of the socket server (not stdin): its blocks may be put together ahead of the caller as well.  The
    // filler then belongs to this reader -- it is stopped and joined when the reader goes, whatever became of the request.
    void connection() { connection_ = true; }
    BlockReader(const BlockReader&) = delete;
    BlockReader& operator=(const BlockReader&) = delete;
    // When the stream is a regular file (`umgap ... < reads.fa`; the command's stdin, wrapped or not, sits on descriptor 0) the
    // rest of the file is mapped and next() hands out views of the mapping: no read(), no copy, and the page faults are taken by
    // the worker threads that tokenize the block.  Call before the first next().  UMGAP_MMAP=0 keeps the reads (A/B).  (A file
    // that is truncated by somebody else while the command runs ends it with SIGBUS instead of a short read.)
    void map_regular_file() {
        const char* env = getenv("UMGAP_MMAP");
        if (env && env[0] == '0') return;
        const int fd = f_ == stdin ? 0 : fileno(f_);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return;
        const off_t at = lseek(fd, 0, SEEK_CUR);
        if (at < 0 || at >= st.st_size) return;                    // (an empty rest of file: the ordinary path says "exhausted")
        const long page = sysconf(_SC_PAGESIZE);
        const off_t base = at / page * page;
        void* m = mmap(nullptr, (size_t)(st.st_size - base), PROT_READ, MAP_PRIVATE, fd, base);
        if (m == MAP_FAILED) return;
        (void)madvise(m, (size_t)(st.st_size - base), MADV_SEQUENTIAL);
        map_ = m;
        map_len_ = (size_t)(st.st_size - base);
        mdata_ = (const char*)m + (at - base);
        mlen_ = (size_t)(st.st_size - at);
    }
    // Returns false when the stream is exhausted.  [data, data+len) holds whole records; the view
    // stays valid until the next call.
    bool next(const char*& data, size_t& len) {
        // the output is gone (closed pipe, full disk): stop here rather than work through the rest of the input (src/commands/*.rs:
        // every write_record(..)? ends the command with the io error)
        if (const int werr = AsyncIo::first_write_error()) throw CliError(std::string(strerror(werr)) + " (os error " + std::to_string(werr) + ")");
        if (map_) {
            if (mpos_ >= mlen_) return false;
            const char* b = mdata_ + mpos_;
            const size_t left = mlen_ - mpos_;
            if (first_) {
                if (!lines_only_ && b[0] != '>') first_line_error_ = true;   // src/io/fasta.rs:44-49
                first_ = false;
            }
            size_t take = left;
            for (size_t want = block_; want < left; want *= 2) {         // the last record start strictly inside the first `want` bytes
                size_t cut = 0;
                for (size_t p = want; p-- > 1;)
                    if (b[p - 1] == '\n' && (lines_only_ || b[p] == '>')) { cut = p; break; }
                if (cut) { take = cut; break; }
            }
            data = b;
            len = take;
            mpos_ += take;
            return true;
        }
        if (ahead_ok()) return next_ahead(data, len);
        started_sync_ = true;
        if (buf_.empty()) buf_.resize(piped_block(block_));   // (not before the first read: a mapped file needs no buffer)
        if (keep_previous_) {
            if (other_.size() < buf_.size()) other_.resize(buf_.size());
            if (carry_) memcpy(other_.data(), buf_.data() + carry_off_, carry_);
            buf_.swap(other_);
        } else if (carry_) { memmove(buf_.data(), buf_.data() + carry_off_, carry_); }
        size_t have = carry_;
        carry_ = 0;
        for (;;) {
            if (!eof_) {
                size_t got = fread(buf_.data() + have, 1, buf_.size() - have, f_);
                if (got == 0) {
                    if (ferror(f_)) throw CliError(std::string(strerror(errno)));
                    eof_ = true;
                }
                have += got;
            }
            if (have == 0) return false;
            if (first_) {
                // src/io/fasta.rs:44-49
                if (!lines_only_ && buf_[0] != '>') { first_line_error_ = true; }
                first_ = false;
            }
            if (eof_) { data = buf_.data(); len = have; return true; }
            // last record start strictly inside the buffer
            size_t cut = 0;
            for (size_t p = have; p-- > 1;)
                if (buf_[p - 1] == '\n' && (lines_only_ || buf_[p] == '>')) { cut = p; break; }
            if (cut > 0) {
                carry_off_ = cut;
                carry_ = have - cut;
                data = buf_.data();
                len = cut;
                return true;
            }
            buf_.resize(buf_.size() * 2);  // one record larger than the block: keep reading
        }
    }
    bool first_line_error() const { return first_line_error_; }
    // plain line-oriented input (TSV): blocks end at any newline, no FASTA header check
    void lines_only() { lines_only_ = true; }

  private:
    // ---- stdin that is a pipe: the blocks are put together AHEAD of the caller, by a thread of their own (late round 4).  next() used
    // to fread 32 MB out of the reader thread's 1 MB chunks itself: 0.35 s of copying per 8 M pairs in front of every stage's tokenizer
    // (seedextend over a pipe 0.99 s against 0.61 over a mapped file).  Four buffers take turns: the one the caller works on, the one
    // before it (keep_previous_block), and what the filler has ready or is filling.  Only for stdin: a command that fails leaves the
    // filler blocked in its read and exits; a server connection (which lives on after a failed request) keeps the synchronous path.
    struct Ahead {
        std::mutex m;
        std::condition_variable cv;
        struct Slot { std::vector<char> buf; size_t len = 0; };
        Slot slots[4];
        size_t produced = 0, consumed = 0, released = 0;     // block counters; block i lives in slot i % 4
        bool done = false, stop = false, first_line_error = false;
        std::exception_ptr err;
    };
    std::shared_ptr<Ahead> ahead_;
    bool ahead_ok() {
        if (ahead_) return true;
        if (started_sync_) return false;
        const char* e = getenv("UMGAP_BLOCK_AHEAD");
        if (e && e[0] == '0') return false;
        if (f_ == stdin) return true;
        if (!connection_ || !AsyncIo::enabled()) return false;
        conn_fd_ = AsyncIo::take_fd(f_);          // (nothing read from the FILE* yet: the filler reads the descriptor itself)
        if (conn_fd_ < 0) { started_sync_ = true; return false; }
        return true;
    }
    // the filler: the loop of next() below, one block after the other into the free slots
    static void fill_ahead(std::shared_ptr<Ahead> A, FILE* f, size_t block, bool lines_only, int taken_fd) {
        std::vector<char> tail;
        bool eof = false, first = true;
        // the descriptor itself where nobody has read from the FILE* yet: read() straight into the block (no pump thread, no chunks)
        const int fd = taken_fd >= 0 ? taken_fd : (AsyncIo::enabled() ? AsyncIo::take_fd(f) : -1);
        auto fill = [&](char* dst, size_t n) -> size_t {      // like fread: n bytes unless the stream ends (0 bytes: end or error)
            if (fd < 0) {
                const size_t got = fread(dst, 1, n, f);
                if (got == 0 && ferror(f)) throw CliError(std::string(strerror(errno)));
                return got;
            }
            size_t have = 0;
            while (have < n) {
                const ssize_t r = ::read(fd, dst + have, n - have);
                if (r < 0) { if (errno == EINTR) continue; throw CliError(std::string(strerror(errno))); }
                if (r == 0) break;
                have += (size_t)r;
            }
            return have;
        };
        try {
            for (size_t i = 0; !eof; i++) {
                Ahead::Slot* S;
                {
                    std::unique_lock<std::mutex> lk(A->m);
                    A->cv.wait(lk, [&] { return A->stop || i - A->released < 4; });
                    if (A->stop) return;
                    S = &A->slots[i % 4];
                }
                std::vector<char>& buf = S->buf;
                if (buf.size() < block) buf.resize(block);
                if (buf.size() < tail.size() + 1) buf.resize(tail.size() * 2);
                if (!tail.empty()) memcpy(buf.data(), tail.data(), tail.size());
                size_t have = tail.size(), len = 0;
                tail.clear();
                for (;;) {
                    if (!eof) {
                        const size_t want = buf.size() - have, got = fill(buf.data() + have, want);
                        if (got < want) eof = true;
                        have += got;
                    }
                    if (have == 0) break;                    // nothing left: no block
                    if (first) {
                        if (!lines_only && buf[0] != '>') { std::lock_guard<std::mutex> g(A->m); A->first_line_error = true; }
                        first = false;
                    }
                    if (eof) { len = have; break; }
                    size_t cut = 0;
                    for (size_t p = have; p-- > 1;)
                        if (buf[p - 1] == '\n' && (lines_only || buf[p] == '>')) { cut = p; break; }
                    if (cut > 0) { tail.assign(buf.data() + cut, buf.data() + have); len = cut; break; }
                    buf.resize(buf.size() * 2);              // one record larger than the block: keep reading
                }
                std::lock_guard<std::mutex> g(A->m);
                if (len) { S->len = len; A->produced = i + 1; }
                if (eof) A->done = true;
                A->cv.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> g(A->m);
            A->err = std::current_exception();
            A->done = true;
            A->cv.notify_all();
        }
    }
    bool next_ahead(const char*& data, size_t& len) {
        if (!ahead_) {
            ahead_ = std::make_shared<Ahead>();
            if (conn_fd_ >= 0) filler_ = std::thread(fill_ahead, ahead_, f_, piped_block(block_), lines_only_, conn_fd_);
            else std::thread(fill_ahead, ahead_, f_, piped_block(block_), lines_only_, -1).detach();
        }
        Ahead& A = *ahead_;
        std::unique_lock<std::mutex> lk(A.m);
        const size_t k = A.consumed;
        A.released = keep_previous_ ? (k ? k - 1 : 0) : k;       // the blocks before that are the filler's again
        A.cv.notify_all();
        A.cv.wait(lk, [&] { return A.produced > k || A.done; });
        if (A.produced > k) {
            first_line_error_ = A.first_line_error;
            data = A.slots[k % 4].buf.data();
            len = A.slots[k % 4].len;
            A.consumed = k + 1;
            return true;
        }
        if (A.err) { std::exception_ptr e = A.err; A.err = nullptr; std::rethrow_exception(e); }
        return false;
    }
    bool started_sync_ = false, connection_ = false;
    int conn_fd_ = -1;
    std::thread filler_;
    FILE* f_;
    size_t block_;
    std::vector<char> buf_, other_;
    size_t carry_ = 0, carry_off_ = 0;
    void* map_ = nullptr;            // map_regular_file(): the mapping, and the stream inside it
    size_t map_len_ = 0, mlen_ = 0, mpos_ = 0;
    const char* mdata_ = nullptr;
    bool eof_ = false, first_ = true, first_line_error_ = false, lines_only_ = false, keep_previous_ = false;
};

// Record-aligned split of a block into at most `parts` ranges [begin[i], begin[i+1]).
inline std::vector<size_t> split_records(const char* d, size_t len, unsigned parts) {
    std::vector<size_t> b{0};
    for (unsigned t = 1; t < parts; t++) {
        size_t p = len * t / parts;
        if (p <= b.back()) continue;
        // advance to the next record start
        const char* q = (const char*)memchr(d + p, '\n', len - p);
        while (q && (size_t)(q - d) + 1 < len && q[1] != '>') q = (const char*)memchr(q + 1, '\n', len - (size_t)(q + 1 - d));
        if (!q || (size_t)(q - d) + 1 >= len) break;
        size_t s = (size_t)(q - d) + 1;
        if (s > b.back()) b.push_back(s);
    }
    b.push_back(len);
    return b;
}

// Line-aligned split of a block into at most `parts` ranges.
inline std::vector<size_t> split_lines(const char* d, size_t len, unsigned parts) {
    std::vector<size_t> b{0};
    for (unsigned t = 1; t < parts; t++) {
        size_t p = len * t / parts;
        if (p <= b.back()) continue;
        const char* q = (const char*)memchr(d + p, '\n', len - p);
        if (!q || (size_t)(q - d) + 1 >= len) break;
        size_t s = (size_t)(q - d) + 1;
        if (s > b.back()) b.push_back(s);
    }
    b.push_back(len);
    return b;
}

// Calls fn(part) for part = 0..n-1, on the threads of a pool that lives as long as the process plus the caller (round 4: a fresh
// std::thread per part and call cost 0.3 ms per call -- four calls per 32 MB block -- and every call started with cold
// thread_local buffers).  Several threads may call it at the same time (the two stages of a command do): the pool serves their
// jobs in turn, each caller works on its own job while it waits.  The first exception of the lowest part is rethrown.
class WorkerPool {
public:
    static WorkerPool& get() {
        static WorkerPool* pool = new WorkerPool(worker_count());     // never destroyed: its idle threads end with the process
        return *pool;
    }
    void run(size_t n, const std::function<void(size_t)>& fn) {
        std::vector<std::exception_ptr> err(n);
        Job job{&fn, &err, n, 0, 0};
        {
            std::unique_lock<std::mutex> lk(m_);
            jobs_.push_back(&job);
            cv_work_.notify_all();
            for (;;) {
                if (job.next < job.n) {                // the caller takes parts of its own job
                    const size_t i = job.next++;
                    if (job.next == job.n) drop(&job);
                    lk.unlock();
                    call(job, i);
                    lk.lock();
                    job.done++;
                } else if (job.done < job.n) cv_done_.wait(lk);
                else break;
            }
        }
        for (auto& e : err)
            if (e) std::rethrow_exception(e);
    }

private:
    struct Job {
        const std::function<void(size_t)>* fn;
        std::vector<std::exception_ptr>* err;
        size_t n, next, done;                          // guarded by m_
    };
    explicit WorkerPool(unsigned threads) {
        for (unsigned t = 0; t < threads; t++) std::thread([this]() { work(); }).detach();
    }
    static void call(Job& j, size_t i) {
        try { (*j.fn)(i); } catch (...) { (*j.err)[i] = std::current_exception(); }
    }
    void drop(Job* j) {                                // m_ held
        for (size_t k = 0; k < jobs_.size(); k++)
            if (jobs_[k] == j) { jobs_.erase(jobs_.begin() + (long)k); return; }
    }
    void work() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            while (jobs_.empty()) cv_work_.wait(lk);
            Job* j = jobs_.front();
            const size_t i = j->next++;
            if (j->next == j->n) drop(j);
            lk.unlock();
            call(*j, i);
            lk.lock();
            if (++j->done == j->n) cv_done_.notify_all();      // (the job lives on its caller's stack until the caller has seen this, under m_)
        }
    }
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Job*> jobs_;                           // jobs that still have parts to hand out
};
inline void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    if (n == 1) { fn(0); return; }
    WorkerPool::get().run(n, fn);
}

// Line iterator over a range of whole records.  `line` excludes '\n' and one trailing '\r'.
struct LineCursor {
    const char* p;
    const char* end;
    bool next(std::string_view& line) {
        if (p >= end) return false;
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        const char* le = e;
        if (le > p && le[-1] == '\r') le--;
        line = std::string_view(p, (size_t)(le - p));
        p = nl ? nl + 1 : end;
        return true;
    }
};

inline bool line_utf8_ok(std::string_view l);

// read_taxa_file (text.hpp) with the lines parsed on worker threads: 2.5 M NCBI rows = 100 MB take 1 s on one thread,
// which every taxa2agg / prot2kmer2agg invocation pays before its first record.  The first problem in file order wins,
// as in the sequential loop of src/taxon.rs:100-128.
inline TaxonRows read_taxa_file_parallel(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        CliError e("Failed opening taxon file.");
        e.causes.push_back(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");
        throw e;
    }
    std::vector<char> data;
    {
        std::vector<char> chunk(8u << 20);
        size_t got;
        while ((got = fread(chunk.data(), 1, chunk.size(), f)) > 0) data.insert(data.end(), chunk.begin(), chunk.begin() + got);
        bool bad = ferror(f);
        fclose(f);
        if (bad) throw CliError("Failed to read all lines.");
    }
    struct Part { TaxonRows rows; std::optional<CliError> error; };
    std::vector<size_t> cuts = split_lines(data.data(), data.size(), worker_count());
    std::vector<Part> parts(cuts.size() - 1);
    parallel_for(parts.size(), [&](size_t i) {
        LineCursor lc{data.data() + cuts[i], data.data() + cuts[i + 1]};
        std::string_view l;
        while (lc.next(l)) {
            if (!line_utf8_ok(l)) { parts[i].error = CliError("Failed to read all lines."); return; }
            try { parse_taxon_line(l, parts[i].rows); } catch (CliError& e) { parts[i].error = e; return; }
        }
    });
    TaxonRows rows;
    for (auto& P : parts) {
        rows.id.insert(rows.id.end(), P.rows.id.begin(), P.rows.id.end());
        rows.parent.insert(rows.parent.end(), P.rows.parent.begin(), P.rows.parent.end());
        rows.valid.insert(rows.valid.end(), P.rows.valid.begin(), P.rows.valid.end());
        rows.rank.insert(rows.rank.end(), P.rows.rank.begin(), P.rows.rank.end());
        if (P.error) throw *P.error;
    }
    return rows;
}

inline bool line_utf8_ok(std::string_view l) {
    for (unsigned char c : l)
        if (c >= 0x80) return valid_utf8(l);
    return true;
}

// decimal digits of v, two at a time from a table, appended in one go
// decimal digits of v at p (room for 20); returns the end
inline char* write_uint(char* p, uint64_t v) {
    static const char pairs[] =
        "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
        "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    char tmp[24];
    char* q = tmp + sizeof(tmp);
    while (v >= 100) {
        const unsigned r = (unsigned)(v % 100);
        v /= 100;
        q -= 2;
        q[0] = pairs[2 * r];
        q[1] = pairs[2 * r + 1];
    }
    if (v >= 10) { q -= 2; q[0] = pairs[2 * v]; q[1] = pairs[2 * v + 1]; }
    else *--q = (char)('0' + v);
    const size_t n = (size_t)(tmp + sizeof(tmp) - q);
    memcpy(p, q, n);
    return p + n;
}
inline void append_uint(std::string& s, uint64_t v) {
    static const char pairs[] =
        "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
        "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    char tmp[24];
    char* p = tmp + sizeof(tmp);
    while (v >= 100) {
        const unsigned r = (unsigned)(v % 100);
        v /= 100;
        p -= 2;
        p[0] = pairs[2 * r];
        p[1] = pairs[2 * r + 1];
    }
    if (v >= 10) { p -= 2; p[0] = pairs[2 * v]; p[1] = pairs[2 * v + 1]; }
    else *--p = (char)('0' + v);
    s.append(p, (size_t)(tmp + sizeof(tmp) - p));
}

}  // namespace umgap_host
