// fuse_parse.hpp — the tokenizer of `umgap prot2kmer2agg`: a record-aligned range of protein FASTA -> the fragments
// `prot2kmer2lca` would look up (src/commands/prot2kmer2lca.rs:172-179) and the headers `uniq` groups them by
// (src/commands/uniq.rs:66-79).  Host side, no GPU.
//
// One pass over the text, 64 bytes at a time: a newline mask and a high-bit mask per 64-byte stretch (SSE2 compare +
// movemask; every x86-64 has it), the lines read off the masks, residues copied once into the part's buffer.  The first
// version walked the range line by line (memchr per line, a byte loop per line for the UTF-8 test, a second byte loop per
// record for the ASCII test, vector growth per line): 315 MB/s per thread, which set the pace of the whole command.
//
// Semantics (src/io/fasta.rs:38-67 with unwrap = true, as the reference's reader gives them): lines end at '\n', ONE trailing
// '\r' is stripped, a last line without '\n' counts; a non-empty line starting with '>' opens a record, every other line is
// appended to the open record; a line that is not valid UTF-8 ends the range (bad_utf8) with the open record dropped; a record
// of fewer than 9 bytes is dropped, header and all (prot2kmer2lca.rs:172); a record in which a 9-byte window edge falls
// inside a multi-byte character is the reference's slicing panic (:175).
#pragma once
#include <emmintrin.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

#include "text.hpp"

namespace umgap_host {

// The residue buffer of a part: plain memory unless the host installs an allocator -- `umgap prot2kmer2agg` installs the library's
// page-locked one and hands the buffers to the device call as they lie (umgap_classify_records_sg).  Contents are not kept over a
// growth.
struct PartBytes {
    uint8_t* p = nullptr;
    bool plain = true;                     // p came from new[] (no allocator installed, or the allocator had nothing left)
    void* (*alloc)(size_t) = nullptr;      // both null: new[] / delete[]
    void (*release)(void*) = nullptr;
    PartBytes() = default;
    PartBytes(const PartBytes&) = delete;
    PartBytes& operator=(const PartBytes&) = delete;
    PartBytes(PartBytes&& o) noexcept : p(o.p), plain(o.plain), alloc(o.alloc), release(o.release) { o.p = nullptr; }
    PartBytes& operator=(PartBytes&& o) noexcept {
        if (this != &o) { drop(); p = o.p; plain = o.plain; alloc = o.alloc; release = o.release; o.p = nullptr; }
        return *this;
    }
    ~PartBytes() { drop(); }
    void drop() {
        if (p) { if (!plain && release) release(p); else delete[] p; }
        p = nullptr;
    }
    // (a page-locked allocation that fails falls back to plain memory: the device call then gathers the piece through its staging)
    void reset_to(size_t bytes) {
        drop();
        p = alloc ? (uint8_t*)alloc(bytes) : nullptr;
        plain = p == nullptr;
        if (!p) p = new uint8_t[bytes];
    }
    uint8_t* get() const { return p; }
};

// (aligned to two cache lines, and the tokenizer works on its own copies of the vectors: the parts of a block sit side by side in
// one array and are written by different threads -- three push_backs per record into neighbouring structs had the threads fight
// over the lines that hold the vectors' end pointers: 2 threads tokenized SLOWER than 1, 16 at a quarter of the single-thread rate each)
struct alignas(128) FusePart {
    // per emitted record (sequence >= 9 residues): its header as `uniq` sees it (two more CRs stripped by the readers of
    // seedextend and uniq, cut at the delimiter) and its sequence length
    std::vector<size_t> hdr_off;
    std::vector<uint32_t> hdr_len;
    std::vector<uint64_t> frag_len;
    std::vector<uint32_t> gstart;   // records that open a new `uniq` group inside this range (record 0 always does)
    PartBytes bytes;                    // the residues of the emitted records, back to back
    size_t n_bytes = 0, cap_bytes = 0;  // (the buffer and the vectors keep their memory when a part is parsed into again: no fresh pages per block)
    bool bad_utf8 = false, panic = false;
    size_t rec_base = 0, byte_base = 0, group_base = 0, hdr_base = 0, hdr_bytes = 0;
    bool merged = false;            // record 0 continues the group that was open before this range
    std::string_view header(const char* d, size_t r) const { return std::string_view(d + hdr_off[r], hdr_len[r]); }
};

namespace fuse_detail {
struct Masks { uint64_t nl, hi; };
inline Masks scan64(const char* p) {
    const __m128i n = _mm_set1_epi8('\n');
    Masks m{0, 0};
    for (int i = 0; i < 4; i++) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16 * i));
        m.nl |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(v, n)) << (16 * i);
        m.hi |= (uint64_t)(uint32_t)_mm_movemask_epi8(v) << (16 * i);
    }
    return m;
}
inline uint64_t below(unsigned k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }   // bits [0, k)
}  // namespace fuse_detail

// The lines of d[begin, end), handed to line(s, e, high) in order: [s, e) is the line without its '\n' (a trailing '\r' is the
// callee's business), high = some byte of it has bit 7 set (only then can it be anything but ASCII).  A last line without '\n'
// counts (BufRead::lines); `stop` set by the callee ends the scan.
template <class LineFn>
inline void scan_lines(const char* d, size_t begin, size_t end, bool& stop, LineFn&& line) {
    using namespace fuse_detail;
    size_t pos = begin, line_start = begin;
    bool line_high = false;   // the current line has a high byte in a stretch already scanned
    while (!stop && pos + 64 <= end) {
        const Masks m = scan64(d + pos);
        uint64_t nl = m.nl, done = 0;   // done: bits of this stretch that belong to lines already handed out
        while (nl) {
            const unsigned k = (unsigned)__builtin_ctzll(nl);
            nl &= nl - 1;
            const bool high = line_high || (m.hi & below(k) & ~done) != 0;
            line(line_start, pos + k, high);
            if (stop) break;
            line_start = pos + k + 1;
            done = below(k + 1);
            line_high = false;
        }
        line_high = line_high || (m.hi & ~done) != 0;
        pos += 64;
    }
    while (!stop && pos < end) {   // the last < 64 bytes
        const unsigned char c = (unsigned char)d[pos];
        if (c == '\n') { line(line_start, pos, line_high); line_start = pos + 1; line_high = false; }
        else if (c >= 0x80) line_high = true;
        pos++;
    }
    if (!stop && line_start < end) line(line_start, end, line_high);   // a last line without '\n'
}

inline void parse_fuse_part(const char* d, size_t begin, size_t end, const std::optional<std::string>& delim, FusePart& P) {
    using namespace fuse_detail;
    const size_t len = end - begin;
    if (P.cap_bytes < len + 16) { P.cap_bytes = len + len / 8 + 4096; P.bytes.reset_to(P.cap_bytes); }
    uint8_t* out = P.bytes.get();
    size_t out_n = 0;
    P.n_bytes = 0; P.bad_utf8 = P.panic = P.merged = false;
    P.rec_base = P.byte_base = P.group_base = P.hdr_base = P.hdr_bytes = 0;
    // the vectors live on this thread's stack while it fills them (their memory is the part's from the last block)
    std::vector<size_t> hdr_off = std::move(P.hdr_off);
    std::vector<uint32_t> hdr_len = std::move(P.hdr_len), gstart = std::move(P.gstart);
    std::vector<uint64_t> frag_len = std::move(P.frag_len);
    hdr_off.clear(); hdr_len.clear(); frag_len.clear(); gstart.clear();
    {
        const size_t guess = len / 96 + 16;   // records of two ~50-residue lines and a short header
        hdr_off.reserve(guess); hdr_len.reserve(guess); frag_len.reserve(guess); gstart.reserve(guess / 2 + 8);
    }
    size_t hdr_bytes = 0;
    bool panic = false, bad_utf8 = false;
    bool open = false, rec_high = false, stop = false;
    size_t rec_start = 0;
    std::string_view cur;

    auto finish = [&]() {
        if (!open) return;
        open = false;
        const size_t n = out_n - rec_start;
        if (n < 9) { out_n = rec_start; return; }   // src/commands/prot2kmer2lca.rs:172: dropped, header and all
        if (rec_high) {  // &seq[i..i + k] panics when a window edge falls inside a character (:175)
            const uint8_t* q = out + rec_start;
            for (size_t i = 0; i + 9 <= n; i++)
                if ((q[i] & 0xC0) == 0x80 || (i + 9 < n && (q[i + 9] & 0xC0) == 0x80)) { panic = true; out_n = rec_start; return; }
        }
        std::string_view h = cur;
        // every reader of the pipe strips one trailing CR from a line (BufRead::lines): seedextend's and uniq's here
        for (int k = 0; k < 2 && !h.empty() && h.back() == '\r'; k++) h.remove_suffix(1);
        if (delim) {
            if (delim->size() == 1) { const void* at = memchr(h.data(), (*delim)[0], h.size()); if (at) h = h.substr(0, (size_t)((const char*)at - h.data())); }
            else { const size_t at = h.find(*delim); if (at != std::string_view::npos) h = h.substr(0, at); }
        }
        const size_t r = hdr_off.size();
        if (r == 0 || h != std::string_view(d + hdr_off[r - 1], hdr_len[r - 1])) { gstart.push_back((uint32_t)r); hdr_bytes += h.size(); }   // src/commands/uniq.rs:70-77
        hdr_off.push_back((size_t)(h.data() - d));
        hdr_len.push_back((uint32_t)h.size());
        frag_len.push_back(n);
    };
    // one line [s, e) of the text (e = its '\n' or the end of the range); high: some byte of it has bit 7 set
    auto line = [&](size_t s, size_t e, bool high) {
        size_t le = e;
        if (le > s && d[le - 1] == '\r') le--;
        const bool utf8 = !high || valid_utf8(std::string_view(d + s, le - s));
        if (le > s && d[s] == '>' && utf8) {
            finish();
            if (panic) { stop = true; return; }
            cur = std::string_view(d + s + 1, le - s - 1);
            rec_start = out_n;
            rec_high = false;
            open = true;
            return;
        }
        if (!utf8) {
            bad_utf8 = true;
            if (open) out_n = rec_start;
            open = false;
            stop = true;
            return;
        }
        memcpy(out + out_n, d + s, le - s);   // unwrap: lines of a record are concatenated (src/io/fasta.rs:62-64)
        out_n += le - s;
        rec_high = rec_high || high;
    };

    scan_lines(d, begin, end, stop, line);
    if (!stop) finish();
    P.n_bytes = out_n;
    P.hdr_bytes = hdr_bytes;
    P.panic = panic;
    P.bad_utf8 = bad_utf8;
    P.hdr_off = std::move(hdr_off); P.hdr_len = std::move(hdr_len); P.frag_len = std::move(frag_len); P.gstart = std::move(gstart);
}

// ---- the records of `umgap prot2kmer2lca` (k = 9, a 9-mer index): the tokenizer behind umgap_index_lookup_records ----------------
// Per record of the range, in input order: its header as printed (the line behind '>', one CR stripped by the reader), whether it
// is printed at all (a sequence shorter than 9 is dropped, header and all: src/commands/prot2kmer2lca.rs:172 -- but still counts as
// a record of its chunk, src/io/fasta.rs:118-137) and the length of its sequence in the part's byte buffer (0 when dropped).
// Invalid UTF-8: the record read so far is emitted, then the stream errors out (bad_utf8); a window edge inside a multi-byte
// character: the reference's slicing panic (:175), the range ends with that record (panic; the record is listed, not printed).
struct alignas(128) KmerPart {
    std::vector<size_t> hdr_off;
    std::vector<uint32_t> hdr_len;
    std::vector<uint64_t> seq_len;      // residues of the record in `bytes` (0: not printed)
    std::vector<uint8_t> emit;
    std::unique_ptr<uint8_t[]> bytes;
    size_t n_bytes = 0, cap_bytes = 0;
    bool bad_utf8 = false, panic = false;
    size_t rec_base = 0, byte_base = 0;
    std::string out;
};

inline void parse_kmer_part(const char* d, size_t begin, size_t end, KmerPart& P) {
    const size_t len = end - begin;
    if (P.cap_bytes < len + 16) { P.cap_bytes = len + len / 8 + 4096; P.bytes.reset(new uint8_t[P.cap_bytes]); }
    uint8_t* out = P.bytes.get();
    size_t out_n = 0;
    P.n_bytes = 0; P.bad_utf8 = P.panic = false;
    P.rec_base = P.byte_base = 0;
    std::vector<size_t> hdr_off = std::move(P.hdr_off);
    std::vector<uint32_t> hdr_len = std::move(P.hdr_len);
    std::vector<uint64_t> seq_len = std::move(P.seq_len);
    std::vector<uint8_t> emit = std::move(P.emit);
    hdr_off.clear(); hdr_len.clear(); seq_len.clear(); emit.clear();
    {
        const size_t guess = len / 64 + 16;
        hdr_off.reserve(guess); hdr_len.reserve(guess); seq_len.reserve(guess); emit.reserve(guess);
    }
    bool have_rec = false, rec_high = false, stop = false, panic = false, bad_utf8 = false;
    size_t rec_start = 0;
    auto finish = [&]() {
        if (!have_rec) return;
        const size_t n = out_n - rec_start;
        if (n < 9) { out_n = rec_start; emit.back() = 0; seq_len.back() = 0; return; }
        if (rec_high) {
            const uint8_t* q = out + rec_start;
            for (size_t i = 0; i + 9 <= n; i++)
                if ((q[i] & 0xC0) == 0x80 || (i + 9 < n && (q[i + 9] & 0xC0) == 0x80)) { panic = true; out_n = rec_start; emit.back() = 0; seq_len.back() = 0; return; }
        }
        seq_len.back() = n;
    };
    auto line = [&](size_t s, size_t e, bool high) {
        size_t le = e;
        if (le > s && d[le - 1] == '\r') le--;
        const bool utf8 = !high || valid_utf8(std::string_view(d + s, le - s));
        if (le > s && d[s] == '>' && utf8) {
            finish();
            if (panic) { stop = true; return; }
            hdr_off.push_back(s + 1);
            hdr_len.push_back((uint32_t)(le - s - 1));
            seq_len.push_back(0);
            emit.push_back(1);
            have_rec = true;
            rec_start = out_n;
            rec_high = false;
            return;
        }
        if (!utf8) { bad_utf8 = true; stop = true; return; }   // the record read so far is emitted below, then the stream errors out
        if (!have_rec) return;
        memcpy(out + out_n, d + s, le - s);
        out_n += le - s;
        rec_high = rec_high || high;
    };
    scan_lines(d, begin, end, stop, line);
    if (!panic) finish();
    P.n_bytes = out_n;
    P.panic = panic;
    P.bad_utf8 = bad_utf8;
    P.hdr_off = std::move(hdr_off); P.hdr_len = std::move(hdr_len); P.seq_len = std::move(seq_len); P.emit = std::move(emit);
}

}  // namespace umgap_host
