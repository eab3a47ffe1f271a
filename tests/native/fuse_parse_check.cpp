// fuse_parse_check.cpp — differential check of the block tokenizer of `umgap prot2kmer2agg` (umgap_amd/csrc/host/fuse_parse.hpp:
// 64 bytes at a time off newline / high-bit masks) against a plain line-by-line restatement of the same rules, on random
// FASTA-like byte soup.  Test infrastructure (tests/test_cli_text_cpu.py builds and runs it; no GPU).
//   fuse_parse_check <cases> <seed>     exit 0 = every case agrees
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../umgap_amd/csrc/host/fuse_parse.hpp"

using namespace umgap_host;

struct Ref {
    std::vector<std::string> hdr;
    std::vector<std::string> seq;
    std::vector<uint32_t> gstart;
    bool bad_utf8 = false, panic = false;
};

// src/io/fasta.rs:38-67 (unwrap) + src/commands/prot2kmer2lca.rs:172-179 + the header as uniq sees it, one line at a time
static Ref reference(const std::string& text, const std::optional<std::string>& delim) {
    Ref R;
    std::vector<std::string> lines;
    size_t p = 0;
    while (p < text.size()) {
        size_t nl = text.find('\n', p);
        size_t e = nl == std::string::npos ? text.size() : nl;
        std::string l = text.substr(p, e - p);
        if (!l.empty() && l.back() == '\r') l.pop_back();
        lines.push_back(l);
        p = nl == std::string::npos ? text.size() : nl + 1;
    }
    bool open = false;
    std::string cur, seq;
    auto finish = [&]() {
        if (!open) return;
        open = false;
        if (seq.size() < 9) return;
        bool ascii = true;
        for (unsigned char c : seq) if (c >= 0x80) ascii = false;
        if (!ascii)
            for (size_t i = 0; i + 9 <= seq.size(); i++)
                if (((unsigned char)seq[i] & 0xC0) == 0x80 || (i + 9 < seq.size() && ((unsigned char)seq[i + 9] & 0xC0) == 0x80)) { R.panic = true; return; }
        std::string h = cur;
        for (int k = 0; k < 2 && !h.empty() && h.back() == '\r'; k++) h.pop_back();
        if (delim) { size_t at = h.find(*delim); if (at != std::string::npos) h = h.substr(0, at); }
        if (R.hdr.empty() || h != R.hdr.back()) R.gstart.push_back((uint32_t)R.hdr.size());
        R.hdr.push_back(h);
        R.seq.push_back(seq);
    };
    for (const std::string& l : lines) {
        const bool utf8 = valid_utf8(l);
        if (!l.empty() && l[0] == '>' && utf8) {
            finish();
            if (R.panic) return R;
            cur = l.substr(1);
            seq.clear();
            open = true;
            continue;
        }
        if (!utf8) { R.bad_utf8 = true; return R; }
        seq += l;
    }
    finish();
    return R;
}

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)atoll(argv[2]) : 1);
    const char* alphabet = "ACDEFGHIKLMNPQRSTVWY";
    for (long c = 0; c < cases; c++) {
        // a range of whole records: starts with a header line; lines of every length around the 64-byte stretches, CRs, empty lines,
        // high bytes (valid and broken UTF-8), '>' inside lines, a missing last newline
        std::string text;
        const int n_rec = 1 + (int)(rng() % 12);
        const int flavour = (int)(rng() % 8);
        for (int r = 0; r < n_rec; r++) {
            text += ">";
            const int hl = (int)(rng() % 20);
            for (int i = 0; i < hl; i++) text += (rng() % 9 == 0) ? '/' : (char)('a' + rng() % 4);
            if (rng() % 6 == 0) text += "\r";
            if (rng() % 9 == 0) text += "\r\r";
            if (flavour == 1 && rng() % 5 == 0) text += "\xC3\xA9";
            text += "\n";
            const int n_lines = (int)(rng() % 4);
            for (int k = 0; k < n_lines; k++) {
                static const int lens[] = {0, 1, 5, 8, 9, 10, 50, 62, 63, 64, 65, 127, 128, 129, 200};
                const int len = lens[rng() % 15];
                for (int i = 0; i < len; i++) {
                    char ch = alphabet[rng() % 20];
                    if (flavour == 2 && rng() % 40 == 0) { text += "\xE2\x82\xAC"; continue; }     // a valid 3-byte character
                    if (flavour == 3 && rng() % 200 == 0) ch = (char)(0x80 + rng() % 0x80);         // a stray high byte
                    if (rng() % 300 == 0) ch = '>';
                    if (rng() % 300 == 0) ch = '*';
                    text += ch;
                }
                if (rng() % 5 == 0) text += "\r";
                text += "\n";
            }
        }
        if (rng() % 3 == 0 && !text.empty()) text.pop_back();                                       // no final newline
        std::optional<std::string> delim;
        if (rng() % 2) delim = "/";
        // the range sits inside a larger buffer at an odd offset (unaligned 16-byte loads, views into `d`)
        const size_t pre = rng() % 70;
        std::string buf = std::string(pre, 'Z') + text + std::string(80, 'Z');
        FusePart P;
        parse_fuse_part(buf.data(), pre, pre + text.size(), delim, P);
        const Ref R = reference(text, delim);
        bool ok = P.bad_utf8 == R.bad_utf8 && P.panic == R.panic && P.hdr_off.size() == R.hdr.size() && P.gstart == R.gstart;
        size_t at = 0, hb = 0;
        for (size_t r = 0; ok && r < R.hdr.size(); r++) {
            ok = ok && std::string(P.header(buf.data(), r)) == R.hdr[r] && P.frag_len[r] == R.seq[r].size() && at + R.seq[r].size() <= P.n_bytes &&
                 memcmp(P.bytes.get() + at, R.seq[r].data(), R.seq[r].size()) == 0;
            at += R.seq[r].size();
        }
        for (uint32_t g : R.gstart) hb += R.hdr[g].size();
        ok = ok && at == P.n_bytes && hb == P.hdr_bytes;
        if (!ok) {
            fprintf(stderr, "case %ld differs: records %zu vs %zu, bad_utf8 %d vs %d, panic %d vs %d, bytes %zu vs %zu\n", c, P.hdr_off.size(), R.hdr.size(), (int)P.bad_utf8,
                    (int)R.bad_utf8, (int)P.panic, (int)R.panic, P.n_bytes, at);
            return 1;
        }
    }
    printf("%ld cases agree\n", cases);
    return 0;
}
