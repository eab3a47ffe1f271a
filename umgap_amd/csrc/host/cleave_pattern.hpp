// The cleavage pattern of `umgap prot2tryp2lca -p` (src/commands/prot2tryp2lca.rs:65-68,96,112-115).
//
// The reference compiles the pattern with Rust's `regex` crate (Cargo.toml: regex = "1.3.1") and runs
// replace_all(pattern, "$1\n$2") twice.  The host side here runs std::regex (ECMAScript grammar), so the pattern is first
// rewritten from the crate's syntax into that grammar.  Read the same way in both: literals, `.`, classes and ranges,
// POSIX classes ([[:alpha:]]), \d \w \s \b and their negations, groups, (?:...), alternation, greedy and lazy quantifiers,
// {m,n}, ^ and $.  Rewritten: a leading flag group (?i) / (?m) / (?im), named groups (?P<n>...) and (?<n>...) (numbered
// all the same; the replacement only uses $1 and $2), \A and \z, \x{41}, a leading ] in a class.  Refused with the
// "regex parse error" the reference would print for them: look-around and backreferences (the crate has neither).
// Refused because std::regex cannot express them (documented as the difference in DESIGN.md section 8): Unicode classes
// \p{..}, class set operations ([a-z&&[^aeiou]]), the flags s x U u and flag groups anywhere but at the start.
#pragma once
#include <pthread.h>

#include <algorithm>
#include <cctype>
#include <cstring>
#include <exception>
#include <regex>
#include <stdexcept>
#include <string>

namespace umgap_host {

struct CleavePatternError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct CleavePattern {
    std::regex re;
    // replace_all(seq, "$1\n$2") run twice: a residue can end one match and start the next
    std::string cut(const std::string& seq) const {
        // libstdc++'s matcher recurses once per character on a greedy pattern (`(.*)([KR])`): a 30 000-residue protein would run through
        // the 8 MB stack of a worker thread.  Long sequences are cut on a thread of their own whose stack is sized for them.
        if (seq.size() <= 4096) return cut_here(seq);
        struct Job { const CleavePattern* self; const std::string* seq; std::string out; std::exception_ptr err; } job{this, &seq, {}, nullptr};
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        // (address space: only the pages used are ever touched.  The matcher nests about five frames per residue -- ~1 KB, several times
        // that in an instrumented build)
        pthread_attr_setstacksize(&attr, std::min<size_t>((size_t)(64u << 20) + seq.size() * 8192, (size_t)8 << 30));
        pthread_t th;
        const int rc = pthread_create(&th, &attr, [](void* p) -> void* {
            Job* j = static_cast<Job*>(p);
            try { j->out = j->self->cut_here(*j->seq); } catch (...) { j->err = std::current_exception(); }
            return nullptr;
        }, &job);
        pthread_attr_destroy(&attr);
        if (rc != 0) throw CleavePatternError("the sequence is too long for the custom cleavage pattern (" + std::to_string(seq.size()) + " residues)");
        pthread_join(th, nullptr);
        if (job.err) std::rethrow_exception(job.err);
        return std::move(job.out);
    }
    std::string cut_here(const std::string& seq) const {
        return std::regex_replace(std::regex_replace(seq, re, "$1\n$2"), re, "$1\n$2");
    }
};

// the crate's syntax -> ECMAScript source + flags; throws CleavePatternError
inline std::string rewrite_cleave_pattern(const std::string& src, std::regex::flag_type& flags) {
    auto fail = [&](const std::string& why) -> std::string { throw CleavePatternError("regex parse error:\n    " + src + "\nerror: " + why); };
    size_t i = 0;
    flags = std::regex::ECMAScript | std::regex::optimize;
    bool multiline = false;
    // a leading (?flags) group
    if (src.compare(0, 2, "(?") == 0) {
        size_t j = 2;
        while (j < src.size() && (std::isalpha((unsigned char)src[j]) || src[j] == '-')) j++;
        if (j < src.size() && src[j] == ')' && j > 2) {
            for (size_t f = 2; f < j; f++) {
                if (src[f] == 'i') flags |= std::regex::icase;
                else if (src[f] == 'm') { flags |= std::regex::multiline; multiline = true; }
                else return fail(std::string("flag '") + src[f] + "' is not supported by this build (only i and m are)");
            }
            i = j + 1;
        }
    }
    std::string out;
    bool in_class = false;
    for (; i < src.size(); i++) {
        char c = src[i];
        if (c == '\\') {
            if (i + 1 >= src.size()) return fail("incomplete escape sequence, reached end of pattern prematurely");
            char e = src[++i];
            if (e >= '1' && e <= '9') return fail("backreferences are not supported");
            if (e == 'p' || e == 'P') return fail("Unicode classes (\\p) are not supported by this build");
            if (!in_class && e == 'A') { if (multiline) return fail("\\A under (?m) is not supported by this build"); out += '^'; continue; }
            if (!in_class && e == 'z') { if (multiline) return fail("\\z under (?m) is not supported by this build"); out += '$'; continue; }
            if (e == 'x' && i + 1 < src.size() && src[i + 1] == '{') {
                size_t close = src.find('}', i + 2);
                if (close == std::string::npos) return fail("invalid hexadecimal digit");
                unsigned long v = 0;
                try { size_t used = 0; v = std::stoul(src.substr(i + 2, close - i - 2), &used, 16); if (used != close - i - 2) throw 0; }
                catch (...) { return fail("invalid hexadecimal digit"); }
                if (v > 0x7F) return fail("code points above U+007F are not supported by this build");
                static const char* hex = "0123456789ABCDEF";
                out += "\\x"; out += hex[v >> 4]; out += hex[v & 15];
                i = close;
                continue;
            }
            // \u{41} \U{41} \u0041 \U00000041: the crate's other spellings of a code point
            if (e == 'u' || e == 'U') {
                std::string digits;
                size_t end = i;
                if (i + 1 < src.size() && src[i + 1] == '{') {
                    const size_t close = src.find('}', i + 2);
                    if (close == std::string::npos) return fail("invalid hexadecimal digit");
                    digits = src.substr(i + 2, close - i - 2);
                    end = close;
                } else {
                    const size_t nd = e == 'u' ? 4 : 8;
                    if (i + nd >= src.size()) return fail("incomplete escape sequence, reached end of pattern prematurely");
                    digits = src.substr(i + 1, nd);
                    end = i + nd;
                }
                unsigned long v = 0;
                try { size_t used = 0; v = std::stoul(digits, &used, 16); if (used != digits.size() || digits.empty()) throw 0; }
                catch (...) { return fail("invalid hexadecimal digit"); }
                if (v > 0x7F) return fail("code points above U+007F are not supported by this build");
                static const char* hex = "0123456789ABCDEF";
                out += "\\x"; out += hex[v >> 4]; out += hex[v & 15];
                i = end;
                continue;
            }
            if (e == '0') return fail("backreferences are not supported");           // (octal is off in the crate: \0 is read as a backreference)
            if (e == 'a') { out += "\\x07"; continue; }                             // bell; ECMAScript would read a literal `a`
            // what both grammars read the same way: the Perl classes, word boundaries, control escapes, \xHH, and the crate's escapable
            // punctuation.  Anything else the crate refuses (`\e`, `\q`, ...) or means something std::regex does not know: refused
            // rather than misread.
            if (std::strchr("dDwWsSbBntrfvx", e) || std::strchr("\\.+*?()|[]{}^$#&-~", e)) { out += '\\'; out += e; continue; }
            return fail(std::string("unrecognized escape sequence \\") + e);
        }
        if (in_class) {
            if (c == '&' && i + 1 < src.size() && src[i + 1] == '&') return fail("class set operations (&&) are not supported by this build");
            if (c == '[' && i + 1 < src.size() && src[i + 1] == ':') {     // [:alpha:] stays
                size_t close = src.find(":]", i + 2);
                if (close == std::string::npos) return fail("unclosed character class");
                out.append(src, i, close + 2 - i);
                i = close + 1;
                continue;
            }
            if (c == '[') return fail("nested character classes are not supported by this build");
            if (c == ']') in_class = false;
            out += c;
            continue;
        }
        if (c == '[') {
            in_class = true;
            out += c;
            if (i + 1 < src.size() && src[i + 1] == '^') { out += '^'; i++; }
            if (i + 1 < src.size() && src[i + 1] == ']') { out += "\\]"; i++; }   // a leading ] is a literal in the crate's syntax
            continue;
        }
        if (c == '(' && i + 1 < src.size() && src[i + 1] == '?') {
            const std::string rest = src.substr(i + 2, 3);
            if (rest.compare(0, 1, ":") == 0) { out += "(?:"; i += 2; continue; }
            if (rest.compare(0, 1, "=") == 0 || rest.compare(0, 1, "!") == 0 || rest.compare(0, 2, "<=") == 0 || rest.compare(0, 2, "<!") == 0)
                return fail("look-around, including look-ahead and look-behind, is not supported");
            if (rest.compare(0, 2, "P<") == 0 || rest.compare(0, 1, "<") == 0) {
                size_t close = src.find('>', i + 2);
                if (close == std::string::npos || close == i + (rest[0] == 'P' ? 4 : 3)) return fail("empty or unclosed capture group name");
                out += '(';
                i = close;
                continue;
            }
            return fail("flag groups are only supported at the start of the pattern by this build");
        }
        if (c == '.') { out += "[^\\n]"; continue; }     // the crate's dot stops at \n only; ECMAScript's also at \r, U+2028 and U+2029
        out += c;
    }
    if (in_class) return fail("unclosed character class");
    return out;
}

inline CleavePattern compile_cleave_pattern(const std::string& src) {
    std::regex::flag_type flags;
    const std::string ecma = rewrite_cleave_pattern(src, flags);
    try { return CleavePattern{std::regex(ecma, flags)}; }
    catch (const std::regex_error& e) { throw CleavePatternError("regex parse error:\n    " + src + "\nerror: " + e.what()); }
}

}  // namespace umgap_host
