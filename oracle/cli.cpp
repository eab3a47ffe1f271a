// cli.cpp — `umgap_oracle <subcommand>`: the CPU oracle as stdin->stdout filters, used by
// tests/ to diff the product CLI byte for byte.  TEST INFRASTRUCTURE ONLY (see
// umgap_oracle.hpp).  Minimal flag parsing: tests always pass flags in separated form.
//
//   pept2lca      [-o] <index.tsv>            src/commands/pept2lca.rs
//   prot2kmer2lca [-o] [-k K] [-j THREADS] <index.tsv>     src/commands/prot2kmer2lca.rs (-j: the lookup on N threads, as rayon does)
//   prot2tryp2lca [-o] [-l N] [-L N] [-k SET] [-d SET] <index.tsv>   src/commands/prot2tryp2lca.rs
//   taxa2agg      [-s] [-r] [-m M] [-a A] [-f F] [-l L] <taxons.tsv>   src/commands/taxa2agg.rs
//   seedextend    [-s N] [-g N] [-r taxons.tsv] [-p N]                 src/commands/seedextend.rs
//   uniq          [-s SEP] [-w] [-d DELIM]                             src/commands/uniq.rs
//
// The index argument is the TSV that `umgap buildindex` consumes (the fst file format is the
// third-party crate's).  Exit codes follow src/main.rs:8: Error => 1, panic => 101.
#include <iostream>

#include "umgap_oracle.hpp"

using namespace umgap_oracle;

static SortedIndex load_index(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(ErrKind::Io, std::string(strerror(errno)));
    return SortedIndex::from_tsv(f);
}

int main(int argc, char** argv) {
    std::ios::sync_with_stdio(false);
    std::vector<std::string> a(argv + 1, argv + argc);
    try {
        if (a.empty()) throw Error(ErrKind::InvalidInvocation, "missing subcommand");
        std::string cmd = a[0];
        auto flag = [&](const std::string& f) {
            for (size_t i = 1; i < a.size(); i++) if (a[i] == f) { a.erase(a.begin() + (long)i); return true; }
            return false;
        };
        auto opt = [&](const std::string& f) -> std::optional<std::string> {
            for (size_t i = 1; i + 1 < a.size(); i++)
                if (a[i] == f) { std::string v = a[i + 1]; a.erase(a.begin() + (long)i, a.begin() + (long)i + 2); return v; }
            return std::nullopt;
        };
        if (cmd == "pept2lca") {
            bool o = flag("-o");
            flag("-m");
            size_t c = 240;
            if (auto v = opt("-c")) c = parse_usize(*v);
            if (a.size() != 2) throw Error(ErrKind::InvalidInvocation, "usage");
            SortedIndex ix = load_index(a[1]);
            pept2lca(std::cin, std::cout, ix, o, c);
        } else if (cmd == "prot2kmer2lca") {
            bool o = flag("-o");
            flag("-m");
            size_t c = 240;
            if (auto v = opt("-c")) c = parse_usize(*v);
            size_t k = 9;
            if (auto v = opt("-k")) k = parse_usize(*v);
            unsigned threads = 1;      // -j N: chunks looked up on N threads (the reference's rayon par_bridge); not a reference flag
            if (auto v = opt("-j")) threads = (unsigned)parse_usize(*v);
            if (a.size() != 2) throw Error(ErrKind::InvalidInvocation, "usage");
            SortedIndex ix = load_index(a[1]);
            prot2kmer2lca_threads(std::cin, std::cout, ix, k, o, c, threads);
        } else if (cmd == "prot2tryp2lca") {
            bool o = flag("-o");
            flag("-m");
            size_t c = 240;
            if (auto v = opt("-c")) c = parse_usize(*v);
            size_t l = 5, L = 50;
            std::string keep, drop;
            if (auto v = opt("-l")) l = parse_usize(*v);
            if (auto v = opt("-L")) L = parse_usize(*v);
            if (auto v = opt("-k")) keep = *v;
            if (auto v = opt("-d")) drop = *v;
            if (a.size() != 2) throw Error(ErrKind::InvalidInvocation, "usage");
            SortedIndex ix = load_index(a[1]);
            prot2tryp2lca(std::cin, std::cout, ix, o, l, L, keep, drop, c);
        } else if (cmd == "taxa2agg") {
            bool scored = flag("-s"), ranked = flag("-r");
            Method m = Method::Tree;
            Strategy s = Strategy::Hybrid;
            float factor = 0.25f, lb = 0.0f;
            if (auto v = opt("-m")) m = parse_method(*v);
            if (auto v = opt("-a")) s = parse_strategy(*v);
            if (auto v = opt("-f")) factor = parse_f32(*v);
            if (auto v = opt("-l")) lb = parse_f32(*v);
            if (a.size() != 2) throw Error(ErrKind::InvalidInvocation, "usage");
            Taxonomy tx(read_taxa_file(a[1]));
            // tie rule of the product: smallest Euler-tour first occurrence, then id
            std::unordered_map<TaxonId, size_t> first;
            {
                size_t i = 0;
                for (auto& p : euler_tour(tx.tree)) { first.emplace(p.first, i); i++; }
            }
            auto pick = [&](const std::set<TaxonId>& set) {
                TaxonId best = *set.begin();
                size_t bk = SIZE_MAX;
                for (TaxonId t : set) {
                    auto it = first.find(t);
                    size_t k = it == first.end() ? SIZE_MAX : it->second;
                    if (k < bk) { bk = k; best = t; }
                }
                return best;
            };
            taxa2agg(std::cin, std::cout, std::cerr, tx, scored, ranked, m, s, factor, lb, pick);
        } else if (cmd == "seedextend") {
            size_t s = 2, g = 0, p = 5;
            if (auto v = opt("-s")) s = parse_usize(*v);
            if (auto v = opt("-g")) g = parse_usize(*v);
            if (auto v = opt("-p")) p = parse_usize(*v);
            std::optional<TaxonList> by_id;
            if (auto v = opt("-r")) by_id = TaxonList::new_with_unknown(read_taxa_file(*v), true);
            try {
                // --overflow-checks: the outcome of a build with overflow checks (debug); default = release, what ships
                seedextend(std::cin, std::cout, s, g, by_id ? &*by_id : nullptr, p, flag("--overflow-checks"));
            } catch (Error& e) { std::cout.flush(); throw; }
        } else if (cmd == "uniq") {
            std::string sep = "\n";
            if (auto v = opt("-s")) sep = *v;
            bool w = flag("-w");
            std::optional<std::string> d = opt("-d");
            uniq(std::cin, std::cout, sep, w, d);
        } else {
            throw Error(ErrKind::InvalidInvocation, "unknown subcommand " + cmd);
        }
        std::cout.flush();
        return 0;
    } catch (Error& e) {
        std::cout.flush();
        std::cerr << "Error: " << e.msg << "\n";
        for (auto& c : e.causes) std::cerr << "Caused by: " << c << "\n";
        return 1;
    } catch (Panic& p) {
        std::cout.flush();
        std::cerr << "thread 'main' panicked at '" << p.msg << "'\n";
        return 101;
    }
}
