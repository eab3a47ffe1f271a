// Drives umgap_amd/csrc/host/cleave_pattern.hpp for tests/test_cli_text_cpu.py: lines "pattern<TAB>sequence" on stdin,
// one line out per line in: the sequence after the two substitutions with its newlines shown as '|', or "ERROR <message>".
#include <cstdio>
#include <iostream>
#include <string>

#include "../../umgap_amd/csrc/host/cleave_pattern.hpp"

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) return 2;
        try {
            const umgap_host::CleavePattern p = umgap_host::compile_cleave_pattern(line.substr(0, tab));
            std::string cut = p.cut(line.substr(tab + 1));
            for (char& c : cut) if (c == '\n') c = '|';
            std::printf("%s\n", cut.c_str());
        } catch (const umgap_host::CleavePatternError& e) {
            std::string what = e.what();
            for (char& c : what) if (c == '\n') c = '|';
            std::printf("ERROR %s\n", what.c_str());
        }
    }
    return 0;
}
