// CPU-only exerciser for the host decoders (formats.cpp): prints key=value facts about the reference's files.
#include "formats.hpp"
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
using namespace gsc;
static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string d = argv[1];
    const char* names[3] = {"chacha", "aes128", "aes256"}; const char* files[3] = {"r1cs.chacha20", "r1cs.aes128", "r1cs.aes256"};
    for (int i = 0; i < 3; i++) {
        auto b = slurp(d + "/" + files[i]);
        R1csFile cs = parse_r1cs(b.data(), b.size());
        SolverProgram sp = build_solver_program(cs);
        printf("%s.instr=%zu %s.wires=%zu %s.constraints=%zu %s.inversions=%zu %s.tables=%zu %s.words=%zu %s.levels=%zu %s.width=%zu %s.commit_level=%zu\n", names[i], cs.n_instr(), names[i], cs.n_wires(),
               names[i], cs.n_constraints, names[i], sp.n_inversions, names[i], sp.n_tables, names[i], sp.words.size(), names[i], sp.n_levels, names[i], sp.max_level_width, names[i], sp.commit_level);
        if (i == 0) { try { parse_r1cs(b.data(), b.size() - 100); printf("truncated.r1cs=accepted\n"); } catch (const std::exception&) { printf("truncated.r1cs=rejected\n"); } }
    }
    auto k = slurp(d + "/pk.chacha20");
    PkFile pk = parse_pk(k.data(), k.size());
    printf("pk.n=%llu pk.A=%zu pk.B=%zu pk.Z=%zu pk.K=%zu pk.B2=%zu\n", (unsigned long long)pk.domain_n, pk.g1_A.size() / 32, pk.g1_B.size() / 32, pk.g1_Z.size() / 32, pk.g1_K.size() / 32, pk.g2_B.size() / 64);
    try { parse_pk(k.data(), k.size() - 1); printf("truncated.pk=accepted\n"); } catch (const std::exception&) { printf("truncated.pk=rejected\n"); }
    return 0;
}
