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
        {   // the latency-path layout of the same program (build_few_program): every op exactly once, and — replaying the levels — every
            // wire an op reads was produced by an earlier level (or is an input), also after the check-only constraints moved to the end
            const FewProgram fp = build_few_program(sp);
            const uint32_t nlev = sp.sched[0]; const uint32_t* lstart = sp.sched.data() + 1; const uint32_t* ops = sp.sched.data() + 2 + nlev;
            std::vector<uint8_t> have(cs.n_wires(), 0);
            for (size_t w = 0; w < cs.n_public + cs.n_secret; w++) have[w] = 1;
            size_t expect = 0, ncount = 0, bad = 0, last_width = 0;
            for (uint32_t l = 0; l < nlev; l++) for (uint32_t k = lstart[l]; k < lstart[l + 1]; k++) {
                const uint32_t at = ops[k], op = sp.words[at] & 0xFF;
                if (op == OP_COUNT) ncount++; else expect += op == OP_LOOKUP ? sp.words[at + 2] : 1;
            }
            for (uint32_t l = 0; l < nlev; l++) {
                std::vector<uint32_t> produced;
                auto reads = [&](uint32_t toff, uint32_t n) { for (uint32_t t = 0; t < n; t++) { const uint32_t wid = fp.terms[2 * (toff + t) + 1]; if (wid != WIRE_CONST && (wid >= have.size() || !have[wid])) bad++; } };
                for (uint32_t i = fp.level_start[l]; i < fp.level_start[l + 1]; i++) {
                    const uint32_t* d = fp.ops.data() + 8 * (size_t)i; const uint32_t op = d[0] & 0xFF;
                    reads(d[4], d[5] + d[6] + d[7]);
                    if (op == OP_R1C) { if (d[0] >> 8) produced.push_back(d[2]); }
                    else if (op == OP_NBITS || op == OP_RANDOMIZE || op == OP_COMMIT) for (uint32_t q = 0; q < d[2]; q++) produced.push_back(d[1] + q);
                    else if (op == OP_LOOKUP) produced.push_back(d[1]);
                    else bad++;
                }
                for (uint32_t c = fp.count_first[l]; c < fp.count_first[l + 1]; c++) {
                    const uint32_t at = fp.count_ops[4 * c], nq = fp.count_ops[4 * c + 2], ntab = sp.words[at + 2];
                    for (uint32_t q = 0; q < nq; q++) { uint32_t w = fp.count_qoff[fp.count_ops[4 * c + 1] + q]; for (int e = 0; e < 2; e++) { const uint32_t n = sp.words[w]; for (uint32_t t = 0; t < n; t++) { const uint32_t wid = sp.words[w + 2 + 2 * t]; if (wid != WIRE_CONST && !have[wid]) bad++; } w += 1 + 2 * n; } }
                    for (uint32_t r = 0; r < ntab; r++) produced.push_back(sp.words[at + 1] + r);
                }
                for (uint32_t w : produced) { if (w >= have.size() || have[w]) bad++; else have[w] = 1; }
                if (fp.level_start[l + 1] > fp.level_start[l]) last_width = fp.level_start[l + 1] - fp.level_start[l];
            }
            size_t unsolved = 0; for (uint8_t h : have) unsolved += !h;
            printf("%s.few_ops=%zu %s.few_expected=%zu %s.few_count_ops=%zu %s.few_bad=%zu %s.few_unsolved=%zu %s.few_last_level=%zu\n", names[i], fp.level_start.back() + 0ul, names[i], expect, names[i], (size_t)(fp.count_ops.size() / 4) + 0 * ncount,
                   names[i], bad, names[i], unsolved, names[i], last_width);
        }
        if (i == 0) { try { parse_r1cs(b.data(), b.size() - 100); printf("truncated.r1cs=accepted\n"); } catch (const std::exception&) { printf("truncated.r1cs=rejected\n"); } }
    }
    auto k = slurp(d + "/pk.chacha20");
    PkFile pk = parse_pk(k.data(), k.size());
    printf("pk.n=%llu pk.A=%zu pk.B=%zu pk.Z=%zu pk.K=%zu pk.B2=%zu\n", (unsigned long long)pk.domain_n, pk.g1_A.size() / 32, pk.g1_B.size() / 32, pk.g1_Z.size() / 32, pk.g1_K.size() / 32, pk.g2_B.size() / 64);
    try { parse_pk(k.data(), k.size() - 1); printf("truncated.pk=accepted\n"); } catch (const std::exception&) { printf("truncated.pk=rejected\n"); }
    return 0;
}
