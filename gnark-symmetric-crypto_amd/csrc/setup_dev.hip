// Device side of Groth16 Setup: multiples of the group generators (setup.hpp).
#include "setup.hpp"
#include "kernels.hpp"
#include <stdexcept>
#include <string>

namespace gsc {

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); } while (0)

namespace {
struct Buf {
    void* p = nullptr; size_t bytes = 0; bool secret = false;      // secret: zeroed before it is freed, also when an exception unwinds
    explicit Buf(size_t n, bool is_secret = false) : bytes(n ? n : 1), secret(is_secret) { HIP_CHECK(hipMalloc(&p, bytes)); }
    Buf(const Buf&) = delete; Buf& operator=(const Buf&) = delete;
    ~Buf() { if (p) { if (secret) { (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize(); } (void)hipFree(p); } }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

template <class AffT, class XyzzT, class Shift, class Build, class Mul>
void run(const uint8_t* gen_mont, const uint8_t* scalars_le, size_t n, uint8_t* out, uint8_t* inf, Shift shift, Build build, Mul mul) {
    constexpr int c = 16; const int nwin = msm_windows(c); const size_t D = (size_t)1 << (c - 1);
    hipStream_t s = nullptr;
    Buf gen(sizeof(AffT)), bases(sizeof(AffT) * nwin), d_src(4 * nwin), d_shift(4 * nwin), table(sizeof(AffT) * nwin * D);
    HIP_CHECK(hipMemcpy(gen.p, gen_mont, sizeof(AffT), hipMemcpyHostToDevice));
    std::vector<uint32_t> src(nwin, 0u), sh(nwin);
    for (int j = 0; j < nwin; j++) sh[j] = (uint32_t)(c * j);
    HIP_CHECK(hipMemcpy(d_src.p, src.data(), 4 * nwin, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(d_shift.p, sh.data(), 4 * nwin, hipMemcpyHostToDevice));
    shift(gen.as<AffT>(), d_src.as<uint32_t>(), d_shift.as<uint32_t>(), (size_t)nwin, bases.as<AffT>(), s);
    const uint32_t cap = 256;
    std::vector<MsmRowSeg> segs;
    for (int j = 0; j < nwin; j++) for (uint32_t f = 0; f < D; f += cap) segs.push_back(MsmRowSeg{(uint32_t)j, f + 1, cap, 0u, (uint64_t)j * D + f});
    Buf d_segs(sizeof(MsmRowSeg) * segs.size()), scratch(sizeof(XyzzT) * segs.size() * cap);
    HIP_CHECK(hipMemcpy(d_segs.p, segs.data(), sizeof(MsmRowSeg) * segs.size(), hipMemcpyHostToDevice));
    build(bases.as<AffT>(), d_segs.as<MsmRowSeg>(), segs.size(), cap, table.as<AffT>(), scratch.as<XyzzT>(), s);
    HIP_CHECK(hipGetLastError());
    Buf d_sc(32 * n, true), d_out(sizeof(AffT) * n), d_inf(n);      // the scalars are toxic waste in disguise: never left in freed device memory
    HIP_CHECK(hipMemcpy(d_sc.p, scalars_le, 32 * n, hipMemcpyHostToDevice));
    mul(table.as<AffT>(), c, nwin, d_sc.as<fe>(), n, d_out.as<fe>(), d_inf.as<uint8_t>(), s);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, d_out.p, sizeof(AffT) * n, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(inf, d_inf.p, n, hipMemcpyDeviceToHost));
}
}  // namespace

void setup_generator_muls(int device, bool g2, const uint8_t* gen_mont, const uint8_t* scalars_le, size_t n, uint8_t* out, uint8_t* inf) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available: Setup computes its group elements on the GPU");
    HIP_CHECK(hipSetDevice(device));
    if (!n) return;
    if (g2) run<G2Aff, G2Xyzz>(gen_mont, scalars_le, n, out, inf, launch_shift_bases_g2, launch_build_rows_g2, launch_fixed_mul_g2);
    else run<G1Aff, G1Xyzz>(gen_mont, scalars_le, n, out, inf, launch_shift_bases_g1, launch_build_rows_g1, launch_fixed_mul_g1);
}

}  // namespace gsc
