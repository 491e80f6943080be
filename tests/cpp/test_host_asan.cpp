// Host half of the model builder (vq_amd/csrc/engine_host.hip) under AddressSanitizer + UBSan, CPU only: duplicate
// detection, centring, and every tile-image builder over a sweep of shapes, none of which needs a device.  Built by
// `make asan` (hipcc, -fsanitize=address,undefined on the host side only); run by tests/test_abi_cpu.py.
// The translation unit is included whole so that its file-local builders are visible here.
#include "../../vq_amd/csrc/engine_host.hip"

#include <cstdio>
#include <random>

using namespace vq;

// The kernels themselves are not linked into this program: the launcher look-ups answer with a launcher that refuses.
namespace vq {
static hipError_t no_device(const EngineArgs &, bool, hipStream_t) { return hipErrorNoDevice; }
static bool pow2_in(int v, int lo, int hi) { return v >= lo && v <= hi && (v & (v - 1)) == 0; }
LaunchFn streamed_launcher(int, int R1, int, int *spb) { *spb = 128; return pow2_in(R1, 16, 256) ? &no_device : nullptr; }
LaunchFn omega_finish_launcher(int R) { return R == 32 || R == 64 || R == 128 ? &no_device : nullptr; }
LaunchFn streamed_finish_launcher(int, int R1, int) { return R1 == 64 || R1 == 128 ? &no_device : nullptr; }
LaunchFn resident_launcher(int R, int *spb) { *spb = 64; return pow2_in(R, 16, 128) ? &no_device : nullptr; }
LaunchFn resident_pipe_launcher(int R, bool) { return pow2_in(R, 16, 256) ? &no_device : nullptr; }
LaunchFn resident_cut_launcher(int R) { return pow2_in(R, 16, 256) ? &no_device : nullptr; }
LaunchFn resident_finish_launcher(int R) { return pow2_in(R, 16, 128) ? &no_device : nullptr; }
LaunchFn resident_staged_launcher(int R, size_t *bytes) { *bytes = 8 * 16896; return pow2_in(R, 32, 128) ? &no_device : nullptr; }
LaunchFn omega_pipe_launcher(int R) { return pow2_in(R, 32, 128) ? &no_device : nullptr; }
LaunchFn lowrank_launcher(int R1) { return pow2_in(R1, 32, 256) ? &no_device : nullptr; }
LaunchFn direct_launcher(int R, int KP, int *spb) { *spb = 64; return pow2_in(R, 32, 256) && (KP == 4 || KP == 8) ? &no_device : nullptr; }
LaunchFn long_launcher(int tiles) { return tiles == 1 || tiles == 2 || tiles == 4 || tiles == 8 ? &no_device : nullptr; }
LaunchFn local_long_launcher(int tiles) { return tiles == 1 || tiles == 2 || tiles == 4 ? &no_device : nullptr; }
LaunchFn omega_long_launcher(int R2) { return pow2_in(R2, 32, 128) ? &no_device : nullptr; }
LaunchFn local_pj_launcher(int R1, int *spb) { *spb = R1 == 256 ? 256 : R1 >= 64 ? 512 : 384; return pow2_in(R1, 16, 256) ? &no_device : nullptr; }
}  // namespace vq

namespace {
int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); ++g_fail; } } while (0)

std::vector<float> uniform(size_t n, uint32_t seed, float lo = -1.0f, float hi = 1.0f)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(lo, hi);
    std::vector<float> v(n);
    for (float &x : v) x = u(g);
    return v;
}
}  // namespace

int main()
{
    // ---- rows that repeat an earlier row of their class are left out of the screening
    {
        std::vector<float> Q = {1, 2, 3, 4, 1, 2, 9, 9, 1, 2, -0.0f, 5, 1, 2, 0.0f, 5};      // K = 8, L = 2: rows 0, 2, 4, 6 equal; 5 and 7 equal up to -0
        std::vector<uint32_t> none;
        std::vector<bool> d = shadowed_rows(Q, 8, 2, none);
        CHECK(!d[0] && !d[1] && d[2] && !d[3] && d[4] && !d[5] && d[6] && d[7]);
        std::vector<uint32_t> cls = {0, 0, 1, 0, 0, 2, 0, 3};                               // row 2 has another class: kept; row 4 repeats row 0's
        d = shadowed_rows(Q, 8, 2, cls);
        CHECK(!d[2] && d[4] && !d[7]);
        cls[7] = 2;
        CHECK(shadowed_rows(Q, 8, 2, cls)[7]);
        CHECK(shadowed_rows(std::vector<float>{7.0f}, 1, 1, none).size() == 1);
    }
    // ---- centring: off for data around the origin, on (8 significant bits, exact shift) for data far from it
    {
        std::vector<float> mu, Wc;
        const std::vector<float> W0 = uniform(64 * 20, 1);
        CHECK(!centre_prototypes(W0, 64, 20, mu, Wc) && mu.empty());
        std::vector<float> W1 = uniform(64 * 20, 2);
        for (float &x : W1) x += 100.0f;
        CHECK(centre_prototypes(W1, 64, 20, mu, Wc) && mu.size() == 20 && Wc.size() == W1.size());
        for (size_t i = 0; i < W1.size(); ++i) CHECK((double)Wc[i] == (double)W1[i] - (double)mu[i % 20]);       // Sterbenz: exact
        for (float m : mu) {
            int e;
            const double f = std::frexp((double)m, &e) * 256.0;
            CHECK(f == std::nearbyint(f));
        }
        std::vector<float> Wn = W1;
        Wn[5] = NAN;
        mu.clear();
        CHECK(!centre_prototypes(Wn, 64, 20, mu, Wc));
    }
    // ---- three bf16 pieces add up to the f32 value exactly (normal numbers; what the bf16x6 projections rest on)
    {
        const std::vector<float> v = uniform(20000, 77, -3.0f, 3.0f);
        for (size_t i = 0; i < v.size(); ++i) {
            const float x = i % 3 == 0 ? v[i] * 1e-20f : i % 3 == 1 ? v[i] * 1e20f : v[i];
            uint16_t pc[3];
            split3_host(x, pc);
            CHECK((double)from_bf16(pc[0]) + (double)from_bf16(pc[1]) + (double)from_bf16(pc[2]) == (double)x);
        }
        // k_omega_pipe's Omega pieces: piece (q, t), class c, lane, element e
        EngineModel o;
        o.K = 40; o.D = 100; o.M = 70;
        o.R1 = 128; o.R2 = 128;
        const std::vector<float> Om = uniform((size_t)o.M * o.D, 5), P = uniform((size_t)o.K * o.M, 6);
        std::vector<float> norms(o.K, 1.0f), a, b;
        std::vector<uint32_t> none;
        CHECK(bind_omega_pipe<128>(o, a, b, Om, P, norms, none) && o.pj && a.size() * 4 == (size_t)8 * 4 * 3072);
        const uint16_t *pcs = reinterpret_cast<const uint16_t *>(a.data());
        for (uint32_t row = 0; row < 128; row += 3)
            for (uint32_t col = 0; col < 128; col += 7) {
                const int q = col / 16, t = row / 32, lane = (row % 32) + 32 * ((col / 4) & 1), e = (col & 3) + 4 * ((col / 8) & 1);
                double sum = 0.0;
                for (int c = 0; c < 3; ++c) sum += (double)from_bf16(pcs[((((size_t)q * 4 + t) * 3 + c) * 64 + lane) * 8 + e]);
                CHECK(sum == (row < o.M && col < o.D ? (double)Om[(size_t)row * o.D + col] : 0.0));
            }
    }
    // ---- every image builder over shapes around its boundaries
    const uint32_t Ds[] = {1, 3, 16, 31, 64, 100, 128, 200, 256, 257, 784};
    const uint32_t Ks[] = {1, 2, 8, 9, 33, 256, 1000};
    for (uint32_t D : Ds)
        for (uint32_t K : Ks) {
            const std::vector<float> W = uniform((size_t)K * D, D * 131 + K);
            std::vector<float> norms(K, 1.0f), img1, img2;
            std::vector<uint32_t> cls(K);
            for (uint32_t k = 0; k < K; ++k) cls[k] = k % 3;
            EngineModel m;
            m.K = K; m.D = D;
            m.R1 = D > 256 ? 0u : (uint32_t)pad_pow2(D, 16);
            bool ok = bind_direct(m, img1, W, cls) || bind_long(m, img1, img2, W, norms, cls) || bind_euclid<16>(m, img1, W, norms, cls) ||
                      bind_euclid<32>(m, img1, W, norms, cls) || bind_euclid<64>(m, img1, W, norms, cls) ||
                      bind_euclid<128>(m, img1, W, norms, cls) || bind_euclid<256>(m, img1, W, norms, cls);
            CHECK(ok && m.launch != nullptr && !img1.empty());
            for (uint32_t M : {1u, 16u, 17u, 64u, 128u}) {
                const std::vector<float> Om = uniform((size_t)M * D, M + D);
                std::vector<float> P = uniform((size_t)K * M, K + M);
                EngineModel o;
                o.K = K; o.D = D; o.M = M;
                o.R1 = D > 256 ? 0u : (uint32_t)pad_pow2(D, 16);
                o.R2 = (uint32_t)pad_pow2(M, 16);
                std::vector<float> a, b;
                ok = bind_omega_long<32>(o, a, b, Om, P, norms, cls) || bind_omega_long<64>(o, a, b, Om, P, norms, cls) ||
                     bind_omega_long<128>(o, a, b, Om, P, norms, cls) || bind_lowrank<256>(o, a, b, Om, P, norms, cls) ||
                     bind_lowrank<128>(o, a, b, Om, P, norms, cls) || bind_lowrank<64>(o, a, b, Om, P, norms, cls) ||
                     bind_lowrank<32>(o, a, b, Om, P, norms, cls) || bind_omega_pipe<128>(o, a, b, Om, P, norms, cls) ||
                     bind_omega_pipe<64>(o, a, b, Om, P, norms, cls) || bind_omega_pipe<32>(o, a, b, Om, P, norms, cls) ||
                     bind_omega<16, 16>(o, a, b, Om, P, norms, cls) || bind_omega<32, 16>(o, a, b, Om, P, norms, cls) ||
                     bind_omega<32, 32>(o, a, b, Om, P, norms, cls) || bind_omega<64, 64>(o, a, b, Om, P, norms, cls) ||
                     bind_omega<128, 128>(o, a, b, Om, P, norms, cls) || bind_omega<256, 16>(o, a, b, Om, P, norms, cls) ||
                     bind_omega<256, 64>(o, a, b, Om, P, norms, cls);
                if (ok) CHECK(!a.empty() && o.launch != nullptr);                            // (shapes no builder takes go two-pass)
                if (K <= 9) {                                                                // local Omegas: K x M x D
                    const std::vector<float> Oms = uniform((size_t)K * M * D, K * 7 + M), negp = uniform((size_t)K * M, 3);
                    EngineModel l;
                    l.K = K; l.D = D; l.M = M;
                    l.R1 = o.R1;
                    std::vector<float> c, d;
                    ok = bind_local_long(l, c, d, Oms, negp) || bind_local_pj<16>(l, c, Oms, negp, cls) || bind_local_pj<32>(l, c, Oms, negp, cls) ||
                         bind_local_pj<64>(l, c, Oms, negp, cls) || bind_local_pj<128>(l, c, Oms, negp, cls) || bind_local_pj<256>(l, c, Oms, negp, cls) ||
                         bind_local<16>(l, c, Oms, negp, cls) || bind_local<32>(l, c, Oms, negp, cls) ||
                         bind_local<64>(l, c, Oms, negp, cls) || bind_local<128>(l, c, Oms, negp, cls) || bind_local<256>(l, c, Oms, negp, cls);
                    CHECK(ok && !c.empty());
                    CHECK(l.pj == (D <= 256));
                    if (l.pj && D > 64 && D <= 128) {       // Geom3 tiles: the three pieces of every weight add up to it exactly, tails in place
                        using G3 = Geom3<128>;
                        const uint32_t tpp = (M + 31) / 32;
                        CHECK(c.size() * 4 == (size_t)l.ngroups1 * G3::STAGE_BYTES && l.ngroups1 == K * tpp && l.samples_per_batch == 512);
                        const unsigned char *bytes = reinterpret_cast<const unsigned char *>(c.data());
                        for (uint32_t j = 0; j < K; ++j)
                            for (uint32_t r = 0; r < M; r += 7)
                                for (uint32_t dd = 0; dd < D; dd += 5) {
                                    const unsigned char *tile = bytes + ((size_t)j * tpp + r / 32) * G3::TILE_BYTES;
                                    double sum = 0.0;
                                    for (int cc = 0; cc < 3; ++cc) {
                                        uint16_t pc;
                                        std::memcpy(&pc, tile + G3::piece_off((int)(r % 32), (int)(dd / 16), (int)((dd / 4) & 1), cc) + 2 * ((dd & 3) + 4 * ((dd / 8) & 1)), 2);
                                        sum += (double)from_bf16(pc);
                                    }
                                    CHECK(sum == (double)Oms[((size_t)j * M + r) * D + dd]);
                                    float init;
                                    std::memcpy(&init, tile + G3::W_BYTES + 4 * (r % 32), 4);
                                    CHECK(init == negp[(size_t)j * M + r]);
                                }
                    }
                }
            }
        }
    // ---- the whole builder: without a device the upload fails with a HIP status, never with a crash or a leak
    {
        std::string why;
        hipError_t herr = hipSuccess;
        const std::vector<float> W = uniform(256 * 128, 9);
        EngineModel *e = engine_build(W, 256, 128, std::vector<float>(), OMEGA_NONE, 0, std::vector<uint32_t>(), BuildOptions(), why, herr);
        if (e) engine_free(e);                                                              // (a GPU was present after all)
        else CHECK(herr != hipSuccess || !why.empty());
    }
    std::printf("%s\n", g_fail ? "FAILED" : "host_asan_ok");
    return g_fail ? 1 : 0;
}
