// Standalone benchmark / checker for sq_linear_ts_f16 (no Python, no torch: starts in a second on a GPU box).
//   hipcc -O2 -std=c++17 tools/ts_bench.cpp -o tools/ts_bench -Iinclude -Lsequoia_amd/lib -lsequoia_hip -Wl,-rpath,'$ORIGIN/../sequoia_amd/lib'
//   tools/ts_bench [M ...]         (default M = 48 64 128)
// Weights rotate over enough distinct buffers (> 512 MB) that neither L2 nor the 256 MiB Infinity Cache holds them.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sequoia_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline float urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.0f; }

static void fill_half(std::vector<_Float16>& v, float scale) { for (auto& x : v) x = (_Float16)((urand() * 2.f - 1.f) * scale); }

struct Shape { const char* name; int n_out, k, silu, res; };

int main(int argc, char** argv) {
    const char* only = getenv("TS_ONLY");
    std::vector<int> ms;
    for (int i = 1; i < argc; ++i) ms.push_back(atoi(argv[i]));
    if (ms.empty()) ms = {48, 64, 128};
    // TS_ARCH=13b: the Llama-2-13b projection shapes (configuration D) instead of the 7B ones
    const bool a13 = getenv("TS_ARCH") && !strcmp(getenv("TS_ARCH"), "13b");
    const Shape shapes7[] = {{"qkv", 12288, 4096, 0, 0}, {"o+res", 4096, 4096, 0, 1}, {"gate_up+silu", 11008, 4096, 1, 0},
                             {"down+res", 4096, 11008, 0, 1}, {"lm_head", 32000, 4096, 0, 0}};
    const Shape shapes13[] = {{"qkv", 15360, 5120, 0, 0}, {"o+res", 5120, 5120, 0, 1}, {"gate_up+silu", 13824, 5120, 1, 0},
                              {"down+res", 5120, 13824, 0, 1}, {"lm_head", 32000, 5120, 0, 0}};
    // TS_ARCH=70b: the FULL-WIDTH Llama-2-70b projections (configuration E at TP = 1: 64 query / 8 KV heads of 128, inter 28672)
    const bool a70 = getenv("TS_ARCH") && !strcmp(getenv("TS_ARCH"), "70b");
    const Shape shapes70[] = {{"qkv", 10240, 8192, 0, 0}, {"o+res", 8192, 8192, 0, 1}, {"gate_up+silu", 28672, 8192, 1, 0},
                              {"down+res", 8192, 28672, 0, 1}, {"lm_head", 32000, 8192, 0, 0}};
    const Shape* shapes_p = a70 ? shapes70 : (a13 ? shapes13 : shapes7);
    std::vector<Shape> shapes(shapes_p, shapes_p + 5);
    const size_t slab_cap = 512ull << 20;
    void* slab; CK(hipMalloc(&slab, slab_cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        if (only && !strstr(only, sh.name)) continue;
        const int wrows = sh.silu ? 2 * sh.n_out : sh.n_out;
        const size_t wbytes = (size_t)wrows * sh.k * 2;
        // TS_NBUF=1: the same weight buffer every launch (hot: whatever the 256 MiB Infinity Cache keeps of it is re-read from
        // there) -- the cold / hot difference bounds what a run-ahead prefetch into the Infinity Cache could buy
        const int nbuf = getenv("TS_NBUF") ? atoi(getenv("TS_NBUF")) : (int)((640ull << 20) / wbytes) + 1;
        std::vector<_Float16> hw((size_t)wrows * sh.k);
        fill_half(hw, 0.05f);
        std::vector<void*> dws(nbuf);
        void* dw_rm; CK(hipMalloc(&dw_rm, wbytes)); CK(hipMemcpy(dw_rm, hw.data(), wbytes, hipMemcpyHostToDevice));
        for (int b = 0; b < nbuf; ++b) {
            CK(hipMalloc(&dws[b], wbytes));
            if (sq_repack_linear_weight_f16(dw_rm, dws[b], wrows, sh.k, nullptr) != 0) { printf("repack failed: %s\n", sq_last_error()); return 1; }
        }
        CK(hipDeviceSynchronize()); CK(hipFree(dw_rm));
        const int units = sh.n_out / 16;
        for (int m : ms) {
            std::vector<_Float16> ha((size_t)m * sh.k), hr((size_t)m * sh.n_out);
            fill_half(ha, 1.0f); fill_half(hr, 1.0f);
            void *da, *da_rm, *dr, *dout;
            const size_t mpad = (size_t)((m + 15) / 16) * 16;
            CK(hipMalloc(&da_rm, ha.size() * 2)); CK(hipMalloc(&da, mpad * sh.k * 2)); CK(hipMalloc(&dr, hr.size() * 2));
            CK(hipMalloc(&dout, mpad * sh.n_out * 2));
            CK(hipMemcpy(da_rm, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
            if (sq_repack_rows_frag_f16(da_rm, sh.k, da, m, sh.k, nullptr) != 0) { printf("row repack failed\n"); return 1; }
            CK(hipDeviceSynchronize()); CK(hipFree(da_rm));
            CK(hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
            // (tiles, splits) candidates
            std::vector<std::pair<int, int>> cands;
            const int max_u = sh.silu ? 4 : 8;
            for (int upt = (getenv("TS_MIN_U") ? atoi(getenv("TS_MIN_U")) : 1); upt <= max_u; ++upt)
                for (int sp : {1, 2, 3, 4, 6, 8}) {
                    const int tiles = (units + upt - 1) / upt;
                    if (sp > 1 && (sh.silu || sh.n_out > 16384)) continue;
                    if (sp == 1 && sh.res == 0 && false) continue;
                    const long wgs = (long)tiles * sp;
                    if (wgs < 128 || wgs > 2100) continue;
                    if ((sh.k / 32) < sp * 4 * 2) continue;
                    cands.push_back({tiles, sp});
                }
            for (int t : {256, 512}) if (!sh.silu || true) { if ((units + t - 1) / t <= max_u && units >= t) cands.push_back({t, 1}); }
            if (getenv("TS_TILES") && getenv("TS_SPLITS")) { cands.clear(); cands.push_back({atoi(getenv("TS_TILES")), atoi(getenv("TS_SPLITS"))}); }
            if (const char* cs = getenv("TS_CANDS")) {          // explicit list: "64x4,96x2,..."
                cands.clear();
                for (const char* q = cs; *q;) {
                    int t = 0, sp = 0, used = 0;
                    if (sscanf(q, "%dx%d%n", &t, &sp, &used) != 2) break;
                    cands.push_back({t, sp});
                    q += used;
                    if (*q == ',') ++q;
                }
            }
            for (auto [tiles, splits] : cands) {
                const size_t need = sq_linear_ts_workspace_bytes(m, sh.n_out, splits);
                if (need > slab_cap) continue;
                const bool use_res = sh.res && splits == 1;
                const int out_frag = sh.silu ? 1 : 0;          // the SwiGLU output feeds down_proj: fragment-major
                auto run = [&](int b) {
                    return sq_linear_ts_f16(da, dws[b % nbuf], use_res ? dr : nullptr, dout, sh.n_out, out_frag, m, sh.n_out, sh.k, sh.silu,
                                            tiles, splits, slab, slab_cap, nullptr);
                };
                CK(hipMemset(dout, 0xff, mpad * sh.n_out * 2));
                int rc = run(0);
                if (rc != 0) { printf("%-13s M=%3d tiles=%4d splits=%2d rc=%d (%s)\n", sh.name, m, tiles, splits, rc, sq_last_error()); continue; }
                CK(hipDeviceSynchronize());
                std::vector<_Float16> ho(mpad * sh.n_out);
                std::vector<float> hs;
                if (splits > 1) { hs.resize((size_t)splits * m * sh.n_out); CK(hipMemcpy(hs.data(), slab, hs.size() * 4, hipMemcpyDeviceToHost)); }
                else CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
                double max_err = 0; int bad = 0;
                const int rows[] = {0, m / 3, m / 2, m - 1};
                for (int r : rows)
                    for (int c = 0; c < sh.n_out; c += 97) {
                        double acc = 0, acc2 = 0;
                        for (int kk = 0; kk < sh.k; ++kk) {
                            acc += (double)ha[(size_t)r * sh.k + kk] * (double)hw[(size_t)c * sh.k + kk];
                            if (sh.silu) acc2 += (double)ha[(size_t)r * sh.k + kk] * (double)hw[(size_t)(c + sh.n_out) * sh.k + kk];
                        }
                        double ref, got;
                        if (splits > 1) {
                            float sum = 0; for (int s2 = 0; s2 < splits; ++s2) sum += hs[((size_t)s2 * m + r) * sh.n_out + c];
                            got = sum; ref = acc;
                        } else {
                            if (sh.silu) { float g = (float)(_Float16)(float)acc; _Float16 sg = (_Float16)(g / (1.f + expf(-g))); ref = (float)(_Float16)((float)sg * (float)(_Float16)(float)acc2); }
                            else { _Float16 h = (_Float16)(float)acc; ref = use_res ? (float)(_Float16)((float)h + (float)hr[(size_t)r * sh.n_out + c]) : (float)h; }
                            const size_t mtp = mpad / 16;
                            const size_t foff = ((((size_t)(c >> 5)) * mtp + (r >> 4)) * 64 + ((c >> 3) & 3) * 16 + (r & 15)) * 8 + (c & 7);
                            got = (float)ho[out_frag ? foff : (size_t)r * sh.n_out + c];
                        }
                        const double err = fabs(got - ref);
                        if (!(err <= 4e-3 + 4e-3 * fabs(ref))) ++bad;
                        if (err > max_err || std::isnan(got)) max_err = std::isnan(got) ? 1e9 : err;
                    }
                const int reps = 40;
                for (int i = 0; i < 5; ++i) run(i + 1);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) run(i);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms_; CK(hipEventElapsedTime(&ms_, e0, e1));
                const double us = ms_ * 1e3 / reps;
                printf("%-13s M=%3d tiles=%4d splits=%2d wgs=%5d  %7.1f us  %5.2f TB/s   max_err %.2e %s\n", sh.name, m, tiles, splits,
                       tiles * splits, us, wbytes / us / 1e6, max_err, bad ? "MISMATCH" : "ok");
                fflush(stdout);
            }
            CK(hipFree(da)); CK(hipFree(dr)); CK(hipFree(dout));
        }
        for (void* p : dws) CK(hipFree(p));
    }
    return 0;
}
