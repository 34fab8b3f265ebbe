// experiment + self-check: row-cooperative Fp / G1 arithmetic (hip/fpc_dev.cuh, hip/msm_cl.cuh) against the one-lane-per-element
// form (hip/g1_dev.cuh) and the host library (ff/g1.hpp): products, sums, differences, point additions with every special case,
// the 512-thread reduction tree, the plane kernel + tree + Horner against the host's Pippenger; then latencies of dependent chains.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkcnn_amd/csrc scripts/exp/cl_lab.hip -o scripts/exp/cl_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "hip/msm_cl.cuh"
#include "ff/g1.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// out[0][i] = a b, out[1][i] = a + b, out[2][i] = a - b: one-lane form
__global__ void k_ops_sl(fp_t *out, const fp_t *a, const fp_t *b, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fp_mul(a[i], b[i]);
    out[n + i] = fp_add(a[i], b[i]);
    out[2 * n + i] = fp_sub(a[i], b[i]);
}
// the same in row form: one element per 16 lanes
__global__ void k_ops_cl(fp_t *out, const fp_t *a, const fp_t *b, uint32_t n) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const uint32_t m = fpc_mod_limb();
    const uint32_t x = (i < n && l < 12) ? a[i].v[l] : 0u, y = (i < n && l < 12) ? b[i].v[l] : 0u;
    const uint32_t p = fpc_mul(x, y, m), s = fpc_add(x, y, m), d = fpc_sub(x, y, m);
    if (i < n && l < 12) { out[i].v[l] = p; out[n + i].v[l] = s; out[2 * n + i].v[l] = d; }
}
__global__ void k_padd_sl(g1j_t *out, const g1j_t *p, const g1j_t *q, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g1_add_any(p[i], q[i]);
}
__global__ void k_padd_cl(g1j_t *out, const g1j_t *p, const g1j_t *q, uint32_t n) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t m = fpc_mod_limb();
    const g1c_t a = i < n ? g1c_load(p + i) : g1c_inf(), b = i < n ? g1c_load(q + i) : g1c_inf();
    const g1c_t r = g1c_add(a, b, m);
    if (i < n) g1c_store(out + i, r);
}
// dependent chains: x <- x * y, `iters` times
__global__ void k_chain_sl(fp_t *out, const fp_t *a, int iters) {
    fp_t x = a[threadIdx.x], y = a[64 + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = fp_mul(x, y);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_chain_cl(fp_t *out, const fp_t *a, int iters) {
    const uint32_t r = threadIdx.x >> 4, l = threadIdx.x & 15, m = fpc_mod_limb();
    uint32_t x = l < 12 ? a[r].v[l] : 0u;
    const uint32_t y = l < 12 ? a[64 + r].v[l] : 0u;
    for (int i = 0; i < iters; ++i) x = fpc_mul(x, y, m);
    if (l < 12) out[blockIdx.x * 4 + r].v[l] = x;
}
__global__ void k_pchain_sl(g1j_t *out, const g1j_t *p, int iters) {
    g1j_t x = p[threadIdx.x];
    const g1j_t y = p[64 + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = g1_add_any(x, y);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_pchain_cl(g1j_t *out, const g1j_t *p, int iters) {
    const uint32_t r = threadIdx.x >> 4, m = fpc_mod_limb();
    g1c_t x = g1c_load(p + r);
    const g1c_t y = g1c_load(p + 64 + r);
    for (int i = 0; i < iters; ++i) x = g1c_add(x, y, m);
    g1c_store(out + blockIdx.x * 4 + r, x);
}
__global__ void __launch_bounds__(512) k_tree(g1j_t *out, const g1j_t *in, uint32_t n_seg, uint32_t n_in) { k_cl_tree(out, in, n_seg, n_in); }
__global__ void __launch_bounds__(128) k_horner(g1j_t *out, const g1j_t *in) { k_cl_horner(out, in); }
__global__ void __launch_bounds__(64) k_acc(g1j_t *out, const fr_t *mag, uint64_t ld, const uint32_t *idx, const g1a_t *T, uint32_t m, uint32_t cols, uint32_t cpt, uint32_t wsplit, uint32_t w_lo) {
    k_planes_acc(out, mag, ld, idx, T, m, cols, cpt, wsplit, w_lo, 32u);
}
__global__ void __launch_bounds__(64) k_wtab(g1a_t *T, g1j_t *J, fp_t *pre, uint32_t m, uint32_t w0, uint32_t w1) { k_window_tables(T, J, pre, m, w0, w1); }
__global__ void k_mags(fr_t *mag, const fr_t *s, uint64_t ld, uint32_t cols) { k_scalar_mags(mag, s, ld, nullptr, cols); }
// the round-5 pair for the same sum
__global__ void __launch_bounds__(64) k_planes5(g1j_t *out, const fr_t *mag, uint64_t ld, const uint32_t *idx, const g1a_t *T, uint32_t m, uint32_t cols, uint32_t cpt, uint32_t wsplit, uint32_t w_lo) {
    k_msm_planes(out, mag, ld, idx, T, m, cols, cpt, wsplit, w_lo);
}
__global__ void __launch_bounds__(512) k_finish5(g1j_t *outJ, const g1j_t *partials, uint32_t nparts) { k_msm_finish(outJ, partials, nparts); }

static zkff::Xoshiro rng;
static zkff::Fp rand_fp() {
    uint64_t t[6];
    for (int i = 0; i < 6; ++i) t[i] = rng.next();
    t[5] &= 0x0fffffffffffffffULL;        // < 2^380 < p
    return zkff::Fp::fromCanonical(t);
}
static zkff::Fr rand_fr() {
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) t[i] = rng.next();
    t[3] &= 0x3fffffffffffffffULL;
    return zkff::Fr(zkff::MontField<zkff::FrParams>::fromCanonical(t));
}
template <class F> static float time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    rng.seed(0x5EED0006);
    int bad = 0;
    // ---- field operations ----
    const uint32_t n = 4096;
    std::vector<zkff::Fp> ha(n), hb(n);
    for (uint32_t i = 0; i < n; ++i) { ha[i] = rand_fp(); hb[i] = rand_fp(); }
    // edge values: 0, 1, p - 1, equal operands
    ha[0].clear(); hb[1].clear(); ha[2] = zkff::Fp::one(); hb[3] = ha[3]; ha[4] = -zkff::Fp::one(); hb[4] = -zkff::Fp::one(); ha[5] = -zkff::Fp::one(); hb[5] = zkff::Fp::one();
    fp_t *da, *db, *o1, *o2;
    CK(hipMalloc((void **) &da, n * 48)); CK(hipMalloc((void **) &db, n * 48)); CK(hipMalloc((void **) &o1, 3 * n * 48)); CK(hipMalloc((void **) &o2, 3 * n * 48));
    CK(hipMemcpy(da, ha.data(), n * 48, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), n * 48, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ops_sl, dim3(n / 64), dim3(64), 0, 0, o1, da, db, n);
    hipLaunchKernelGGL(k_ops_cl, dim3(n * 16 / 256), dim3(256), 0, 0, o2, da, db, n);
    std::vector<zkff::Fp> r1(3 * n), r2(3 * n);
    CK(hipMemcpy(r1.data(), o1, 3 * n * 48, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, 3 * n * 48, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        const zkff::Fp want[3] = {ha[i] * hb[i], ha[i] + hb[i], ha[i] - hb[i]};
        for (int k = 0; k < 3; ++k) {
            if (std::memcmp(&r1[k * n + i], &want[k], 48)) { if (bad++ < 5) printf("one-lane op %d differs from the host at %u\n", k, i); }
            if (std::memcmp(&r2[k * n + i], &want[k], 48)) { if (bad++ < 5) printf("row op %d differs from the host at %u\n", k, i); }
        }
    }
    printf("field ops (mul, add, sub) x %u: %s\n", n, bad ? "MISMATCH" : "row form == one-lane form == host");
    // ---- point additions with special cases ----
    const uint32_t np = 1024;
    std::vector<zkff::G1> P(np), Q(np);
    zkff::G1 g = zkff::G1::generator();
    for (uint32_t i = 0; i < np; ++i) { P[i] = g * rand_fr(); Q[i] = P[i] * rand_fr(); }
    P[0] = zkff::G1(); Q[1] = zkff::G1(); P[2] = zkff::G1(); Q[2] = zkff::G1(); Q[3] = P[3]; Q[4] = -P[4];
    { zkff::G1 t = P[5]; zkff::G1::dbl(t, t); zkff::G1 u = t + P[5]; Q[5] = u - t - P[5] + P[5]; }       // P[5] in other coordinates
    g1j_t *dp, *dq, *po1, *po2;
    CK(hipMalloc((void **) &dp, np * 144)); CK(hipMalloc((void **) &dq, np * 144)); CK(hipMalloc((void **) &po1, np * 144)); CK(hipMalloc((void **) &po2, np * 144));
    CK(hipMemcpy(dp, P.data(), np * 144, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, Q.data(), np * 144, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_padd_sl, dim3(np / 64), dim3(64), 0, 0, po1, dp, dq, np);
    hipLaunchKernelGGL(k_padd_cl, dim3(np * 16 / 256), dim3(256), 0, 0, po2, dp, dq, np);
    std::vector<zkff::G1> s1(np), s2(np);
    CK(hipMemcpy(s1.data(), po1, np * 144, hipMemcpyDeviceToHost)); CK(hipMemcpy(s2.data(), po2, np * 144, hipMemcpyDeviceToHost));
    int pbad = 0;
    for (uint32_t i = 0; i < np; ++i) {
        const zkff::G1 want = P[i] + Q[i];
        if (s1[i] != want) { if (pbad++ < 5) printf("one-lane point addition wrong at %u\n", i); }
        if (s2[i] != want) { if (pbad++ < 5) printf("row point addition wrong at %u\n", i); }
        if (!want.isInf() && i > 5 && std::memcmp(&s1[i], &s2[i], 144)) { if (pbad++ < 5) printf("row / one-lane coordinates differ at %u\n", i); }
    }
    printf("point additions x %u (infinity, P = Q, P = -Q included): %s\n", np, pbad ? "MISMATCH" : "row form == one-lane form == host");
    bad += pbad;
    // ---- reduction tree ----
    for (uint32_t n_in : {1u, 5u, 32u, 64u, 100u, 256u}) {
        const uint32_t groups = np / n_in;
        hipLaunchKernelGGL(k_tree, dim3(1, groups), dim3(512), 0, 0, po2, dp, n_in, n_in);
        CK(hipMemcpy(s2.data(), po2, groups * 144, hipMemcpyDeviceToHost));
        int tb = 0;
        for (uint32_t gidx = 0; gidx < groups; ++gidx) {
            zkff::G1 want;
            for (uint32_t k = 0; k < n_in; ++k) want = want + P[gidx * n_in + k];
            if (s2[gidx] != want) tb++;
        }
        printf("k_cl_tree, %u groups of %u: %s\n", groups, n_in, tb ? "MISMATCH" : "ok");
        bad += tb;
    }
    // ---- a full MSM without byte tables: window tables -> planes -> tree -> tree -> Horner, against the host's Pippenger and round 5's pair ----
    {
        const uint32_t m = 2048, rows = 2;
        std::vector<zkff::G1> gj(m);
        for (uint32_t j = 0; j < m; ++j) gj[j] = (j == 7) ? zkff::G1() : g * rand_fr();
        gj[9] = gj[8];                                          // a repeated generator (P = Q inside a chain)
        std::vector<zkff::G1Affine> ga;
        zkff::batchToAffine(gj, ga);
        std::vector<zkff::Fr> sc(rows * m);
        for (auto &x : sc) x = rand_fr();
        for (uint32_t j = 0; j < 64; ++j) sc[j] = zkff::Fr((long long) j - 32);   // small and negative scalars, a zero
        sc[m + 8] = sc[m + 9] = rand_fr();
        g1a_t *dT; g1j_t *dJ; fp_t *dpre; fr_t *dsc, *dmag; g1j_t *part, *part2, *part3, *res;
        CK(hipMalloc((void **) &dT, (size_t) 32 * m * 96)); CK(hipMalloc((void **) &dJ, (size_t) 31 * m * 144)); CK(hipMalloc((void **) &dpre, (size_t) 31 * m * 48));
        CK(hipMalloc((void **) &dsc, rows * m * 32)); CK(hipMalloc((void **) &dmag, rows * m * 32));
        CK(hipMemcpy(dT, ga.data(), m * 96, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, sc.data(), rows * m * 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_wtab, dim3(m / 64), dim3(64), 0, 0, dT, dJ, dpre, m, 1u, 32u);
        hipLaunchKernelGGL(k_mags, dim3(8, rows), dim3(256), 0, 0, dmag, dsc, (uint64_t) m, m);
        const uint32_t cpt = 1, wsplit = 2, gx = (m / 64 / cpt) * wsplit, nparts = gx * 64;
        CK(hipMalloc((void **) &part, (size_t) rows * 8 * nparts * 144)); CK(hipMalloc((void **) &part2, (size_t) rows * 8 * 64 * 144));
        CK(hipMalloc((void **) &part3, (size_t) rows * 8 * 144)); CK(hipMalloc((void **) &res, rows * 144));
        auto run6 = [&]() {
            hipLaunchKernelGGL(k_acc, dim3(gx, rows * 8), dim3(64), 0, 0, part, dmag, (uint64_t) m, (const uint32_t *) nullptr, dT, m, m, cpt, wsplit, 0u);
            hipLaunchKernelGGL(k_tree, dim3(nparts / 64, rows * 8), dim3(512), 0, 0, part2, part, nparts, 64u);
            hipLaunchKernelGGL(k_tree, dim3(1, rows * 8), dim3(512), 0, 0, part3, part2, nparts / 64, nparts / 64);
            hipLaunchKernelGGL(k_horner, dim3(rows), dim3(128), 0, 0, res, part3);
        };
        run6();
        zkff::G1 got[2];
        CK(hipMemcpy(got, res, rows * 144, hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < rows; ++r) {
            const zkff::G1 want = zkff::msmCPU(&sc[r * m], ga.data(), m);
            const bool ok = got[r] == want;
            printf("MSM row %u (%u generators, full-width scalars, no byte table): %s\n", r, m, ok ? "== host Pippenger" : "MISMATCH");
            bad += !ok;
        }
        // timings: the new pipeline, kernel by kernel, against round 5's
        const float t_acc = time_ms([&]() { hipLaunchKernelGGL(k_acc, dim3(gx, rows * 8), dim3(64), 0, 0, part, dmag, (uint64_t) m, (const uint32_t *) nullptr, dT, m, m, cpt, wsplit, 0u); });
        const float t_t1 = time_ms([&]() { hipLaunchKernelGGL(k_tree, dim3(nparts / 64, rows * 8), dim3(512), 0, 0, part2, part, nparts, 64u); });
        const float t_t2 = time_ms([&]() { hipLaunchKernelGGL(k_tree, dim3(1, rows * 8), dim3(512), 0, 0, part3, part2, nparts / 64, nparts / 64); });
        const float t_h = time_ms([&]() { hipLaunchKernelGGL(k_horner, dim3(rows), dim3(128), 0, 0, res, part3); });
        const float t_all = time_ms(run6);
        printf("round 6: planes_acc %.3f ms (%u waves), tree 4096->64 %.3f ms, tree 64->1 %.3f ms, horner %.3f ms; back to back %.3f ms\n", t_acc, gx * rows * 8, t_t1, t_t2, t_h, t_all);
        const uint32_t cpt5 = 8, wsplit5 = 32, np5 = (m / 64 / cpt5) * wsplit5;
        g1j_t *p5;
        CK(hipMalloc((void **) &p5, (size_t) rows * 8 * np5 * 144));
        auto run5 = [&]() {
            hipLaunchKernelGGL(k_planes5, dim3(np5, rows * 8), dim3(64), 0, 0, p5, dmag, (uint64_t) m, (const uint32_t *) nullptr, dT, m, m, cpt5, wsplit5, 0u);
            hipLaunchKernelGGL(k_finish5, dim3(rows), dim3(512), 0, 0, res, p5, np5);
        };
        run5();
        CK(hipMemcpy(got, res, rows * 144, hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < rows; ++r) bad += !(got[r] == zkff::msmCPU(&sc[r * m], ga.data(), m));
        printf("round 5 (k_msm_planes + k_msm_finish): %.3f ms\n", time_ms(run5));
    }
    // ---- latencies ----
    {
        fp_t *co;
        CK(hipMalloc((void **) &co, (size_t) 8192 * 64 * 48));
        const int it = 2000;
        const float a = time_ms([&]() { hipLaunchKernelGGL(k_chain_sl, dim3(1), dim3(64), 0, 0, co, da, it); });
        const float b = time_ms([&]() { hipLaunchKernelGGL(k_chain_cl, dim3(1), dim3(64), 0, 0, co, da, it); });
        printf("dependent Fp products, a lone wave: one-lane form %.3f us each, row form %.3f us each\n", 1e3 * a / it, 1e3 * b / it);
        // saturated: 8 waves per SIMD
        const float c = time_ms([&]() { hipLaunchKernelGGL(k_chain_sl, dim3(8192), dim3(64), 0, 0, co, da, 200); });
        const float d = time_ms([&]() { hipLaunchKernelGGL(k_chain_cl, dim3(8192), dim3(64), 0, 0, co, da, 200); });
        printf("throughput at 8 waves per SIMD: one-lane form %.1f G products/s, row form %.1f G products/s\n", 8192.0 * 64 * 200 / c / 1e6, 8192.0 * 4 * 200 / d / 1e6);
        const float e = time_ms([&]() { hipLaunchKernelGGL(k_pchain_sl, dim3(1), dim3(64), 0, 0, po1, dp, 200); });
        const float f = time_ms([&]() { hipLaunchKernelGGL(k_pchain_cl, dim3(1), dim3(64), 0, 0, po2, dp, 200); });
        printf("dependent general point additions, a lone wave: one-lane form %.2f us each, row form %.2f us each\n", 1e3 * e / 200, 1e3 * f / 200);
    }
    printf(bad ? "FAILED: %d mismatches\n" : "all checks passed\n", bad);
    return bad ? 1 : 0;
}
