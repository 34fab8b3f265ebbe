// Structured gate sums of a direct-convolution layer (NCONV; reference src/neuralNetwork.cpp:240-275 emits its gates, reference
// src/prover.cpp:224-233 / 297-305 sums them gate by gate: channel_out * channel_in * m^2 * positions products per phase).
// With power-of-two channel counts and picture sides the gate indices are bit fields, so the eq tables of both phases factor:
//     beta_g[(p, co, X, Y)] = sum_k P_k[p] A_k[co] S_k[X, Y]          (k: the one or two points the layer's claims sit at)
//     beta_u[(p, ci, tx, ty)] = Pu[p] C[ci] D[tx, ty]
// and the sums over the gates (g = (p, co, X, Y), u = (p, ci, tx, ty), v = weight (co, ci, dx, dy), X = (tx - dx + pad) >> stride) become
//     phase 1:  M[u]  = sum_k P_k[p] sum_{dx,dy} S_k[X, Y] WA_k[ci, dx, dy],      WA_k[ci, dx, dy] = sum_co A_k[co] W[co, ci, dx, dy]
//     phase 2:  M[v]  = V_u C[ci] sum_k A_k[co] E_k[dx, dy],     E_k[dx, dy] = (sum_p P_k[p] Pu[p]) sum_{tx,ty} S_k[X, Y] D[tx, ty]
// -- the SAME field elements as the gate-by-gate sums (distributivity, nothing is rounded), for channel_out * channel_in * m^2 products
// instead of that times the number of positions. The upload checks the layer's gate list against this pattern record by record
// before it enables the path (zk_upload_circuit_hinted); otherwise the generic segmented reduction runs.
#pragma once
#include "types.cuh"

// part[(chunk * K + k) * len + o] = sum over the chunk's co of A_k[co] * W[co * len + o];  len = CI * m^2; grid (ceil(len / 256), chunks)
struct k_conv_wa_f {
    fr_t *part; const fr_t *W; const fr_t *tabs; uint32_t len, CO, per; int K;
    __device__ __forceinline__ void operator()() const {
    const uint32_t o = blockIdx.x * ZK_BLOCK + threadIdx.x;
    if (o >= len) return;
    const uint32_t c0 = blockIdx.y * per, c1 = min(CO, c0 + per);
    const fr_t *A0 = tabs + CT_A0 * CONV_TAB_STRIDE, *A1 = tabs + CT_A1 * CONV_TAB_STRIDE;
    fr_t acc0 = fr_zero(), acc1 = fr_zero();
    for (uint32_t co = c0; co < c1; ++co) {
        const fr_t w = fr_load(W + (size_t) co * len + o);
        acc0 = fr_add(acc0, fr_mul(w, fr_load(A0 + co)));
        if (K > 1) acc1 = fr_add(acc1, fr_mul(w, fr_load(A1 + co)));
    }
    fr_store(part + ((size_t) blockIdx.y * K) * len + o, acc0);
    if (K > 1) fr_store(part + ((size_t) blockIdx.y * K + 1) * len + o, acc1);
    }
};

// window origin -> output coordinate; false if (t - d) is not the origin of a window of the layer
__device__ __forceinline__ bool conv_out_coord(uint32_t t, uint32_t d, const conv_desc &c, uint32_t n_out, uint32_t &X) {
    const int32_t xo = (int32_t) t - (int32_t) d + (int32_t) c.pad;
    if (xo < 0 || (xo & ((1 << c.ls) - 1))) return false;
    X = (uint32_t) xo >> c.ls;
    return X < n_out;
}

// phase 1: M[u] for every entry of the previous layer's table (u >= the layer's size: 0)
struct k_conv_m1_f {
    fr_t *M; const fr_t *WA; const fr_t *tabs; conv_desc c; int K; uint64_t len;
    __device__ __forceinline__ void operator()() const {
    const uint64_t n_u = (uint64_t) c.pp * c.CI * c.nxi * c.nyi;
    const uint32_t wlen = c.CI * c.m * c.m;
    for (uint64_t u = blockIdx.x * (uint64_t) ZK_BLOCK + threadIdx.x; u < len; u += (uint64_t) gridDim.x * ZK_BLOCK) {
        if (u >= n_u) { fr_store(M + u, fr_zero()); continue; }
        const uint32_t ty = (uint32_t) u & (c.nyi - 1), tx = (uint32_t) (u >> c.by_i) & (c.nxi - 1);
        const uint32_t ci = (uint32_t) (u >> (c.by_i + c.bx_i)) & (c.CI - 1), p = (uint32_t) (u >> (c.by_i + c.bx_i + c.bc_i));
        fr_t acc = fr_zero();
        for (int k = 0; k < K; ++k) {
            const fr_t *S = tabs + (k ? CT_S1 : CT_S0) * CONV_TAB_STRIDE;
            const fr_t *wa = WA + (size_t) k * wlen + (size_t) ci * c.m * c.m;
            fr_t inner = fr_zero();
            for (uint32_t dx = 0; dx < c.m; ++dx) {
                uint32_t X, Y;
                if (!conv_out_coord(tx, dx, c, c.nxo, X)) continue;
                for (uint32_t dy = 0; dy < c.m; ++dy) {
                    if (!conv_out_coord(ty, dy, c, c.nyo, Y)) continue;
                    inner = fr_add(inner, fr_mul(fr_load(S + ((X << c.by_o) | Y)), fr_load(wa + dx * c.m + dy)));
                }
            }
            acc = fr_add(acc, fr_mul(inner, fr_load(tabs + (k ? CT_P1 : CT_P0) * CONV_TAB_STRIDE + p)));
        }
        fr_store(M + u, acc);
    }
    }
};

// phase 2, step 1: E[k * m^2 + dx * m + dy]; grid (m^2, K)
struct k_conv_e_f {
    fr_t *E; const fr_t *tabs; conv_desc c; int K;
    __device__ __forceinline__ void operator()() const {
    __shared__ fr_t smem[ZK_BLOCK / 64];
    const int k = blockIdx.y;
    if (k >= K || blockIdx.x >= c.m * c.m) return;               // (rows of a fused grid beyond this lane's own)
    const uint32_t dx = blockIdx.x / c.m, dy = blockIdx.x % c.m;
    const fr_t *S = tabs + (k ? CT_S1 : CT_S0) * CONV_TAB_STRIDE, *D = tabs + CT_D * CONV_TAB_STRIDE;
    fr_t acc[1] = {fr_zero()};
    for (uint32_t pos = threadIdx.x; pos < c.nxi * c.nyi; pos += ZK_BLOCK) {
        const uint32_t ty = pos & (c.nyi - 1), tx = pos >> c.by_i;
        uint32_t X, Y;
        if (!conv_out_coord(tx, dx, c, c.nxo, X) || !conv_out_coord(ty, dy, c, c.nyo, Y)) continue;
        acc[0] = fr_add(acc[0], fr_mul(fr_load(S + ((X << c.by_o) | Y)), fr_load(D + pos)));
    }
    fr_block_sum<1>(acc, smem);
    if (threadIdx.x == 0) {
        const fr_t *P = tabs + (k ? CT_P1 : CT_P0) * CONV_TAB_STRIDE, *Pu = tabs + CT_PU * CONV_TAB_STRIDE;
        fr_t g = fr_zero();
        for (uint32_t p = 0; p < c.pp; ++p) g = fr_add(g, fr_mul(fr_load(P + p), fr_load(Pu + p)));
        fr_store(E + (size_t) k * c.m * c.m + blockIdx.x, fr_mul(acc[0], g));
    }
    }
};
// phase 2, step 2: AE[co * m^2 + d] = sum_k A_k[co] E_k[d]  (CO * m^2 entries), then CV[ci] = Vu * C[ci]  (CI entries) behind them
struct k_conv_ae_f {
    fr_t *AE; const fr_t *E; const fr_t *tabs; conv_desc c; int K; fr_t Vu;
    __device__ __forceinline__ void operator()() const {
    const uint32_t mm = c.m * c.m, n_ae = c.CO * mm;
    const uint32_t i = blockIdx.x * ZK_BLOCK + threadIdx.x;
    if (i < n_ae) {
        const uint32_t co = i / mm, d = i % mm;
        fr_t v = fr_mul(fr_load(tabs + CT_A0 * CONV_TAB_STRIDE + co), fr_load(E + d));
        if (K > 1) v = fr_add(v, fr_mul(fr_load(tabs + CT_A1 * CONV_TAB_STRIDE + co), fr_load(E + mm + d)));
        fr_store(AE + i, v);
    } else if (i < n_ae + c.CI) {
        fr_store(AE + i, fr_mul(Vu, fr_load(tabs + CT_C * CONV_TAB_STRIDE + (i - n_ae))));
    }
    }
};
// phase 2, step 3: M[v'] = CV[ci] * AE[co, d] for the weight v' stands for (ori_v: subset number -> raw layer-0 index); v' >= n_v: 0
struct k_conv_m2_f {
    fr_t *M; const uint32_t *ori_v; const fr_t *AE; conv_desc c; uint32_t n_v; uint64_t len;
    __device__ __forceinline__ void operator()() const {
    const uint32_t mm = c.m * c.m;
    const fr_t *CV = AE + (size_t) c.CO * mm;
    for (uint64_t v = blockIdx.x * (uint64_t) ZK_BLOCK + threadIdx.x; v < len; v += (uint64_t) gridDim.x * ZK_BLOCK) {
        if (v >= n_v) { fr_store(M + v, fr_zero()); continue; }
        const uint32_t raw = ori_v[v] - c.wstart, d = raw % mm, q = raw / mm;
        fr_store(M + v, fr_mul(fr_load(CV + (q & (c.CI - 1))), fr_load(AE + (size_t) (q >> c.bc_i) * mm + d)));
    }
    }
};

