// libzkcnn_hip.so: the verifier's gate-walking work on the GPU -- the wiring predicates of a layer and the layer-0 check over
// the resident gate lists (reference src/verifier.cpp:36-116, 304-325). Optional: the host verifier computes the same values by itself.
#include <algorithm>
#include <cstring>
#include "ctx.hpp"
#include "verifier_kernels.cuh"

static inline int32_t eq_table(zk_ctx *ctx, fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b, uint64_t tail_start, const HFr &tail_scale) {
    return zk_eq_table_dev(ctx, out, n, r0, a, r1, b, tail_start, tail_scale, ~0ull);
}
static inline int32_t eq1(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init) { return zk_eq_table1_dev(ctx, out, n, r, init); }
static inline int32_t eq_table1(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init) { return zk_eq_table1_dev(ctx, out, n, r, init); }
static inline int32_t phi_table(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse) { return zk_phi_table_dev(ctx, out, rx, scale, n, inverse); }

// ------------------------------------------------------------------------------------------------
// verifier side: wiring predicates over the resident gate lists (reference src/verifier.cpp:36-116, 304-325)
// ------------------------------------------------------------------------------------------------
static int32_t verifier_buffers(zk_ctx *ctx) {
    if (ctx->v_bg) return ZK_OK;
    int32_t rc;
    const uint64_t cap_uv = std::max(std::max(ctx->beta_u_cap, ctx->beta_g_cap), ctx->sz.sub);
    const uint64_t cap_g = std::max<uint64_t>(ctx->beta_g_cap, ctx->L[0].val_len);      // the layer-0 check needs eq over all of layer 0
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->v_bg, cap_g * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->v_bu, cap_uv * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->v_bv, cap_uv * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->v_gs, ctx->beta_gs_cap * 32)))
        return rc;
    return ZK_OK;
}

extern "C" int32_t zk_verifier_predicates(zk_ctx *ctx, int32_t layer, const uint64_t *r_0, const uint64_t *r_1, const uint64_t alpha[4],
                                          const uint64_t beta[4], const uint64_t relu_rou[4], const uint64_t *r_u, const uint64_t *r_v,
                                          const uint64_t *r_u2, const uint64_t *r_v2, uint64_t uni[8], uint64_t bin[12]) {
    ZK_CHECK_READY();
    if (layer < 1 || layer >= (int) ctx->L.size()) return ZK_ERR_ARG;
    int32_t rc;
    if ((rc = verifier_buffers(ctx))) return rc;
    const dev_layer &cur = ctx->L[layer];
    const zk_layer_desc &d = cur.d;
    const HFr *R0 = reinterpret_cast<const HFr *>(r_0), *R1 = reinterpret_cast<const HFr *>(r_1), *RU = reinterpret_cast<const HFr *>(r_u),
              *RV = reinterpret_cast<const HFr *>(r_v), *RU2 = reinterpret_cast<const HFr *>(r_u2), *RV2 = reinterpret_cast<const HFr *>(r_v2);
    const HFr al = H(alpha), be = H(beta), scale = H(d.scale);
    const int bl = d.bit_length, fft_bl = d.fft_bit_length, fft_blh = fft_bl - 1;
    fr_t *res = ctx->d_result + 16;                       // [0,1] uni, [2,3] bin list 0, [4,5] bin list 1, [6] table dot
    ZK_HIP(hipMemsetAsync(res, 0, 8 * 32, ctx->stream));
    const bool xf = d.ty == ZK_FFT || d.ty == ZK_IFFT, dot = d.ty == ZK_DOT_PROD;
    if (xf) {
        if ((rc = phi_table(ctx, ctx->v_gs, R0, scale, fft_bl, d.ty == ZK_IFFT))) return rc;
        if ((rc = eq1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
        const uint64_t n = 1ull << d.max_bl_u;
        const uint32_t g = std::min<uint32_t>(grid_for(n, 1024), ctx->partial_blocks);
        ZK_LAUNCH(PC_GATE_SUM, 0.0, k_dot_indexed, dim3(g), dim3(ZK_BLOCK), ctx->partials, ctx->v_gs, (const uint32_t *) nullptr, ctx->v_bu, n);
        ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<1>, dim3(1), dim3(ZK_BLOCK), res + 6, ctx->partials, g, 0);
    } else if (d.ty == ZK_PADDING) {
        if (!RU2) return ZK_ERR_ARG;
        if ((rc = eq_table(ctx, ctx->v_bv, bl - fft_blh, RU2 + fft_bl, al, RV2, be, ~0ull, HFr::one()))) return rc;
        if ((rc = eq1(ctx, ctx->v_gs, fft_blh, R0, HFr::one()))) return rc;
        ZK_LAUNCH(PC_EQ, 0.0, k_outer_expand, dim3(grid_for(cur.val_len)), dim3(ZK_BLOCK), ctx->v_bg, ctx->v_bv, ctx->v_gs, fft_blh, cur.val_len);
        if ((rc = eq1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
    } else if (dot) {
        if (!RU2) return ZK_ERR_ARG;
        const int cnt_bl = bl - fft_bl, cnt_bl2 = d.max_bl_u - fft_bl;
        if ((rc = eq1(ctx, ctx->v_bg, cnt_bl, RU2 + fft_bl - 1, al))) return rc;
        HFr same = HFr::one();                            // eq(r_0, r_u) on the frequency bits
        for (int j = 0; j < fft_bl; ++j) same = same * (R0[j] * RU[j] + (HFr::one() - R0[j]) * (HFr::one() - RU[j]));
        if ((rc = eq1(ctx, ctx->v_bu, cnt_bl2, RU + fft_bl, same))) return rc;
    } else {
        const bool tail = d.zero_start_id < d.size;
        if ((rc = eq_table(ctx, ctx->v_bg, bl, R0, al * scale, R1, be * scale, tail ? d.zero_start_id : ~0ull, tail ? H(relu_rou) : HFr::one())))
            return rc;
        if ((rc = eq1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
    }
    if (!xf && cur.n_uni2) {
        gate_args A;
        std::memset(&A, 0, sizeof(A));
        A.recs = cur.uni2; A.n = cur.n_uni2;
        A.beta_g = ctx->v_bg; A.beta_u = ctx->v_bu; A.two_mul = ctx->two_mul; A.phase = 2;
        const uint32_t g = std::min<uint32_t>(grid_for(cur.n_uni2, 1024), ctx->partial_blocks);
        ZK_LAUNCH(PC_GATE_SUM, 0.0, k_gate_sum2, dim3(g), dim3(ZK_BLOCK), ctx->partials, A);
        ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<2>, dim3(1), dim3(ZK_BLOCK), res, ctx->partials, g, 0);
    }
    HFr bv0 = HFr::one();
    if (d.need_phase2) {
        if ((rc = eq1(ctx, ctx->v_bv, d.max_bl_v, RV, HFr::one()))) return rc;
        for (int j = 0; j < d.max_bl_v; ++j) bv0 = bv0 * (HFr::one() - RV[j]);          // beta_v[0]
        for (int b = 0; b < 2; ++b) {
            if (!cur.n_p2[b]) continue;
            const uint32_t g = std::min<uint32_t>(grid_for(cur.n_p2[b], 1024), ctx->partial_blocks);
            ZK_LAUNCH(PC_GATE_SUM, 0.0, k_pred_bin, dim3(g), dim3(ZK_BLOCK), ctx->partials, cur.p2[b], cur.n_p2[b], ctx->v_bg, ctx->v_bu,
                      ctx->v_bv, ctx->two_mul, dot ? 0 : 1);
            ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<2>, dim3(1), dim3(ZK_BLOCK), res + 2 + 2 * b, ctx->partials, g, 0);
        }
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(ctx->h_result + 16, res, 8 * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    const HFr *h = ctx->h_result + 16;
    HFr u0 = h[0], u1 = xf ? h[6] : h[1];
    if (d.need_phase2) { u0 = u0 * bv0; u1 = u1 * bv0; }
    put(uni, u0);
    put(uni + 4, u1);
    put(bin, h[2]);                 // l = 0: u and v in layer 0
    put(bin + 4, h[5]);             // l = 1: both in the previous layer
    put(bin + 8, h[3]);             // l = 2: u in the previous layer, v in layer 0
    if (!h[4].isZero()) { ctx->err = "bin gate with u in layer 0 and v in the previous layer"; return ZK_ERR_STATE; }
    return ZK_OK;
}

// g(r_u0) of the layer-0 check: sum over layers of sum_j eq(r_u0, ori_id[j]) * sig * eq(r_{u,v}[layer], j)  (reference src/verifier.cpp:304-325)
extern "C" int32_t zk_verifier_input_predicate(zk_ctx *ctx, const uint64_t *r_u0, const uint64_t *const *r_u, const uint64_t *const *r_v,
                                               const uint64_t *sig_u, const uint64_t *sig_v, uint32_t n, uint64_t out[4]) {
    ZK_CHECK_READY();
    if (n + 1 != ctx->L.size()) return ZK_ERR_ARG;
    int32_t rc;
    if ((rc = verifier_buffers(ctx))) return rc;
    fr_t *res = ctx->d_result + 16;
    ZK_HIP(hipMemsetAsync(res, 0, 32, ctx->stream));
    if ((rc = eq1(ctx, ctx->v_bg, ctx->L[0].d.bit_length, reinterpret_cast<const HFr *>(r_u0), HFr::one()))) return rc;
    for (size_t i = 1; i < ctx->L.size(); ++i) {
        const dev_layer &Li = ctx->L[i];
        for (int side = 0; side < 2; ++side) {
            const int bl = side ? Li.d.bit_length_v[0] : Li.d.bit_length_u[0];
            const uint32_t cnt = side ? Li.d.size_v[0] : Li.d.size_u[0];
            const uint64_t *pt = side ? r_v[i] : r_u[i];
            if (bl < 0 || !cnt) continue;
            if (!pt) return ZK_ERR_ARG;
            if ((rc = eq1(ctx, ctx->v_bu, bl, reinterpret_cast<const HFr *>(pt), H((side ? sig_v : sig_u) + 4 * (i - 1))))) return rc;
            const uint32_t g = std::min<uint32_t>(grid_for(cnt, 1024), ctx->partial_blocks);
            ZK_LAUNCH(PC_LIU, 0.0, k_dot_indexed, dim3(g), dim3(ZK_BLOCK), ctx->partials, ctx->v_bg, side ? Li.ori_v : Li.ori_u, ctx->v_bu, (uint64_t) cnt);
            ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<1>, dim3(1), dim3(ZK_BLOCK), res, ctx->partials, g, 1);
        }
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(ctx->h_result + 16, res, 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    put(out, ctx->h_result[16]);
    return ZK_OK;
}

