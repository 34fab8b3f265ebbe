// Kernel-level entry points of the C-ABI (host arrays in / out: what the parity tests and the micro-benchmarks call) -- included by sumcheck.hip.
#pragma once

// ------------------------------------------------------------------------------------------------
// kernel-level entry points (host arrays in / out)
// ------------------------------------------------------------------------------------------------
#define CHECK_CTX() ZK_CHECK_CTX()

template <int OP>
static int32_t binop(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 3 * n * 32);
    if (rc) return rc;
    fr_t *da = (fr_t *) ctx->scratch.p, *db = da + n, *dz = db + n;
    ZK_HIP(hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(PC_MISC, 0.0, k_fr_binop<OP>, dim3(grid_for(n)), dim3(ZK_BLOCK), dz, da, db, n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dz, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
extern "C" int32_t zk_k_fr_mul(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<0>(ctx, out, a, b, n); }
extern "C" int32_t zk_k_fr_add(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<1>(ctx, out, a, b, n); }
extern "C" int32_t zk_k_fr_sub(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<2>(ctx, out, a, b, n); }

extern "C" int32_t zk_k_eq_table(zk_ctx *ctx, uint64_t *out, int32_t n, const uint64_t *r0, const uint64_t *r1,
                                 const uint64_t alpha[4], const uint64_t beta[4]) {
    CHECK_CTX();
    if (n < 0 || n > ZK_MAX_VARS) return ZK_ERR_ARG;
    const uint64_t len = 1ull << n;
    int32_t rc = zk_scratch(ctx, len * 32);
    if (rc) return rc;
    rc = eq_table(ctx, (fr_t *) ctx->scratch.p, n, reinterpret_cast<const HFr *>(r0), H(alpha), reinterpret_cast<const HFr *>(r1),
                  H(beta), ~0ull, HFr::one());
    if (rc) return rc;
    ZK_HIP(hipMemcpyAsync(out, ctx->scratch.p, len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_k_phi_table(zk_ctx *ctx, uint64_t *out, const uint64_t *rx, const uint64_t scale[4], int32_t n,
                                  int32_t inverse) {
    CHECK_CTX();
    if (n < 1 || n > 20) return ZK_ERR_ARG;
    const uint64_t cnt = inverse ? 1ull << n : 1ull << (n - 1);
    int32_t rc = zk_scratch(ctx, cnt * 32);
    if (rc) return rc;
    if ((rc = phi_table(ctx, (fr_t *) ctx->scratch.p, reinterpret_cast<const HFr *>(rx), H(scale), n, inverse != 0))) return rc;
    ZK_HIP(hipMemcpyAsync(out, ctx->scratch.p, cnt * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_k_round_quadratic(zk_ctx *ctx, uint64_t *V, uint64_t *M, uint64_t n, const uint64_t r[4], int32_t first,
                                        uint64_t out_abc[12], uint64_t *n_out) {
    CHECK_CTX();
    if (n < 2 || (n & (n - 1)) || (!first && n < 4)) return ZK_ERR_ARG;
    int32_t rc = zk_scratch(ctx, 4 * n * 32);
    if (rc) return rc;
    fr_t *dV = (fr_t *) ctx->scratch.p, *dM = dV + n, *dV2 = dM + n, *dM2 = dV2 + n;
    ZK_HIP(hipMemcpyAsync(dV, V, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dM, M, n * 32, hipMemcpyHostToDevice, ctx->stream));
    // the PRODUCT kernel (k_round_quad2 as quad_round launches it for tables above the fine-grained limit), one table pair
    const uint64_t npairs = first ? n / 2 : n / 4;
    round2_args A;
    std::memset(&A, 0, sizeof(A));
    A.Vin[0] = dV; A.Min[0] = dM; A.Vout[0] = dV2; A.Mout[0] = dM2;
    A.n[0] = n;
    A.nl[0] = n;
    A.blocks[0] = std::min<uint32_t>((uint32_t) std::min<uint64_t>((npairs + 127) / 128, 1024), ctx->partial_blocks / 2);
    A.r = to_dev(H(r));
    A.first = first ? 1 : 0;
    A.partials = ctx->partials;
    A.counter = ctx->d_counter;
    A.slot = (host_slot *) ctx->d_slot;
    A.seq = ++ctx->slot_seq;
    if (first) ZK_LAUNCH(PC_ROUND_QUAD, 0.0, k_round_quad2<RQ_FIRST>, dim3(A.blocks[0]), dim3(ZK_BLOCK), A);
    else ZK_LAUNCH(PC_ROUND_QUAD, 0.0, k_round_quad2<RQ_FOLD_P1>, dim3(A.blocks[0]), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    if ((rc = wait_slot(ctx, A.seq))) return rc;
    for (int k = 0; k < 3; ++k) ctx->h_result[k] = ctx->h_slot->v[k];
    const uint64_t nn = first ? n : n / 2;
    if (!first) {
        ZK_HIP(hipMemcpy(V, dV2, nn * 32, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(M, dM2, nn * 32, hipMemcpyDeviceToHost));
    }
    HFr a = ctx->h_result[0], c = ctx->h_result[1], p1 = ctx->h_result[2];
    put(out_abc, a);
    put(out_abc + 4, p1 - a - c);
    put(out_abc + 8, c);
    *n_out = nn;
    return ZK_OK;
}

// ---- micro-benchmarks (device resident, HIP events on the context's stream) ----
template <class Launch>
static int32_t time_launches(zk_ctx *ctx, uint32_t iters, double *sec, Launch launch) {
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0));
    ZK_HIP(hipEventCreate(&e1));
    // warm-up: the clocks of a GPU that was idle while the host prepared the operands take milliseconds to come up (round 5: the 2^24-entry round
    // measured 0.40 ms over the first 20 launches of a process against 0.33 ms in steady state) -- run untimed launches for ~50 ms first
    const auto w0 = std::chrono::steady_clock::now();
    do {
        for (int i = 0; i < 4; ++i) launch();
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.05);
    ZK_HIP(hipEventRecord(e0, ctx->stream));
    for (uint32_t i = 0; i < iters; ++i) launch();
    ZK_HIP(hipEventRecord(e1, ctx->stream));
    ZK_HIP(hipEventSynchronize(e1));
    ZK_HIP(hipGetLastError());
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *sec = (double) ms * 1e-3 / iters;
    return ZK_OK;
}

extern "C" int32_t zk_bench_fr_mul(zk_ctx *ctx, uint64_t n_threads, uint32_t muls_per_thread, uint32_t iters, double *sec) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 2 * n_threads * 32);
    if (rc) return rc;
    int lg = 0;
    while ((2ull << lg) <= n_threads) ++lg;
    std::vector<HFr> r(lg, HFr(0x1234567LL));
    for (int i = 0; i < lg; ++i) r[i] = r[i] * HFr((long long) (i * 7919 + 3)) + HFr(0x9e3779b9LL);
    ZK_HIP(hipMemsetAsync(ctx->scratch.p, 0, 2 * n_threads * 32, ctx->stream));
    if ((rc = eq_table1(ctx, (fr_t *) ctx->scratch.p, lg, r.data(), HFr(77LL)))) return rc;
    const uint32_t blocks = (uint32_t) ((n_threads + ZK_BLOCK - 1) / ZK_BLOCK);
    return time_launches(ctx, iters, sec, [&] {
        ZK_LAUNCH_RAW(PC_MISC, 0.0, k_bench_fr_mul, dim3(blocks), dim3(ZK_BLOCK), (fr_t *) ctx->scratch.p, muls_per_thread, n_threads);
    });
}

extern "C" int32_t zk_bench_copy(zk_ctx *ctx, uint64_t bytes, uint32_t iters, double *sec) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 2 * bytes);
    if (rc) return rc;
    uint4 *src = (uint4 *) ctx->scratch.p, *dst = src + bytes / 16;
    ZK_HIP(hipMemsetAsync(src, 1, bytes, ctx->stream));
    return time_launches(ctx, iters, sec, [&] {
        ZK_LAUNCH_RAW(PC_MISC, 0.0, k_bench_copy, dim3(4096), dim3(ZK_BLOCK), dst, src, bytes / 16);
    });
}

// The dominant kernel of the FFT-conv configs in isolation: one non-first quadratic round (k_round_quad2, the product kernel) on two
// 2^log_n-entry tables of pseudo-random elements. Algorithmic bytes = 96 * 2^log_n.
extern "C" int32_t zk_bench_round_quadratic(zk_ctx *ctx, uint32_t log_n, uint32_t iters, double *sec_per_launch,
                                            double *algorithmic_bytes) {
    CHECK_CTX();
    if (log_n < 2 || log_n > 28) return ZK_ERR_ARG;
    const uint64_t n = 1ull << log_n;
    int32_t rc = zk_scratch(ctx, 3 * n * 32);
    if (rc) return rc;
    fr_t *dV = (fr_t *) ctx->scratch.p, *dM = dV + n, *dO = dM + n;
    // fill with field elements: eq tables of a fixed point are as good as random for timing purposes
    std::vector<HFr> r(log_n);
    zkff::Xoshiro g;
    g.seed(0x5EED0002ULL);
    for (auto &x : r) {
        uint64_t t[4] = {g.next(), g.next(), g.next(), g.next() & 0x3fffffffffffffffULL};
        x = HFr(zkff::MontField<zkff::FrParams>::fromCanonical(t));
    }
    if ((rc = eq_table1(ctx, dV, (int) log_n, r.data(), HFr(7LL)))) return rc;
    if ((rc = eq_table1(ctx, dM, (int) log_n, r.data(), HFr(11LL)))) return rc;
    // the product kernel exactly as quad_round launches it for one large table pair (fold + round sums + grid finish + host slot)
    round2_args A;
    std::memset(&A, 0, sizeof(A));
    A.Vin[0] = dV; A.Min[0] = dM; A.Vout[0] = dO; A.Mout[0] = dO + n / 2;
    A.n[0] = n;
    A.nl[0] = n;
    A.blocks[0] = std::min<uint32_t>((uint32_t) std::min<uint64_t>((n / 4 + 127) / 128, 1024), ctx->partial_blocks / 2);
    A.r = to_dev(r[0]);
    A.skip_p1 = 1;                        // as every round but the first of a phase runs: b comes from the running claim
    A.partials = ctx->partials;
    A.counter = ctx->d_counter;
    A.slot = (host_slot *) ctx->d_slot;
    *algorithmic_bytes = 96.0 * (double) n;
    rc = time_launches(ctx, iters, sec_per_launch, [&] {
        A.seq = ++ctx->slot_seq;
        ZK_LAUNCH_RAW(PC_ROUND_QUAD, 96.0 * (double) n, k_round_quad2<RQ_FOLD>, dim3(A.blocks[0]), dim3(ZK_BLOCK), A);
    });
    if (rc) return rc;
    return wait_slot(ctx, ctx->slot_seq);
}

