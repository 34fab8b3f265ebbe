// libzkcnn_hip.so: witness generation on the GPU. The generator's three heavy loops behind host arrays (calcNormalLayer,
// calcFFTLayer, calcDotProdLayer: reference src/neuralNetwork.cpp:918-965), and the resident witness PROGRAM that re-evaluates a whole
// witness in HBM for the next picture (zk_witness_program_upload / zk_witness_rerun).
#include <algorithm>
#include <cstring>
#include "ctx.hpp"
#include "witness_kernels.cuh"

// ---- witness kernels of the FFT convolution (host arrays in / out) ----
extern "C" int32_t zk_witness_ntt(zk_ctx *ctx, uint64_t *dst, const uint64_t *src, int32_t logn, int32_t inverse, uint64_t count) {
    ZK_CHECK_CTX();
    if (logn < 1 || logn > 12 || !count) return ZK_ERR_ARG;        // 2^12 x 32 B = 128 KiB is what fits in LDS
    const uint32_t len = 1u << logn, in_len = inverse ? len : len / 2, out_len = inverse ? len / 2 : len;
    fr_t *pw = zk_powers_of_root(ctx, logn, inverse != 0);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    const size_t in_bytes = (size_t) count * in_len * 32, out_bytes = (size_t) count * out_len * 32;
    int32_t rc = zk_scratch(ctx, in_bytes + out_bytes);
    if (rc) return rc;
    fr_t *d_in = (fr_t *) ctx->scratch.p, *d_out = d_in + (size_t) count * in_len;
    ZK_HIP(hipMemcpyAsync(d_in, src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HFr ilen;
    HFr::inv(ilen, HFr((unsigned long long) len));
    const size_t lds = (size_t) len * 32;
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_batch), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const uint32_t threads = std::min<uint32_t>(1024, std::max<uint32_t>(64, len / 2));
    for (uint64_t b0 = 0; b0 < count; b0 += 1u << 30) {
        const uint32_t nb = (uint32_t) std::min<uint64_t>(1u << 30, count - b0);
        prof_begin(ctx, PC_MISC, 64.0 * len * nb);
        hipLaunchKernelGGL(k_ntt_batch, dim3(nb), dim3(threads), lds, ctx->stream, d_out + b0 * out_len, d_in + b0 * in_len, pw, logn, in_len,
                           out_len, to_dev(ilen), inverse ? 1 : 0);
        prof_end(ctx, PC_MISC);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(dst, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_witness_dotprod(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const uint64_t *F, uint64_t n_in,
                                      const zk_bin_gate *gates, uint64_t n_gates, int32_t fft_bl) {
    ZK_CHECK_CTX();
    if (fft_bl < 1 || fft_bl > 20 || !n_out || !n_in) return ZK_ERR_ARG;
    std::vector<gate_rec> recs(n_gates);
    for (uint64_t k = 0; k < n_gates; ++k) {
        if (gates[k].g >= n_out || gates[k].u >= n_in || gates[k].v >= n_in) { ctx->err = "dot-prod gate out of range"; return ZK_ERR_ARG; }
        gate_rec r = {gates[k].g, gates[k].u, gates[k].v, 0};      // key = u, aux = v for this kernel
        recs[k] = r;
    }
    // CSR by output vector g (stable: order inside a row does not matter, the sum is exact)
    std::vector<uint32_t> ptr(n_out + 1, 0);
    for (const gate_rec &r : recs) ++ptr[r.g + 1];
    for (uint64_t g = 0; g < n_out; ++g) ptr[g + 1] += ptr[g];
    std::vector<gate_rec> sorted(n_gates);
    {
        std::vector<uint32_t> pos(ptr.begin(), ptr.end() - 1);
        for (const gate_rec &r : recs) sorted[pos[r.g]++] = r;
    }
    const size_t len = (size_t) 1 << fft_bl;
    const size_t bytes = (n_in + n_out) * len * 32 + n_gates * sizeof(gate_rec) + (n_out + 1) * 4 + 64;
    int32_t rc = zk_scratch(ctx, bytes);
    if (rc) return rc;
    fr_t *dF = (fr_t *) ctx->scratch.p, *dO = dF + n_in * len;
    gate_rec *dR = (gate_rec *) (dO + n_out * len);
    uint32_t *dP = (uint32_t *) (dR + n_gates);
    ZK_HIP(hipMemcpyAsync(dF, F, n_in * len * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dR, sorted.data(), n_gates * sizeof(gate_rec), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dP, ptr.data(), (n_out + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    for (uint64_t g0 = 0; g0 < n_out; g0 += 32768) {
        const uint32_t ng = (uint32_t) std::min<uint64_t>(32768, n_out - g0);
        dim3 grid((uint32_t) ((len + ZK_BLOCK - 1) / ZK_BLOCK), ng);
        ZK_LAUNCH(PC_DOT, 0.0, k_dot_witness, grid, dim3(ZK_BLOCK), dO + g0 * len, dF, dR, dP + g0, fft_bl);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dO, n_out * len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// ---- witness of a generic layer: every gate evaluated on the GPU (reference src/neuralNetwork.cpp:918-935) ----
static int32_t grow_buf(zk_ctx *ctx, dev_buf &b, size_t bytes, size_t keep = 0) {
    if (b.bytes >= bytes) return ZK_OK;
    size_t want = std::max(bytes, b.bytes + b.bytes / 2);
    void *np = nullptr;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (hipMalloc(&np, want) != hipSuccess) {
        (void) hipGetLastError();
        want = bytes;
        ZK_HIP(hipMalloc(&np, want));
    }
    if (keep && b.p) {
        ZK_HIP(hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (b.p) ZK_HIP(hipFree(b.p));
    b.p = np;
    b.bytes = want;
    return ZK_OK;
}

extern "C" int32_t zk_witness_release(zk_ctx *ctx) {
    ZK_CHECK_CTX();
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->w_val0.p) { ZK_HIP(hipFree(ctx->w_val0.p)); ctx->w_val0 = dev_buf(); }
    for (dev_buf &b : ctx->w_stage) if (b.p) { ZK_HIP(hipFree(b.p)); b = dev_buf(); }
    ctx->w_val0_len = 0;
    return ZK_OK;
}

extern "C" int32_t zk_witness_input(zk_ctx *ctx, uint64_t offset, const uint64_t *values, uint64_t n) {
    ZK_CHECK_CTX();
    if (offset == 0) ctx->w_val0_len = 0;                            // a new layer 0 starts
    if (offset > ctx->w_val0_len) return ZK_ERR_ARG;                 // the copy has no holes
    int32_t rc = grow_buf(ctx, ctx->w_val0, (offset + n) * 32, ctx->w_val0_len * 32);
    if (rc) return rc;
    if (n) ZK_HIP(hipMemcpyAsync((fr_t *) ctx->w_val0.p + offset, values, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ctx->w_val0_len = std::max<uint64_t>(ctx->w_val0_len, offset + n);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// true when equal g are adjacent; range-checks the list on the way
template <class G, class Check>
static int grouped_by_output(const G *gates, uint64_t n, uint64_t n_out, std::vector<uint8_t> &seen, Check in_range) {
    std::fill(seen.begin(), seen.end(), 0);
    bool grouped = true;
    uint32_t prev = 0xffffffffu;
    for (uint64_t k = 0; k < n; ++k) {
        const G &gt = gates[k];
        if (gt.g >= n_out || !in_range(gt)) return -1;
        if (gt.g != prev) {
            if (seen[gt.g]) grouped = false;
            seen[gt.g] = 1;
            prev = gt.g;
        }
    }
    return grouped ? 1 : 0;
}
template <class G>
static void regroup(std::vector<G> &dst, const G *gates, uint64_t n, uint64_t n_out) {
    std::vector<uint64_t> pos(n_out + 1, 0);
    for (uint64_t k = 0; k < n; ++k) ++pos[gates[k].g + 1];
    for (uint64_t g = 0; g < n_out; ++g) pos[g + 1] += pos[g];
    dst.resize(n);
    for (uint64_t k = 0; k < n; ++k) dst[pos[gates[k].g]++] = gates[k];
}

extern "C" int32_t zk_witness_gates(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const zk_uni_gate *uni, uint64_t n_uni,
                                    const zk_bin_gate *bin, uint64_t n_bin, const uint64_t *prev, uint64_t n_prev,
                                    const uint64_t *two_mul, uint32_t n_two_mul, const uint64_t scale[4]) {
    ZK_CHECK_CTX();
    if (!n_out || n_out > 0xfffffffeull || !n_two_mul) return ZK_ERR_ARG;
    static_assert(sizeof(uni_gate_dev) == sizeof(zk_uni_gate) && sizeof(bin_gate_dev) == sizeof(zk_bin_gate), "gate layout");
    const uint64_t n0 = ctx->w_val0_len;
    if (!prev) n_prev = n0;                                          // layer 1: the previous layer IS layer 0
    std::vector<uint8_t> seen(n_out);
    std::vector<zk_uni_gate> uni_sorted;
    std::vector<zk_bin_gate> bin_sorted;
    int g1 = grouped_by_output(uni, n_uni, n_out, seen, [&](const zk_uni_gate &gt) {
        return gt.sc < n_two_mul && gt.u < (gt.lu ? n_prev : n0); });
    int g2 = g1 < 0 ? -1 : grouped_by_output(bin, n_bin, n_out, seen, [&](const zk_bin_gate &gt) {
        return gt.sc < n_two_mul && gt.l <= 2 && gt.u < (gt.l == 0 ? n0 : n_prev) && gt.v < ((gt.l & 1) ? n_prev : n0); });
    if (g1 < 0 || g2 < 0) { ctx->err = "witness gate operand out of range"; return ZK_ERR_ARG; }
    if (!g1) { regroup(uni_sorted, uni, n_uni, n_out); uni = uni_sorted.data(); }
    if (!g2) { regroup(bin_sorted, bin, n_bin, n_out); bin = bin_sorted.data(); }

    const uint64_t blocks_u = (n_uni + ZK_BLOCK - 1) / ZK_BLOCK, blocks_b = (n_bin + ZK_BLOCK - 1) / ZK_BLOCK;
    const uint64_t slots = 2 * std::max<uint64_t>(std::max(blocks_u, blocks_b), 1);
    int32_t rc;
    dev_buf &bG = ctx->w_stage[0], &bP = ctx->w_stage[1], &bO = ctx->w_stage[2], &bC = ctx->w_stage[3], &bT = ctx->w_stage[4];
    if ((rc = grow_buf(ctx, bG, std::max<size_t>(n_uni * sizeof(zk_uni_gate), n_bin * sizeof(zk_bin_gate)) + 16)) ||
        (rc = grow_buf(ctx, bP, (prev ? std::max<uint64_t>(n_prev, 1) : 1) * 32)) || (rc = grow_buf(ctx, bO, 3 * n_out * 32)) ||
        (rc = grow_buf(ctx, bC, slots * (32 + 4))) || (rc = grow_buf(ctx, bT, (size_t) n_two_mul * 32)))
        return rc;
    fr_t *dA = (fr_t *) bO.p, *dB = dA + n_out, *dO = dB + n_out;
    fr_t *carry_val = (fr_t *) bC.p;
    uint32_t *carry_key = (uint32_t *) (carry_val + slots);
    const fr_t *v0 = (const fr_t *) ctx->w_val0.p, *vp = prev ? (const fr_t *) bP.p : v0, *tm = (const fr_t *) bT.p;
    if (prev && n_prev) ZK_HIP(hipMemcpyAsync(bP.p, prev, n_prev * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(bT.p, two_mul, (size_t) n_two_mul * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemsetAsync(dA, 0, 2 * n_out * 32, ctx->stream));
    if (n_uni) {
        ZK_HIP(hipMemcpyAsync(bG.p, uni, n_uni * sizeof(zk_uni_gate), hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(PC_GATE, 44.0 * (double) n_uni, k_eval_uni, dim3((uint32_t) blocks_u), dim3(ZK_BLOCK), dA, carry_key, carry_val,
                  (const uni_gate_dev *) bG.p, n_uni, v0, vp, tm);
        ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks_u)), dim3(ZK_BLOCK), dA, carry_key, carry_val, 2 * blocks_u);
    }
    if (n_bin) {
        ZK_HIP(hipMemcpyAsync(bG.p, bin, n_bin * sizeof(zk_bin_gate), hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(PC_GATE, 80.0 * (double) n_bin, k_eval_bin, dim3((uint32_t) blocks_b), dim3(ZK_BLOCK), dB, carry_key, carry_val,
                  (const bin_gate_dev *) bG.p, n_bin, v0, vp, tm);
        ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks_b)), dim3(ZK_BLOCK), dB, carry_key, carry_val, 2 * blocks_b);
    }
    const HFr sc = H(scale);
    ZK_LAUNCH(PC_MISC, 0.0, k_eval_combine, dim3(grid_for(n_out)), dim3(ZK_BLOCK), dO, dA, dB, to_dev(sc), sc == HFr::one() ? 0 : 1, n_out);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dO, n_out * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// channel_in chunks of k_conv_eval: about 2^20 threads per launch, so that layers with few outputs (2 x 2 pictures, 512 channels) still fill the GPU
static uint32_t conv_eval_chunks(const conv_desc &c, uint64_t n_out) {
    const uint64_t want = std::max<uint64_t>(1, ((1ull << 20) + n_out - 1) / n_out);
    const uint32_t per = (uint32_t) std::max<uint64_t>(1, (c.CI + want - 1) / want);
    return (c.CI + per - 1) / per;
}

// ------------------------------------------------------------------------------------------------
// resident witness program: the next picture without the host round trip of the layer values
// ------------------------------------------------------------------------------------------------
// The program has a static part -- the validated steps, the operations and windows, the gate lists grouped by output: functions of the CIRCUIT --
// and a per-session part (work buffers, the integer copies of the tensors, the range words). The static part belongs to the resident circuit's
// registry entry: the first context that uploads the program builds it there (under the entry's lock), every other context of the circuit -- one
// that uploads the same program later, or a clone (zk_ctx_clone) that never sees the host data -- adopts it and allocates only its own buffers.
static int32_t wp_build_static(zk_ctx *ctx, shared_circuit *e, const zk_witness_op *ops, uint64_t n_ops, const uint32_t *windows, uint64_t n_windows,
                               const zk_witness_step *steps, uint32_t n_steps, const zk_layer_desc *layers, int32_t n_layers);

extern "C" int32_t zk_witness_program_upload(zk_ctx *ctx, const zk_witness_op *ops, uint64_t n_ops, const uint32_t *windows, uint64_t n_windows,
                                             const zk_witness_step *steps, uint32_t n_steps, const zk_layer_desc *layers, int32_t n_layers) {
    ZK_CHECK_READY();
    static_assert(sizeof(zk_witness_op) == sizeof(wit_op) && sizeof(zk_witness_op) == 12 && sizeof(zk_witness_step) == 48, "program records");
    if (ctx->wp_ready) { ctx->err = "witness program already uploaded"; return ZK_ERR_STATE; }
    if (n_layers != (int32_t) ctx->L.size() || !steps || !n_steps || (n_ops && !ops) || (n_windows && !windows)) return ZK_ERR_ARG;
    shared_circuit *e = (shared_circuit *) ctx->circuit;
    if (!e) return ZK_ERR_STATE;
    {
        // (the attach below reads e->L[i].ev_* and e->wp_*, which another session's first upload writes under this lock: round-4 advisor finding)
        std::lock_guard<std::mutex> g(e->mtx);
        if (!e->wp_ready) {
            const std::vector<dev_layer> keep = ctx->L;          // (a failed build leaves the context as it was)
            ctx->alloc_sink = &e->owned;
            ctx->sink_bytes = 0;
            int32_t rc = wp_build_static(ctx, e, ops, n_ops, windows, n_windows, steps, n_steps, layers, n_layers);
            ctx->alloc_sink = nullptr;
            if (rc == ZK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "witness program upload failed"; rc = ZK_ERR_HIP; }
            if (rc) { ctx->L = keep; return rc; }
            e->bytes += ctx->sink_bytes;
            e->wp_ready = true;
        }
        return zk_witness_program_adopt(ctx);
    }
}

static int32_t wp_build_static(zk_ctx *ctx, shared_circuit *e, const zk_witness_op *ops, uint64_t n_ops, const uint32_t *windows, uint64_t n_windows,
                               const zk_witness_step *steps, uint32_t n_steps, const zk_layer_desc *layers, int32_t n_layers) {
    const uint64_t n0 = ctx->L[0].d.size;
    int32_t rc;
    // ---- the steps: every index an operation touches must exist ----
    uint32_t n_ranges = 0;
    std::vector<uint8_t> evaluated(n_layers, 0);
    evaluated[0] = 1;
    for (uint32_t k = 0; k < n_steps; ++k) {
        const zk_witness_step &st = steps[k];
        if (st.layer < 0 || st.layer >= n_layers) { ctx->err = "witness step: layer out of range"; return ZK_ERR_ARG; }
        if (st.what == 1) {
            if (st.layer < 1 || !evaluated[st.layer - 1]) { ctx->err = "witness step: layer evaluated before its inputs"; return ZK_ERR_ARG; }
            evaluated[st.layer] = 1;
        } else if (st.what == 2) {
            if (!evaluated[st.layer]) { ctx->err = "witness step: range of a layer not yet evaluated"; return ZK_ERR_ARG; }
            ++n_ranges;
        } else if (st.what == 0) {
            if (st.op_begin > st.op_end || st.op_end > n_ops || !evaluated[st.layer]) { ctx->err = "witness step: bad operation span"; return ZK_ERR_ARG; }
            const uint64_t src_n = ctx->L[st.layer].d.size;
            const bool sum = st.op_begin < st.op_end && ops[st.op_begin].op == 3;
            uint64_t n_win = 0;
            if (sum) {
                if (st.win < 1 || st.win_begin > n_windows) { ctx->err = "witness step: bad window table"; return ZK_ERR_ARG; }
                n_win = (n_windows - st.win_begin) / (uint64_t) st.win;
            }
            uint64_t used_win = 0;
            for (uint64_t j = st.op_begin; j < st.op_end; ++j) {
                const zk_witness_op &op = ops[j];
                if (op.op > 3 || op.dst >= n0 || op.shift > 63 || (op.op == 3) != sum || ((op.op == 2) != (ops[st.op_begin].op == 2)) ||
                    (sum ? op.src >= n_win : op.src >= src_n)) { ctx->err = "witness operation out of range"; return ZK_ERR_ARG; }
                if (sum) used_win = std::max<uint64_t>(used_win, (uint64_t) op.src + 1);
            }
            for (uint64_t w = st.win_begin; w < st.win_begin + used_win * (uint64_t) (sum ? st.win : 0); ++w)
                if (windows[w] >= src_n) { ctx->err = "witness window entry out of range"; return ZK_ERR_ARG; }
        } else {
            ctx->err = "witness step: unknown kind";
            return ZK_ERR_ARG;
        }
    }
    for (int i = 1; i < n_layers; ++i)
        if (!evaluated[i]) { ctx->err = "witness program does not evaluate every layer"; return ZK_ERR_ARG; }
    // ---- gate lists grouped by output, operands in layer 0 as raw layer-0 indices ----
    uint64_t max_out = 1, max_blocks = 1, max_conv_part = 0, max_conv_in = 0, conv_w_total = 0;
    std::vector<uint64_t> conv_w_off(n_layers, 0);
    for (int i = 1; i < n_layers; ++i) {
        const zk_layer_desc &S = layers[i];
        dev_layer &D = ctx->L[i];
        if (S.size != D.d.size || S.ty != D.d.ty) { ctx->err = "witness program: layer descriptors differ from the uploaded circuit"; return ZK_ERR_ARG; }
        const uint64_t n_out = S.size, n_prev = ctx->L[i - 1].d.size;
        if (S.ty == ZK_FFT || S.ty == ZK_IFFT) {
            if (S.fft_bit_length < 1 || S.fft_bit_length > 12) { ctx->err = "witness program: transform longer than 2^12"; return ZK_ERR_ARG; }
            continue;
        }
        if (S.ty == ZK_DOT_PROD) {
            const int fb = S.fft_bit_length;
            if (fb < 1 || fb > 20) return ZK_ERR_ARG;
            const uint64_t vec_out = n_out >> fb, vec_in = n_prev >> fb;
            std::vector<uint32_t> ptr(vec_out + 1, 0);
            for (uint64_t k = 0; k < S.n_bin; ++k) {
                const zk_bin_gate &gt = S.bin_gates[k];
                if (gt.g >= vec_out || gt.u >= vec_in || gt.v >= vec_in) { ctx->err = "dot-prod gate out of range"; return ZK_ERR_ARG; }
                ++ptr[gt.g + 1];
            }
            for (uint64_t g = 0; g < vec_out; ++g) ptr[g + 1] += ptr[g];
            std::vector<gate_rec> sorted(S.n_bin);
            std::vector<uint32_t> pos(ptr.begin(), ptr.end() - 1);
            for (uint64_t k = 0; k < S.n_bin; ++k) {
                const zk_bin_gate &gt = S.bin_gates[k];
                gate_rec r = {gt.g, gt.u, gt.v, 0};
                sorted[pos[gt.g]++] = r;
            }
            if ((rc = zk_upload(ctx, &D.ev_dot, sorted)) || (rc = zk_upload(ctx, &D.ev_dot_ptr, ptr))) return rc;
            continue;
        }
        std::vector<zk_uni_gate> uni(S.uni_gates, S.uni_gates + S.n_uni);
        // a convolution whose pattern was checked at upload is evaluated from its two tensors (k_conv_eval): its bin gates are not kept
        const bool conv_eval = D.conv_ok;
        std::vector<zk_bin_gate> bin;
        if (!conv_eval) bin.assign(S.bin_gates, S.bin_gates + S.n_bin);
        else {
            const conv_desc &c = D.conv;
            const uint64_t n_in = (uint64_t) c.pp * c.CI * c.nxi * c.nyi, n_w = (uint64_t) c.CO * c.CI * c.m * c.m;
            if (n_in > ctx->L[i - 1].val_len || c.wstart + n_w > ctx->L[0].val_len) { ctx->err = "witness program: convolution tensors outside their layers"; return ZK_ERR_ARG; }
            max_conv_part = std::max<uint64_t>(max_conv_part, (uint64_t) conv_eval_chunks(c, n_out) * n_out);
            max_conv_in = std::max(max_conv_in, n_in);
            conv_w_off[i] = conv_w_total;
            conv_w_total += n_w;
        }
        bool ok = true;
        for (zk_uni_gate &gt : uni) {
            if (gt.lu == 0) { if (gt.u >= S.size_u[0]) { ok = false; break; } gt.u = S.ori_id_u[gt.u]; }
        }
        for (zk_bin_gate &gt : bin) {
            if (!ok || gt.l > 2) { ok = false; break; }
            if (gt.l == 0) { if (gt.u >= S.size_u[0]) { ok = false; break; } gt.u = S.ori_id_u[gt.u]; }
            if (!(gt.l & 1)) { if (gt.v >= S.size_v[0]) { ok = false; break; } gt.v = S.ori_id_v[gt.v]; }
        }
        if (!ok) { ctx->err = "witness program: gate operand outside its subset"; return ZK_ERR_ARG; }
        std::vector<uint8_t> seen(n_out);
        int g1 = grouped_by_output(uni.data(), uni.size(), n_out, seen, [&](const zk_uni_gate &gt) {
            return (int) gt.sc < ctx->n_two_mul && gt.u < (gt.lu ? n_prev : n0); });
        int g2 = g1 < 0 ? -1 : grouped_by_output(bin.data(), bin.size(), n_out, seen, [&](const zk_bin_gate &gt) {
            return (int) gt.sc < ctx->n_two_mul && gt.u < (gt.l == 0 ? n0 : n_prev) && gt.v < ((gt.l & 1) ? n_prev : n0); });
        if (g1 < 0 || g2 < 0) { ctx->err = "witness gate operand out of range"; return ZK_ERR_ARG; }
        if (!g1) { std::vector<zk_uni_gate> t; regroup(t, uni.data(), uni.size(), n_out); uni.swap(t); }
        if (!g2) { std::vector<zk_bin_gate> t; regroup(t, bin.data(), bin.size(), n_out); bin.swap(t); }
        zk_uni_gate *du = nullptr;
        zk_bin_gate *db = nullptr;
        if ((rc = zk_upload(ctx, &du, uni)) || (rc = zk_upload(ctx, &db, bin))) return rc;
        D.ev_uni = du; D.n_ev_uni = uni.size();
        D.ev_bin = db; D.n_ev_bin = bin.size();
        D.ev_conv = conv_eval;
        max_out = std::max(max_out, n_out);
        max_blocks = std::max<uint64_t>(max_blocks, (std::max(uni.size(), bin.size()) + ZK_BLOCK - 1) / ZK_BLOCK);
    }
    std::vector<zk_witness_op> vops(ops, ops + n_ops);
    std::vector<uint32_t> vwin(windows, windows + n_windows);
    zk_witness_op *d_ops = nullptr;
    if ((rc = zk_upload(ctx, &d_ops, vops)) || (rc = zk_upload(ctx, &e->wp_windows, vwin))) return rc;
    e->wp_ops = d_ops;
    e->wp_n_ops = n_ops;
    e->wp_n_windows = n_windows;
    e->wp_steps.assign(steps, steps + n_steps);
    e->wp_n_ranges = n_ranges;
    e->wp_max_out = max_out; e->wp_max_blocks = max_blocks; e->wp_max_conv_part = max_conv_part; e->wp_max_conv_in = max_conv_in;
    e->wp_conv_w_total = conv_w_total; e->wp_conv_w_off = conv_w_off;
    for (int i = 1; i < n_layers; ++i) {                 // the lists grouped by output belong to the circuit: whoever attaches finds them in the entry
        dev_layer &S = e->L[i];
        const dev_layer &D = ctx->L[i];
        S.ev_uni = D.ev_uni; S.ev_bin = D.ev_bin; S.n_ev_uni = D.n_ev_uni; S.n_ev_bin = D.n_ev_bin;
        S.ev_dot = D.ev_dot; S.ev_dot_ptr = D.ev_dot_ptr; S.ev_conv = D.ev_conv;
    }
    return ZK_OK;
}

// the per-session side: the circuit's program (built above, by this context or by another) becomes this context's
int32_t zk_witness_program_adopt(zk_ctx *ctx) {
    shared_circuit *e = (shared_circuit *) ctx->circuit;
    if (!e || !e->wp_ready) { ctx->err = "the resident circuit has no witness program"; return ZK_ERR_STATE; }
    if (ctx->wp_ready) return ZK_OK;
    int32_t rc;
    const int32_t n_layers = (int32_t) ctx->L.size();
    for (int i = 1; i < n_layers; ++i) {
        const dev_layer &S = e->L[i];
        dev_layer &D = ctx->L[i];
        D.ev_uni = S.ev_uni; D.ev_bin = S.ev_bin; D.n_ev_uni = S.n_ev_uni; D.n_ev_bin = S.n_ev_bin;
        D.ev_dot = S.ev_dot; D.ev_dot_ptr = S.ev_dot_ptr; D.ev_conv = S.ev_conv;
    }
    ctx->wp_ops = e->wp_ops;
    ctx->wp_windows = e->wp_windows;
    ctx->wp_n_ops = e->wp_n_ops;
    ctx->wp_n_windows = e->wp_n_windows;
    const uint64_t max_out = e->wp_max_out, max_blocks = e->wp_max_blocks, max_conv_part = e->wp_max_conv_part, max_conv_in = e->wp_max_conv_in,
                   conv_w_total = e->wp_conv_w_total;
    const std::vector<uint64_t> &conv_w_off = e->wp_conv_w_off;
    const uint32_t n_ranges = e->wp_n_ranges;
    if (max_conv_part && (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_conv_part, max_conv_part * 32))) return rc;
    if (conv_w_total) {
        if ((rc = zk_dev_alloc(ctx, (void **) &ctx->wp_in64, max_conv_in * 8)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_w64, conv_w_total * 8)) ||
            (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_cb_in, 2 * (size_t) n_layers * 8)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_cb_w, 2 * (size_t) n_layers * 8)))
            return rc;
        for (int i = 1; i < n_layers; ++i)
            if (ctx->L[i].ev_conv) ctx->L[i].ev_w64 = ctx->wp_w64 + conv_w_off[i];
        ctx->wp_w64_valid = false;
    }
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->wp_tmp, 2 * max_out * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_carry_val, 2 * max_blocks * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_carry_key, 2 * max_blocks * 4)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_ranges, (2 * (size_t) n_ranges + 1 + n_layers) * 8)))
        return rc;
    ZK_HIP(hipHostMalloc((void **) &ctx->h_wp_ranges, (2 * (size_t) n_ranges + 1 + n_layers) * 8));
    {
        std::vector<wit_segment> seg(n_layers);
        for (int i = 0; i < n_layers; ++i) { seg[i].p = ctx->L[i].val; seg[i].n = ctx->L[i].d.size; }
        wit_segment *d_seg = nullptr;
        if ((rc = zk_upload(ctx, &d_seg, seg))) return rc;
        ctx->wp_segments = d_seg;
    }
    ctx->wp_n_ranges = n_ranges;
    ctx->wp_steps = e->wp_steps;
    ctx->wp_ready = true;
    return ZK_OK;
}

// FFT / IFFT layer between two resident value tables (the host-array variant is zk_witness_ntt)
static int32_t resident_ntt(zk_ctx *ctx, const dev_layer &D, const dev_layer &P) {
    const int logn = D.d.fft_bit_length;
    const bool inverse = D.d.ty == ZK_IFFT;
    const uint32_t len = 1u << logn, in_len = inverse ? len : len / 2, out_len = inverse ? len / 2 : len;
    const uint64_t count = D.d.size / out_len;
    if (!count || (uint64_t) count * in_len > P.val_len) { ctx->err = "transform layer larger than its input"; return ZK_ERR_ARG; }
    fr_t *pw = zk_powers_of_root(ctx, logn, inverse);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    HFr ilen;
    HFr::inv(ilen, HFr((unsigned long long) len));
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_batch), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const uint32_t threads = std::min<uint32_t>(1024, std::max<uint32_t>(64, len / 2));
    prof_begin(ctx, PC_MISC, 64.0 * len * count);
    hipLaunchKernelGGL(k_ntt_batch, dim3((uint32_t) count), dim3(threads), (size_t) len * 32, ctx->stream, D.val, P.val, pw, logn, in_len, out_len,
                       to_dev(ilen), inverse ? 1 : 0);
    prof_end(ctx, PC_MISC);
    return ZK_OK;
}

extern "C" int32_t zk_witness_rerun(zk_ctx *ctx, const uint64_t *picture, uint64_t n_picture, uint64_t *ranges, uint32_t n_ranges,
                                    uint64_t *last_layer, uint64_t n_last) {
    ZK_CHECK_READY();
    if (!ctx->wp_ready) { ctx->err = "no witness program on this context"; return ZK_ERR_STATE; }
    dev_layer &L0 = ctx->L[0];
    if (!picture || !n_picture || n_picture > L0.d.size || n_ranges != ctx->wp_n_ranges || (n_ranges && !ranges) ||
        (last_layer && n_last > ctx->L.back().d.size))
        return ZK_ERR_ARG;
    int32_t rc;
    ZK_HIP(hipMemcpyAsync(L0.val, picture, n_picture * 32, hipMemcpyHostToDevice, ctx->stream));
    const size_t n_lay = ctx->L.size(), wp_words = 2 * (size_t) n_ranges + 1 + n_lay;
    ZK_HIP(hipMemsetAsync(ctx->wp_ranges, 0, wp_words * 8, ctx->stream));
    uint32_t *flags = (uint32_t *) (ctx->wp_ranges + 2 * (size_t) n_ranges);
    uint32_t range_k = 0;
    if (ctx->wp_w64) {
        ZK_HIP(hipMemsetAsync(ctx->wp_cb_in, 0, 2 * n_lay * 8, ctx->stream));
        if (!ctx->wp_w64_valid) {           // first picture, or layer 0 was written from outside since: the weights as integers
            ZK_HIP(hipMemsetAsync(ctx->wp_cb_w, 0, 2 * n_lay * 8, ctx->stream));
            for (size_t i = 1; i < n_lay; ++i) {
                const dev_layer &C = ctx->L[i];
                if (!C.ev_conv) continue;
                const uint64_t n_w = (uint64_t) C.conv.CO * C.conv.CI * C.conv.m * C.conv.m;
                ZK_LAUNCH(PC_MISC, 40.0 * (double) n_w, k_fr_to_i64, dim3(grid_for(n_w)), dim3(ZK_BLOCK), C.ev_w64, (const fr_t *) L0.val + C.conv.wstart, n_w, ctx->wp_cb_w + 2 * i,
                          C.conv.CO, C.conv.CI * C.conv.m * C.conv.m);
            }
            ctx->wp_w64_valid = true;
        }
    }
    for (const zk_witness_step &st : ctx->wp_steps) {
        dev_layer &D = ctx->L[st.layer];
        if (st.what == 0) {
            const uint64_t n = st.op_end - st.op_begin;
            if (!n) continue;
            ZK_LAUNCH(PC_MISC, 0.0, k_witness_aux, dim3(grid_for(n)), dim3(ZK_BLOCK), L0.val, (const fr_t *) D.val, (const wit_op *) ctx->wp_ops + st.op_begin, n,
                      ctx->wp_windows ? ctx->wp_windows + st.win_begin : nullptr, (uint32_t) st.win, flags);
        } else if (st.what == 2) {
            ZK_LAUNCH(PC_MISC, 0.0, k_witness_range, dim3(grid_for(D.d.size, 512)), dim3(ZK_BLOCK), ctx->wp_ranges + 2 * (size_t) range_k, (const fr_t *) D.val,
                      (uint64_t) D.d.size, flags);
            ++range_k;
        } else {
            const dev_layer &P = ctx->L[st.layer - 1];
            if (D.d.ty == ZK_FFT || D.d.ty == ZK_IFFT) {
                if ((rc = resident_ntt(ctx, D, P))) return rc;
            } else if (D.d.ty == ZK_DOT_PROD) {
                const int fb = D.d.fft_bit_length;
                const uint64_t len = 1ull << fb, n_out = D.d.size >> fb;
                for (uint64_t g0 = 0; g0 < n_out; g0 += 32768) {
                    const uint32_t ng = (uint32_t) std::min<uint64_t>(32768, n_out - g0);
                    dim3 grid((uint32_t) ((len + ZK_BLOCK - 1) / ZK_BLOCK), ng);
                    ZK_LAUNCH(PC_DOT, 0.0, k_dot_witness, grid, dim3(ZK_BLOCK), D.val + g0 * len, (const fr_t *) P.val, (const gate_rec *) D.ev_dot, D.ev_dot_ptr + g0, fb);
                }
            } else {
                const uint64_t n_out = D.d.size;
                fr_t *dA = ctx->wp_tmp, *dB = dA + n_out;
                ZK_HIP(hipMemsetAsync(dA, 0, 2 * n_out * 32, ctx->stream));
                if (D.n_ev_uni) {
                    const uint64_t blocks = (D.n_ev_uni + ZK_BLOCK - 1) / ZK_BLOCK;
                    ZK_LAUNCH(PC_GATE, 44.0 * (double) D.n_ev_uni, k_eval_uni, dim3((uint32_t) blocks), dim3(ZK_BLOCK), dA, ctx->wp_carry_key, ctx->wp_carry_val,
                              (const uni_gate_dev *) D.ev_uni, D.n_ev_uni, (const fr_t *) L0.val, (const fr_t *) P.val, (const fr_t *) ctx->two_mul);
                    ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks)), dim3(ZK_BLOCK), dA, ctx->wp_carry_key, ctx->wp_carry_val, 2 * blocks);
                }
                if (D.ev_conv) {
                    const conv_desc &c = D.conv;
                    const uint32_t chunks = conv_eval_chunks(c, n_out), per = (c.CI + chunks - 1) / chunks;
                    fr_t *part = chunks == 1 ? dB : ctx->wp_conv_part;
                    // in 64-bit integers when the tensors allow it (decided on the device: witness_kernels.cuh), in the field otherwise
                    const uint64_t n_in = (uint64_t) c.pp * c.CI * c.nxi * c.nyi;
                    unsigned long long *cb_in = ctx->wp_cb_in + 2 * (size_t) st.layer, *cb_w = ctx->wp_cb_w + 2 * (size_t) st.layer;
                    const dim3 cgrid((uint32_t) ((n_out + ZK_BLOCK - 1) / ZK_BLOCK), chunks);
                    ZK_LAUNCH(PC_MISC, 40.0 * (double) n_in, k_fr_to_i64, dim3(grid_for(n_in)), dim3(ZK_BLOCK), ctx->wp_in64, (const fr_t *) P.val, n_in, cb_in, 0u, 0u);
                    ZK_LAUNCH(PC_GATE, 16.0 * (double) n_out * c.CI * c.m * c.m, k_conv_eval_i64, cgrid, dim3(ZK_BLOCK), part, (const long long *) ctx->wp_in64,
                              (const long long *) D.ev_w64, c, per, (const unsigned long long *) cb_in, (const unsigned long long *) cb_w, ctx->wp_force_field);
                    ZK_LAUNCH(PC_GATE, 0.0, k_conv_eval, cgrid, dim3(ZK_BLOCK), part, (const fr_t *) P.val, (const fr_t *) L0.val + c.wstart, c, per,
                              (const unsigned long long *) cb_in, (const unsigned long long *) cb_w, ctx->wp_force_field);
                    if (chunks > 1)
                        ZK_LAUNCH(PC_GATE, 0.0, k_sum_rows, dim3((uint32_t) ((n_out + 63) / 64)), dim3(1024), dB, (const fr_t *) part, (uint32_t) n_out, chunks);
                }
                if (D.n_ev_bin) {
                    const uint64_t blocks = (D.n_ev_bin + ZK_BLOCK - 1) / ZK_BLOCK;
                    ZK_LAUNCH(PC_GATE, 80.0 * (double) D.n_ev_bin, k_eval_bin, dim3((uint32_t) blocks), dim3(ZK_BLOCK), dB, ctx->wp_carry_key, ctx->wp_carry_val,
                              (const bin_gate_dev *) D.ev_bin, D.n_ev_bin, (const fr_t *) L0.val, (const fr_t *) P.val, (const fr_t *) ctx->two_mul);
                    ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks)), dim3(ZK_BLOCK), dB, ctx->wp_carry_key, ctx->wp_carry_val, 2 * blocks);
                }
                const HFr sc = H(D.d.scale);
                ZK_LAUNCH(PC_MISC, 0.0, k_eval_combine, dim3(grid_for(n_out)), dim3(ZK_BLOCK), D.val, (const fr_t *) dA, (const fr_t *) dB, to_dev(sc),
                          sc == HFr::one() ? 0 : 1, n_out);
            }
        }
    }
    // where every layer's values end (the round kernels skip the zero tails of the tables they fold)
    ZK_LAUNCH(PC_MISC, 0.0, k_last_nonzero, dim3(64, (uint32_t) n_lay), dim3(ZK_BLOCK), ctx->wp_ranges + 2 * (size_t) n_ranges + 1, (const wit_segment *) ctx->wp_segments);
    ZK_HIP(hipGetLastError());
    for (dev_layer &D : ctx->L) D.val_live = D.val_len;          // until the copy below is back
    ZK_HIP(hipMemcpyAsync(ctx->h_wp_ranges, ctx->wp_ranges, wp_words * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (last_layer && n_last) ZK_HIP(hipMemcpyAsync(last_layer, ctx->L.back().val, n_last * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t k = 0; k < 2 * n_ranges; ++k) ranges[k] = ctx->h_wp_ranges[k];
    for (size_t i = 0; i < n_lay; ++i) ctx->L[i].val_live = ctx->h_wp_ranges[2 * (size_t) n_ranges + 1 + i];
    if (ctx->h_wp_ranges[2 * (size_t) n_ranges] & WIT_FLAG_WIDE) { ctx->err = "a layer value does not fit 63 bits"; return ZK_ERR_STATE; }
    return ZK_OK;
}

// Test hook: which arithmetic the structured convolutions of the LAST zk_witness_rerun were evaluated in. out[0] = convolutions evaluated from
// their tensors, out[1] = of those, the ones that went through k_conv_eval_i64, out[2] = where the first one's weights start in layer 0. force_field >= 0 sets the switch that
// keeps every convolution in field arithmetic (1) or lets the bounds decide (0) for the following calls; < 0 leaves it alone.
extern "C" int32_t zk_witness_conv_paths(zk_ctx *ctx, int32_t force_field, uint64_t out[3]) {
    ZK_CHECK_READY();
    if (!out) return ZK_ERR_ARG;
    out[0] = out[1] = out[2] = 0;
    if (force_field >= 0) ctx->wp_force_field = force_field ? 1 : 0;
    if (!ctx->wp_ready || !ctx->wp_w64) return ZK_OK;
    const size_t n_lay = ctx->L.size();
    std::vector<unsigned long long> cin(2 * n_lay), cw(2 * n_lay);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_HIP(hipMemcpy(cin.data(), ctx->wp_cb_in, 2 * n_lay * 8, hipMemcpyDeviceToHost));
    ZK_HIP(hipMemcpy(cw.data(), ctx->wp_cb_w, 2 * n_lay * 8, hipMemcpyDeviceToHost));
    auto bits = [](unsigned long long x) { int b = 0; while (x) { ++b; x >>= 1; } return b; };
    for (size_t i = 1; i < n_lay; ++i) {
        const dev_layer &D = ctx->L[i];
        if (!D.ev_conv) continue;
        if (!out[0]++) out[2] = D.conv.wstart;
        const bool fits = !(cin[2 * i + 1] | cw[2 * i + 1]) && bits(cin[2 * i]) + bits(cw[2 * i]) + bits((unsigned long long) D.conv.CI * D.conv.m * D.conv.m) <= 62;
        if (fits && ctx->wp_w64_valid && !ctx->wp_force_field) ++out[1];
    }
    return ZK_OK;
}
