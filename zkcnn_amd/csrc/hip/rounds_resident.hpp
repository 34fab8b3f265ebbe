// Host side of the round drivers that are NOT "one launch per round" -- included by sumcheck.hip (one translation unit: the kernels of kernels.cuh /
// fs_tail.cuh have external linkage). The resident kernels of the interactive protocol (k_tail<true>, k_mid<true>: a round is "challenge into the
// mailbox, polynomial out of the mailbox"), the device-side Fiat-Shamir rounds (k_tail<false>, k_mid<false>: the kernel derives the challenges itself),
// and the opt-in hybrid tail (the last rounds of a phase on the host). DESIGN.md 4a, 4h. reference src/prover.cpp:360-426 is what every one of them computes.
#pragma once

// One quadratic round over the live table pairs. reference src/prover.cpp:368-426.
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// Non-interactive mode: all remaining rounds of the phase in one single-workgroup kernel (fs_tail.cuh). Called at the start of a round
// whose live tables are small; fills the context's tail record and leaves every table pair either collapsed or down to its last pair.
// (Round 2 also had CHAINED launches for tables of up to 2^16 entries -- one launch per round, the next challenge derived in its last block;
// transcripts identical, 0.5 ms per vgg11 proof slower than host-driven rounds: removed in round 3.)
static int32_t run_device_rounds(zk_ctx *ctx, const HFr &r, bool with_add_term) {
    const bool first = ctx->round == 0;
    const unsigned long long seq = ++ctx->tail_seq;
    tail_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.first = first ? 1 : 0;
    A.rounds = ctx->phase_rounds - ctx->round;
    A.with_add_term = with_add_term ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    std::memcpy(A.fs_state, ctx->fs_state, 32);
    A.out = (tail_out *) ctx->d_tail;
    A.seq = seq;
    ZK_LAUNCH(PC_TAIL, 0.0, k_tail<false>, dim3(1), dim3(TAIL_THREADS), A);
    ZK_HIP(hipGetLastError());
    volatile unsigned long long *p = &((tail_out *) ctx->h_tail)->seq;
    for (uint64_t spins = 0; *p != seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != seq) { ctx->err = "device rounds were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const tail_out *o = (const tail_out *) ctx->h_tail;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        t.Vsrc = nullptr;
        if (o->pair_state[b] == 1) {
            t.len = 2;
            std::memcpy(&t.tail_v[0], &o->tail_v[b][0], 32);
            std::memcpy(&t.tail_v[1], &o->tail_v[b][1], 32);
            t.tail_valid = true;
        } else {
            t.len = 0;
            t.absorbed = true;
            std::memcpy(&t.final_v, &o->final_v[b], 32);
        }
    }
    std::memcpy(&ctx->add_term, &o->add_term, 32);
    ctx->tail_active = true;
    ctx->tail_count = A.rounds;
    ctx->tail_cursor = 0;
    ctx->last_poly_valid = false;          // (the rounds answered from the record do not maintain the running claim)
    ctx->tail_rounds_total += (uint64_t) ctx->tail_count;
    ++ctx->tail_phases_total;
    return ZK_OK;
}

// ---- persistent rounds of the INTERACTIVE protocol (k_tail<true>, fs_tail.cuh): the kernel is resident for the rest of the phase, a round is
// "challenge into the mailbox, polynomial out of the mailbox" ----
static inline void live_post(zk_ctx *ctx, const HFr &r, uint32_t seq) {
    uint32_t w[8];
    std::memcpy(w, &r, 32);
    live_in *m = (live_in *) ctx->h_live_in;
    // one 16-byte store per chunk (the kernel reads a chunk with one 16-byte load and checks the number in each)
    _mm_store_si128((__m128i *) m->c[0], _mm_set_epi32((int) seq, (int) w[2], (int) w[1], (int) w[0]));
    _mm_store_si128((__m128i *) m->c[1], _mm_set_epi32((int) seq, (int) w[5], (int) w[4], (int) w[3]));
    _mm_store_si128((__m128i *) m->c[2], _mm_set_epi32((int) seq, 0, (int) w[7], (int) w[6]));
    _mm_sfence();
}
// a lane leaves the set of lanes whose tail is running (its last round, an abort, a lost kernel)
static inline void lane_tail_done(zk_ctx *ctx) {
    if (!ctx->lane_tail) return;
    ctx->lane_tail = false;
    zk_batch *b = ctx->batch;
    if (b && b->tails_running > 0 && --b->tails_running == 0) b->tails_open_at = b->n_flushes + 1;     // (the last one out: the others may go on after the NEXT flush)
}
int32_t zk_live_abort(zk_ctx *ctx) {
    if (!ctx->live_active) return ZK_OK;
    live_post(ctx, HFr(0LL), TAIL_ABORT);
    ctx->live_active = false;
    ctx->live_mid = false;
    if (ctx->lane_tail) {
        // the lane's workgroup of a fused tail: it sees the abort word and leaves on its own. No wait here -- the launch ends when the OTHER lanes'
        // workgroups end, and those need this thread (the lanes share it) for their challenges
        lane_tail_done(ctx);
        for (int b = 0; b < 2; ++b) ctx->tp[b].len = 0;
        return ZK_OK;
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_HIP(hipMemsetAsync(ctx->d_counter, 0, 64, ctx->stream));          // (a segment kernel that was sent home may have left arrivals behind ...
    ZK_HIP(hipMemsetAsync(ctx->d_bcast, 0, sizeof(mid_bcast), ctx->stream));   //  ... and the abort word in its broadcast line)
    std::memset(ctx->h_live_in, 0, sizeof(live_in));
    for (int b = 0; b < 2; ++b) ctx->tp[b].len = 0;
    return ZK_OK;
}
// waits for the polynomial of round k of the running kernel; false if the kernel has left (status) or nothing arrives
static int32_t live_wait(zk_ctx *ctx, int k, uint64_t out_abc[12]) {
    // test hook (ZKCNN_TEST_HOOKS=1, ZKCNN_TEST_LIVE_FAIL=n): the n-th resident round of the process loses its kernel (as a time-out would)
    static const long fail_at = (getenv("ZKCNN_TEST_HOOKS") && atoi(getenv("ZKCNN_TEST_HOOKS")) && getenv("ZKCNN_TEST_LIVE_FAIL")) ? atol(getenv("ZKCNN_TEST_LIVE_FAIL")) : -1;
    static std::atomic<long> resident_rounds{0};
    if (fail_at >= 0 && resident_rounds.fetch_add(1) == fail_at) {
        (void) zk_live_abort(ctx);
        for (int b = 0; b < 2; ++b) ctx->tp[b].len = 1;      // (zk_live_abort cleared them; the replay rebuilds every table)
        ctx->err = "test hook: resident round kernel sent home";
        ctx->live_lost = true;
        return ZK_ERR_STATE;
    }
    const tail_out *o = (const tail_out *) ctx->h_tail;
    const uint32_t want = ctx->live_seq32 + (uint32_t) k;
    uint32_t w[24];
    for (uint64_t spins = 0;; ++spins) {
        bool ok = true;
        for (int j = 0; j < 8; ++j) {
            const __m128i c = _mm_load_si128((const __m128i *) o->live.c[j]);
            alignas(16) uint32_t t[4];
            _mm_store_si128((__m128i *) t, c);
            if (t[3] != want) { ok = false; break; }
            w[3 * j] = t[0]; w[3 * j + 1] = t[1]; w[3 * j + 2] = t[2];
        }
        if (ok) break;
        if (ctx->lane_tail) {
            // a lane: the thread goes to the other lanes (the first time round this is what gets the fused launch issued: zk_batch_sync_point)
            int32_t rcy = zk_batch_sync_point(ctx);
            if (rcy) { lane_tail_done(ctx); ctx->live_active = false; return rcy; }
            if (spins < (1ull << 12) || (spins & 255) != 0 || ctx->n_pending) continue;
        }
        if ((ctx->lane_tail || spins > (1ull << 16)) && (spins & 1023) == 0) {
            if (*(volatile const uint32_t *) &o->status != 0 || hipStreamQuery(ctx->stream) == hipSuccess) {
                // (a kernel that has left: one last look, its final message may have landed after the check above)
                bool late = true;
                for (int j = 0; j < 8; ++j) if (((volatile const uint32_t *) o->live.c[j])[3] != want) late = false;
                if (late) continue;
                lane_tail_done(ctx);
                ctx->live_active = false;
                ctx->live_lost = true;                 // (quad_round runs the phase again with a launch per round)
                char msg[256];
                const live_in *mi = (const live_in *) ctx->h_live_in;
                std::snprintf(msg, sizeof(msg), "the resident round kernel left before the phase was over (%s, round %d of %d, status %#x, waiting for %#x, mailbox out %#x in %#x, stream %s)",
                              ctx->live_mid ? "k_mid" : "k_tail", k, ctx->live_count, (unsigned) o->status, (unsigned) want, (unsigned) o->live.c[0][3], (unsigned) mi->c[0][3],
                              hipStreamQuery(ctx->stream) == hipSuccess ? "idle" : "busy");
                ctx->err = msg;
                ctx->live_mid = false;
                return ZK_ERR_STATE;
            }
            if (spins > (1ull << 22)) sched_yield();
        } else __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    std::memcpy(out_abc, w, 96);
    return ZK_OK;
}
// export_log >= 0: the kernel ends when the tables are down to 2^export_log entries and hands them to the host (the hybrid tail goes on from there) --
// if the phase has that many rounds and the hand-over is behind a fold; otherwise the kernel runs the phase to its end as before
static int32_t live_start(zk_ctx *ctx, const HFr &r, bool with_add_term, int export_log = -1) {
    tail_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.first = ctx->round == 0 ? 1 : 0;
    A.rounds = ctx->phase_rounds - ctx->round;
    if (export_log >= 0 && (1u << export_log) <= TAIL_EXPORT_MAX) {
        // rounds until the longer table is down to 2^export_log entries: a first round folds nothing, every other one halves
        uint64_t longest = std::max(ctx->tp[0].len, ctx->tp[1].len);
        int m = A.first ? 1 : 0;
        while (longest > (1ull << export_log)) { longest >>= 1; ++m; }
        if (m >= 2 && m < A.rounds) { A.rounds = m; A.export_tables = 1; }
    }
    A.with_add_term = with_add_term ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.out = (tail_out *) ctx->d_tail;
    A.in = (const live_in *) ctx->d_live_in;
    // sequence numbers of this kernel's rounds: never 0 (a cleared mailbox), never the abort word, never one of the previous kernel's
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    ((tail_out *) ctx->h_tail)->status = 0;
    if (ctx->batch) {
        // a lane: its tail is one workgroup of the batch's fused launch (issued when every lane has parked: the first live_wait)
        live_in *mb = (live_in *) ctx->h_live_in;          // (an abort word left by zk_live_abort: that workgroup is long gone, this one must not read it)
        if (mb->c[0][3] == TAIL_ABORT) std::memset(mb, 0, sizeof(live_in));
        zk_launch_f<k_tail_live_f, TAIL_THREADS>(ctx, PC_TAIL, 0.0, dim3(1), k_tail_live_f{A});
        ctx->lane_tail = true;
        ++ctx->batch->tails_running;
        ++ctx->batch->n_lane_tails;
        ctx->batch->n_lane_tail_rounds += (uint64_t) A.rounds;
    } else {
        ZK_LAUNCH(PC_TAIL, 0.0, k_tail<true>, dim3(1), dim3(TAIL_THREADS), A);
        ZK_HIP(hipGetLastError());
    }
    ctx->live_active = true;
    ctx->live_count = A.rounds;
    ctx->live_cursor = 0;
    ctx->last_poly_valid = false;          // (these rounds do not maintain the running claim)
    ctx->live_rounds_total += (uint64_t) A.rounds;
    ++ctx->live_phases_total;
    return ZK_OK;
}
// a segment of mid-size rounds (k_mid): `rounds` consecutive rounds in which every present pair keeps at least two quads
static int32_t mid_start(zk_ctx *ctx, const HFr &r, bool with_add_term, int rounds, uint32_t blocks) {
    mid_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.rounds = rounds;
    A.with_add_term = with_add_term ? 1 : 0;
    A.first = ctx->round == 0 ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.partials = ctx->partials;
    A.arrive = ctx->d_counter + 2;
    A.bc = (mid_bcast *) ctx->d_bcast;
    A.out = (tail_out *) ctx->d_tail;
    A.in = (const live_in *) ctx->d_live_in;
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    ((tail_out *) ctx->h_tail)->status = 0;
    ZK_LAUNCH(PC_TAIL, 0.0, k_mid<true>, dim3(blocks), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    if (with_add_term) { ctx->add_term = ctx->add_term * (HFr::one() - r); scale_absorbed(ctx, r); }      // (the kernel's blocks apply the same factors to their copy)
    ctx->live_active = true;
    ctx->live_mid = true;
    ctx->live_first = A.first != 0;
    ctx->live_with_add = with_add_term;
    ctx->live_count = rounds;
    ctx->live_cursor = 0;
    ctx->last_poly_valid = false;
    ctx->live_rounds_total += (uint64_t) rounds;
    ++ctx->live_phases_total;
    return ZK_OK;
}
// the same segment in the non-interactive mode (k_mid<false>): the kernel derives the challenges itself, the host waits once and answers
// the verifier's following calls from the record (like run_device_rounds)
static int32_t run_device_mid(zk_ctx *ctx, const HFr &r, bool with_add_term, int rounds, uint32_t blocks) {
    mid_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    const bool first = ctx->round == 0;
    A.rounds = rounds;
    A.with_add_term = with_add_term ? 1 : 0;
    A.first = first ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.partials = ctx->partials;
    A.arrive = ctx->d_counter + 2;
    A.bc = (mid_bcast *) ctx->d_bcast;
    A.out = (tail_out *) ctx->d_tail;
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    std::memcpy(A.fs_state, ctx->fs_state, 32);
    const unsigned long long seq = ++ctx->tail_seq;
    A.seq = seq;
    ZK_LAUNCH(PC_TAIL, 0.0, k_mid<false>, dim3(blocks), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    volatile unsigned long long *p = &((tail_out *) ctx->h_tail)->seq;
    for (uint64_t spins = 0; *p != seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != seq) { ctx->err = "device rounds were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const tail_out *o = (const tail_out *) ctx->h_tail;
    // what the kernel's blocks did to their copies: add_term (1 - r) per round, one fold per round but the phase's first
    if (with_add_term) {
        ctx->add_term = ctx->add_term * (HFr::one() - r);
        for (int k = 1; k < rounds; ++k) { HFr c; std::memcpy(&c, &o->chal[k - 1], 32); ctx->add_term = ctx->add_term * (HFr::one() - c); }
    }
    const int folds = rounds - (first ? 1 : 0);
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len || folds <= 0) continue;
        t.Vsrc = nullptr;
        if (folds & 1) t.cur ^= 1;
        t.len >>= folds;
        t.live = t.len;
    }
    ctx->tail_active = true;
    ctx->tail_count = rounds;
    ctx->tail_cursor = 0;
    ctx->last_poly_valid = false;
    ctx->tail_rounds_total += (uint64_t) rounds;
    ++ctx->tail_phases_total;
    return ZK_OK;
}
// one round of the resident kernel: r is the verifier's challenge for the previous polynomial (round 0 of the kernel got it as a launch argument)
static int32_t live_round(zk_ctx *ctx, const HFr &r, uint64_t out_abc[12]) {
    static const bool timing = getenv("ZKCNN_TIMING") != nullptr;
    const int k = ctx->live_cursor;
    const double t0 = timing ? now_s() : 0;
    if (timing && k > 0) ctx->live_t_host += t0 - ctx->live_t_exit;
    if (k > 0) {
        live_post(ctx, r, ctx->live_seq32 + (uint32_t) k);
        if (ctx->live_mid && ctx->live_with_add) { ctx->add_term = ctx->add_term * (HFr::one() - r); scale_absorbed(ctx, r); }
    }
    int32_t rc = live_wait(ctx, k, out_abc);
    if (rc) return rc;
    if (timing) {
        ctx->live_t_exit = now_s();
        if (k > 0) { ctx->live_t_gpu += ctx->live_t_exit - t0; ++ctx->live_timed_rounds; }
    }
    ++ctx->round;
    ctx->proof_size += 32 * 3;
    if (++ctx->live_cursor < ctx->live_count) return ZK_OK;
    if (ctx->live_mid) {
        // the segment is over: every table was folded live_count times (nothing collapsed, nothing is down to its last pair)
        for (int b = 0; b < 2; ++b) {
            table_pair &t = ctx->tp[b];
            if (!t.len) continue;
            const int folds = ctx->live_count - (ctx->live_first ? 1 : 0);
            if (folds > 0) {
                t.Vsrc = nullptr;
                if (folds & 1) t.cur ^= 1;
                t.len >>= folds;
                t.live = t.len;
            }
        }
        ctx->live_active = false;
        ctx->live_mid = false;
        return ZK_OK;
    }
    // the phase is over: the kernel has posted the bookkeeping scalar and what is left of the tables before its last polynomial, and leaves
    const tail_out *o = (const tail_out *) ctx->h_tail;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        t.Vsrc = nullptr;
        if (o->pair_state[b] == 3) {
            // a shortened tail: what is left of the pair comes to the host, which runs the phase's remaining rounds (host_tail_round)
            const uint32_t n = o->exp_n[b];
            if (n < 2 || n > TAIL_EXPORT_MAX) { ctx->err = "resident tail: exported table of unexpected length"; return ZK_ERR_STATE; }
            ctx->ht_V[b].resize(n);
            ctx->ht_M[b].resize(n);
            std::memcpy(ctx->ht_V[b].data(), o->exp_V[b], (size_t) n * 32);
            std::memcpy(ctx->ht_M[b].data(), o->exp_M[b], (size_t) n * 32);
            t.len = n;
            t.tail_valid = false;
            if (n == 2) { std::memcpy(&t.tail_v[0], &o->exp_V[b][0], 32); std::memcpy(&t.tail_v[1], &o->exp_V[b][1], 32); t.tail_valid = true; }
            ctx->host_tail_active = true;
        } else if (o->pair_state[b] == 1) {
            t.len = 2;
            std::memcpy(&t.tail_v[0], &o->tail_v[b][0], 32);
            std::memcpy(&t.tail_v[1], &o->tail_v[b][1], 32);
            t.tail_valid = true;
        } else {
            t.len = 0;
            t.absorbed = true;
            t.abs_m_valid = false;
            std::memcpy(&t.final_v, &o->final_v[b], 32);
        }
    }
    std::memcpy(&ctx->add_term, &o->add_term, 32);
    ctx->live_ticks_wait += o->ticks_wait;
    ctx->live_ticks_total += o->ticks_total;
    ctx->live_active = false;
    if (ctx->lane_tail) {
        // lock step again: a lane defers nothing before every lane of the batch has left its tail, and all of them go on in the SAME pass of the batch's
        // driver (the pass after the one in which the last lane left: the driver flushes once per pass, and whatever a lane defers on its own goes out
        // on its own -- behind a tail that is still running it would starve the lanes that kernel waits for: wait_slot spins, it does not yield)
        lane_tail_done(ctx);
        while (ctx->batch && (ctx->batch->tails_running > 0 || ctx->batch->n_flushes < ctx->batch->tails_open_at)) {
            int32_t rcy = zk_batch_sync_point(ctx);
            if (rcy) return rcy;
        }
    }
    return ZK_OK;
}

// Hybrid tail: the (small) live tables come to the host ...
static int32_t host_tail_begin(zk_ctx *ctx) {
    export_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        A.V[b] = t.len ? vin(t) : nullptr;
        A.M[b] = t.len ? t.M[t.cur] : nullptr;
        A.n[b] = (uint32_t) t.len;
    }
    A.out = (export_out *) ctx->d_tail;
    A.seq = ++ctx->tail_seq;
    k_export_f f;
    f.a = A;
    zk_launch_f<k_export_f, 512>(ctx, PC_FOLD, 0.0, dim3(1), f);
    ZK_HIP(hipGetLastError());
    const export_out *o = (const export_out *) ctx->h_tail;
    if (ctx->batch) { int32_t rc = zk_batch_sync_point(ctx); if (rc) return rc; }
    volatile const unsigned long long *p = &o->seq;
    for (uint64_t spins = 0; *p != A.seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != A.seq) { ctx->err = "tables were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int b = 0; b < 2; ++b) {
        const uint64_t n = ctx->tp[b].len;
        ctx->ht_V[b].resize(n);
        ctx->ht_M[b].resize(n);
        if (n) {
            std::memcpy(ctx->ht_V[b].data(), o->V[b], n * 32);
            std::memcpy(ctx->ht_M[b].data(), o->M[b], n * 32);
        }
    }
    ctx->host_tail_active = true;
    return ZK_OK;
}
// ... and one round there: the arithmetic of reference src/prover.cpp:368-426 on plain folded tables (what the round kernels compute)
static void host_tail_round(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    const bool first = ctx->round == 0;
    ++ctx->round;
    if (with_add_term) { ctx->add_term = ctx->add_term * (HFr::one() - r); scale_absorbed(ctx, r); }
    HFr a(0LL), c(0LL), p1(0LL);
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        std::vector<HFr> &V = ctx->ht_V[b], &M = ctx->ht_M[b];
        if (!first) {
            for (uint64_t j = 0; j < t.len / 2; ++j) {
                V[j] = V[2 * j] + r * (V[2 * j + 1] - V[2 * j]);
                M[j] = M[2 * j] + r * (M[2 * j + 1] - M[2 * j]);
            }
            t.len >>= 1;
            t.Vsrc = nullptr;
        }
        if (t.len == 1) {                                   // the reference's `total == 1` case (prover.cpp:400-404)
            t.final_v = V[0];
            ctx->add_term = ctx->add_term + V[0] * M[0];
            t.abs_m = M[0];
            t.abs_m_valid = true;
            t.absorbed = true;
            t.len = 0;
            continue;
        }
        for (uint64_t j = 0; j < t.len / 2; ++j) {
            const HFr &v0 = V[2 * j], &v1 = V[2 * j + 1], &m0 = M[2 * j], &m1 = M[2 * j + 1];
            a = a + (v1 - v0) * (m1 - m0);
            c = c + v0 * m0;
            p1 = p1 + v1 * m1;
        }
        if (t.len == 2) { t.tail_v[0] = V[0]; t.tail_v[1] = V[1]; t.tail_valid = true; }
    }
    HFr bcoef = p1 - a - c;
    if (with_add_term) { bcoef = bcoef - ctx->add_term; c = c + ctx->add_term; }
    put(out_abc, a);
    put(out_abc + 4, bcoef);
    put(out_abc + 8, c);
    ctx->proof_size += 32 * 3;
    ++ctx->host_tail_rounds_total;
}

