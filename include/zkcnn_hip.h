/* C-ABI of the MI355X (gfx950) GKR prover: libzkcnn_hip.so.
 *
 * This is the drop-in boundary for the hot path of TAMUCrypto/zkCNN: every entry point below is
 * the binding of ONE method of the reference's `class prover` (reference src/prover.hpp:18-49) or
 * of the Hyrax `polyProver` the reference links (reference src/prover.cpp:503-511,
 * src/verifier.cpp:128,360), with plain pointers and sizes only. The host-side C++14 class
 * `prover` (zkcnn_amd/csrc/host/prover.{hpp,cpp}) keeps the reference's public C++ API and forwards
 * to these calls, so the unchanged interactive verifier drives the GPU.
 *
 * Conventions
 *   - a field element (BLS12-381 scalar field) is 4 x uint64_t little-endian limbs in MONTGOMERY
 *     form (R = 2^256) -- the in-memory form of the host `Fr`;
 *   - a G1 point is affine, 12 x uint64_t: x then y, 6 Montgomery limbs each (R = 2^384), with
 *     x = y = 0 encoding the point at infinity;
 *   - every call returns 0 on success, a negative zk_status otherwise; zk_last_error() has text;
 *   - calls on one context are not thread safe (neither is the reference prover: file-scope statics,
 *     reference src/prover.cpp:9); use one context per GPU / per proof stream.
 */
#ifndef ZKCNN_HIP_H
#define ZKCNN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;

enum zk_status { ZK_OK = 0, ZK_ERR_HIP = -1, ZK_ERR_ARG = -2, ZK_ERR_STATE = -3, ZK_ERR_NOMEM = -4 };

/* gate records exactly as the reference stores them (reference src/circuit.h:15-33) */
typedef struct { uint32_t g, u; uint8_t lu, sc; uint8_t pad_[2]; } zk_uni_gate;          /* 12 bytes */
typedef struct { uint32_t g, u, v; uint8_t sc, l; uint8_t pad_[2]; } zk_bin_gate;        /* 16 bytes */

/* layerType values (reference src/circuit.h:35-37) */
enum zk_layer_type { ZK_INPUT = 0, ZK_FFT, ZK_IFFT, ZK_ADD_BIAS, ZK_RELU, ZK_SQR, ZK_OPT_AVG_POOL, ZK_MAX_POOL,
                     ZK_AVG_POOL, ZK_DOT_PROD, ZK_PADDING, ZK_FCONN, ZK_NCONV, ZK_NCONV_MUL, ZK_NCONV_ADD };

/* one `layer` of the reference layeredCircuit after initSubset (reference src/circuit.h:39-78) */
typedef struct {
    int32_t ty;
    uint32_t size, size_u[2], size_v[2];
    int8_t bit_length, bit_length_u[2], bit_length_v[2], max_bl_u, max_bl_v, fft_bit_length;
    uint8_t need_phase2;
    uint32_t zero_start_id;
    uint64_t scale[4];
    const zk_uni_gate *uni_gates; uint64_t n_uni;
    const zk_bin_gate *bin_gates; uint64_t n_bin;
    const uint32_t *ori_id_u, *ori_id_v;      /* size_u[0] / size_v[0] entries */
} zk_layer_desc;

/* ---- context ---------------------------------------------------------------------------- */
int32_t zk_ctx_create(int32_t device, zk_ctx **out);
void zk_ctx_destroy(zk_ctx *ctx);
const char *zk_last_error(const zk_ctx *ctx);      /* ctx may be NULL: error of the last failed create */
int32_t zk_device_count(void);

/* ---- residency: prover::C and prover::val (reference src/prover.hpp:48-49) move to HBM ------ */
/* Uploads the circuit, sorts every layer's gates by destination table entry (the CSR order the
 * scatter kernels need) and sizes all work buffers. two_mul = layeredCircuit::two_mul. */
int32_t zk_upload_circuit(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul,
                          int32_t n_two_mul);
/* Optional knowledge about direct-convolution layers (layer type NCONV, reference src/neuralNetwork.cpp:240-275 naiveConvLayerFast):
 * the parameters its gate list was generated from. The upload regenerates the list from them and compares it record by record with
 * the layer's gates; only on a match (and power-of-two channel counts and picture sides) are the layer's two large gate sums
 * (reference src/prover.cpp:224-233, 297-305) computed in factored form -- the same field elements for channel_out * channel_in * m^2
 * products instead of that times the number of positions. A hint that does not match is ignored (generic path). */
typedef struct {
    int32_t layer;
    uint32_t pic_parallel, channel_out, channel_in, nx_in, ny_in, nx_out, ny_out, m, padding, log_stride;
    uint32_t weight_start;     /* raw layer-0 index of weight (co = 0, ci = 0, 0, 0) */
} zk_conv_hint;
int32_t zk_upload_circuit_hinted(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul, int32_t n_two_mul,
                                 const zk_conv_hint *hints, uint32_t n_hints);
/* A second context on the SAME resident circuit holding a copy of src's witness (layer values copied in HBM; the witness program, if the circuit
 * has one, adopted): no host copy of the circuit is needed, nothing is sorted or uploaded. What a caller uses to prove many pictures of one model
 * side by side -- the lanes of a batch below -- where the reference builds circuit and witness per picture (reference src/neuralNetwork.cpp:60-142).
 * The clone is independent of src afterwards (src may be destroyed first: the circuit is reference counted). */
int32_t zk_ctx_clone(zk_ctx *src, zk_ctx **out);
/* Test hook: overwrite ONE resident value (an INVALID witness on purpose: the proof must then be rejected, and -- seeded -- still be
 * byte-identical to the CPU oracle's proof of the same corrupted witness). Keeps the bookkeeping of zk_upload_layer_values exact. */
int32_t zk_poke_layer_value(zk_ctx *ctx, int32_t layer, uint64_t index, const uint64_t value[4]);
/* how many layers run the factored path (after an upload) */
int32_t zk_structured_layers(const zk_ctx *ctx);
/* DOT_PROD layers (FFT convolutions over pic_cnt >= 2 pictures, channel_out a power of two) whose gate list is the generator's full pattern
 * (checked gate by gate at upload): their phase-1 table is built in factored form, summed once over channel_out instead of once per picture */
int32_t zk_factored_dot_layers(const zk_ctx *ctx);
/* DOT_PROD phases so far whose Y table was not folded behind X's live prefix: that region was evaluated once at the end of the phase instead
   (reference src/prover.cpp:103-153 folds it every round; same claims, same transcript) */
uint64_t zk_dot_deferred_phases(const zk_ctx *ctx);
/* val[layer] (n = layer size); stored zero-padded to 2^bit_length */
int32_t zk_upload_layer_values(zk_ctx *ctx, int32_t layer, const uint64_t *values, uint64_t n);

/* ---- prover state machine: one call per reference method ------------------------------------ */
int32_t zk_prover_init(zk_ctx *ctx);                                                   /* prover::init            prover.cpp:17  */
int32_t zk_vres(zk_ctx *ctx, const uint64_t *r, uint32_t output_size, uint32_t r_size, uint64_t out[4]);   /* Vres   :434 */
int32_t zk_sumcheck_init_all(zk_ctx *ctx, const uint64_t *r_0, uint32_t n);            /* sumcheckInitAll         :28  */
int32_t zk_sumcheck_init(zk_ctx *ctx, const uint64_t alpha[4], const uint64_t beta[4]); /* sumcheckInit           :43  */
int32_t zk_sumcheck_dotprod_init_phase1(zk_ctx *ctx);                                  /* sumcheckDotProdInitPhase1 :57 */
int32_t zk_sumcheck_init_phase1(zk_ctx *ctx, const uint64_t relu_rou[4]);              /* sumcheckInitPhase1      :155 */
int32_t zk_sumcheck_init_phase2(zk_ctx *ctx);                                          /* sumcheckInitPhase2      :241 */
int32_t zk_sumcheck_dotprod_update1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abcd[16]);   /* :103 cubic a,b,c,d */
int32_t zk_sumcheck_update1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]);            /* :360 quadratic */
int32_t zk_sumcheck_update2(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]);            /* :364 */
int32_t zk_sumcheck_dotprod_finalize1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_1[4]);  /* :146 */
int32_t zk_sumcheck_finalize1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_0[4], uint64_t claim_1[4]);   /* :459 */
int32_t zk_sumcheck_finalize2(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_0[4], uint64_t claim_1[4]);   /* :473 */
int32_t zk_sumcheck_liu_init(zk_ctx *ctx, const uint64_t *s_u, const uint64_t *s_v, uint32_t n);    /* sumcheckLiuInit :312 */
int32_t zk_sumcheck_liu_update(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]);        /* :385 */
int32_t zk_sumcheck_liu_finalize(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_1[4]);       /* :487 */
/* zero-knowledge mode, masked evaluation claims (no reference counterpart: reference README.md:5 "not fully zero-knowledge"; protocol in
 * zkcnn_amd/csrc/host/zk_mask.hpp (2)). After the LAST update call of a phase: what multiplies each operand in that round, A_b(t) = out[3b] + out[3b+1] t +
 * out[3b+2] t^2 for the operand pairs b = 0, 1 (6 field elements) -- the host adds Z' t (1 - t) sum_b M_b A_b(t) to the polynomial. After
 * zk_sumcheck_finalize1 / zk_sumcheck_dotprod_finalize1: the claims left the prover as claim_b + d_b, phase 2 is initialised from the masked values. */
int32_t zk_sumcheck_tail_pairs(zk_ctx *ctx, uint64_t out[24]);
int32_t zk_sumcheck_claims_adjust(zk_ctx *ctx, const uint64_t d0[4], const uint64_t d1[4]);
uint64_t zk_proof_bytes(const zk_ctx *ctx);                                            /* proof_size counter  prover.hpp:44 */

/* ---- Hyrax commitment of layer 0 (protocol: zkcnn_amd/csrc/hyrax-bls12-381/polyCommit.hpp) ------ */
/* commitInput: prover.cpp:503-511. gens: n_gens = 2^(bl0 - bl0/2) affine points; out_comm: 2^(bl0/2) affine points */
int32_t zk_commit_input(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, uint64_t *out_comm, uint64_t n_rows);
int32_t zk_hyrax_open_init(zk_ctx *ctx, const uint64_t *x, uint32_t n);
int32_t zk_hyrax_open_round(zk_ctx *ctx, uint64_t L[12], uint64_t R[12], uint64_t yL[4], uint64_t yR[4]);
int32_t zk_hyrax_open_fold(zk_ctx *ctx, const uint64_t c[4]);
/* the vector left when the recursion stops (at most `cap` elements are written; *n = its length) */
int32_t zk_hyrax_open_final(zk_ctx *ctx, uint64_t *a, uint32_t cap, uint32_t *n);

/* ---- witness side of the FFT convolution (reference src/neuralNetwork.cpp:937-965, src/utils.cpp:105-145) ------------ */
/* `count` transforms of length 2^logn. Forward (inverse = 0): every input block has in_len = 2^(logn-1) elements and is zero
 * padded, all 2^logn outputs are written. Inverse: 2^logn inputs, scaled by 1/2^logn, the first 2^(logn-1) outputs are written.
 * src / dst are host arrays (calcFFTLayer's val[layer-1] / val[layer]). */
int32_t zk_witness_ntt(zk_ctx *ctx, uint64_t *dst, const uint64_t *src, int32_t logn, int32_t inverse, uint64_t count);
/* calcDotProdLayer: out[(g, t)] = sum over the layer's bin gates (g, u, v) of F[(u, t)] * F[(v, t)], t < 2^fft_bl.
 * F = val[layer-1] (n_in vectors), out = val[layer] (n_out vectors), host arrays. */
int32_t zk_witness_dotprod(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const uint64_t *F, uint64_t n_in, const zk_bin_gate *gates,
                           uint64_t n_gates, int32_t fft_bl);

/* ---- non-interactive mode: device-side Fiat-Shamir rounds (replaces the verifier <-> prover round trips of reference
 * src/verifier.cpp:169-194,214-229,294-302 once a phase's tables are small). `state` points at the 8 words of the host's BLAKE2s
 * challenge chain (host/replay.hpp: fiatShamir), `pending` at its count of message bytes not yet hashed; both must stay valid until
 * zk_fs_attach(ctx, NULL, NULL). While attached, zk_sumcheck_update1/2 and zk_sumcheck_liu_update run all remaining rounds of a phase in
 * one kernel as soon as the live tables have at most 2^12 entries, answer the following calls from that record and fail with
 * ZK_ERR_STATE if a challenge passed in differs from the one the kernel derived. ---- */
int32_t zk_fs_attach(zk_ctx *ctx, const uint32_t *state, const uint64_t *pending);
/* Hybrid tail: tables of at most 2^log_entries entries (log_entries <= 8) are copied to the host when a phase reaches them and its last
 * rounds run there -- O(2^log_entries) host multiplications per round instead of a latency-bound launch and hand-over. Same field elements
 * either way. log_entries = -1 (the state of a new context): the library's default -- off for a context on its own (its small rounds run in
 * the resident kernels), 2^5 for a lane of a lock-step batch; log_entries <= -2: off, every round of every phase is a kernel. */
int32_t zk_set_host_tail(zk_ctx *ctx, int32_t log_entries);
/* reusable = 0: the generators handed to zk_commit_input from now on are used by ONE proof only (the reference's semantics: its verifier draws new
 * random generators for every proof, reference src/verifier.cpp:119-128) -- no byte table is ever built for them, whatever the proof does with the
 * set; reusable != 0 (the state of a new context): a set that is used a second time gets its byte table. */
int32_t zk_set_generator_reuse(zk_ctx *ctx, int32_t reusable);
/* out[i] = scalars[i] * B for ONE base point B given as its byte-window table, table_affine[w * 255 + d - 1] = d 256^w B (32 x 255 affine points, 12 words
 * each, Montgomery form like the generators of zk_commit_input); scalars: n canonical integers of 4 words; out: n Jacobian points of 18 words (the host
 * library's G1). This is the reference verifier's generator loop, src/verifier.cpp:121-126 (gens[i] = G * k_i, k_i from its CSPRNG), which an in-process run
 * in the reference's semantics repeats for every proof: 4 096 scalar multiplications, ~10 ms of eight host threads per proof -- with eight proofs driven by
 * one host thread (zk_batch_*) that loop was a third of a batch proof's wall time. The table is kept on the GPU from its first use. */
int32_t zk_fixed_base_mul(zk_ctx *ctx, const uint64_t *table_affine, uint64_t table_points, const uint64_t *scalars, uint64_t n, uint64_t *out_jacobian);
/* rounds the hybrid tail has run on the host since the context was created (bench: rounds_on_host_per_proof) */
int32_t zk_host_tail_stats(zk_ctx *ctx, uint64_t *rounds);
/* Persistent rounds of the interactive protocol (on by default): once the live tables of a phase hold at most 1024 quads, ONE resident
 * single-workgroup kernel runs all remaining rounds of the phase; zk_sumcheck_update1/2 and zk_sumcheck_liu_update then exchange a round
 * polynomial and the verifier's next challenge with it through two mapped host mailboxes instead of launching a kernel per round
 * (reference src/prover.cpp:368-426 is called once per round either way: same field elements). on = 0: every round is a launch. */
int32_t zk_set_live_rounds(zk_ctx *ctx, int32_t on);
/* Optional bracket around one proof (the session driver calls it; the reference's own main does not need to). Between begin and end the
 * context counts as ACTIVE on its device. The resident round kernels (zk_set_live_rounds) are used only by a proof that is alone on its GPU
 * when it begins: their workgroups wait for one another and for the host, and a resident kernel holds its hardware queue for the rest of a
 * phase -- with several proofs in flight a launch per round interleaves better (measured: 97 vs 93 proofs/s with 8 in flight). */
int32_t zk_proof_begin(zk_ctx *ctx);
int32_t zk_proof_end(zk_ctx *ctx);
/* rounds / phases served by the tail kernel since the context was created (tests, bench) */
int32_t zk_fs_stats(zk_ctx *ctx, uint64_t *rounds, uint64_t *phases);

/* ---- lock-step batches: K proofs on one resident circuit, ONE launch per round for all of them ------------------------------------------
 * The reference proves one picture per process (reference src/main_demo_vgg.cpp:20-43); its prover is driven round by round by the verifier
 * (reference src/verifier.cpp:149-264), one call of sumcheckUpdate1/2 (reference src/prover.cpp:360-426) per round. K contexts that hold the
 * SAME resident circuit (same upload: they share its gate lists) with K different witnesses make exactly the same sequence of calls with
 * different challenges, so their launches can be fused: a context attached to a batch becomes a LANE of it, moves onto the batch's stream,
 * and every launch of a lane that has a batched kernel form (the round kernels first of all) is deferred until all lanes have reached the
 * same point, then issued as one launch over all lanes (blockIdx.y = lane). Results, transcripts and every zk_sumcheck_* signature are
 * unchanged -- a lane is driven through the same per-context calls above.
 *
 * Driving model: the lanes of a batch are driven by ONE host thread that runs their K protocol loops cooperatively (the host library's
 * batchSession does so with one fiber per lane). The yield function is how a lane hands the thread on: the library calls it wherever a lane
 * has deferred a launch or would wait for the GPU; the driver runs the other lanes up to their own such points, calls zk_batch_flush and
 * resumes them. Without a yield function each lane flushes for itself (correct, nothing is fused). Resident round kernels
 * (zk_set_live_rounds) and the device-side Fiat-Shamir chain (zk_fs_attach) are not used by a lane. */
typedef struct zk_batch zk_batch;
#define ZK_BATCH_MAX_LANES 8
typedef void (*zk_yield_fn)(void *user, int32_t lane);
int32_t zk_batch_create(int32_t device, zk_batch **out);
/* detaches every lane (each goes back to its own stream) and frees the batch */
void zk_batch_destroy(zk_batch *b);
/* ctx becomes the next lane (its index is returned in *lane, may be NULL). ZK_ERR_ARG: other device, already a lane, batch full;
 * ZK_ERR_STATE: no circuit yet, or not the resident circuit of lane 0. */
int32_t zk_batch_attach(zk_batch *b, zk_ctx *ctx, int32_t *lane);
int32_t zk_batch_detach(zk_batch *b, zk_ctx *ctx);
int32_t zk_batch_set_yield(zk_batch *b, zk_yield_fn fn, void *user);
/* issues every deferred launch: one launch per kernel class over all lanes that deferred one */
int32_t zk_batch_flush(zk_batch *b);
/* out[0] = launches issued by flushes, out[1] = lane launches they stood for, out[2] = flushes, out[3] = lanes */
int32_t zk_batch_stats(const zk_batch *b, uint64_t out[4]);
const char *zk_batch_last_error(const zk_batch *b);

/* ---- zero-knowledge mode of the commitment (no reference counterpart: reference README.md:5 "not fully
 * zero-knowledge"): the generator set has one more entry H = gens[n_gens - 1] and every commitment is blinded, Com(v; s) = <v, g> + s H -- */
/* zk_commit_input with n_gens = 2^cb + 1 generators and one blinding factor per row: out_comm[i] = <row_i, g> + blinds[i] H */
int32_t zk_commit_input_blinded(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, const uint64_t *blinds, uint64_t *out_comm, uint64_t n_rows);
/* blinded row commitments of a HOST matrix (n_rows x cols scalars) over the generators of the last zk_commit_input[_blinded] call
 * (cols = their number without H): masking-polynomial coefficients, the random vector of a proof of dot product. blinds may be NULL. */
int32_t zk_commit_vector(zk_ctx *ctx, const uint64_t *scalars, uint64_t n_rows, uint64_t cols, const uint64_t *blinds, uint64_t *out);
/* w = L^T Z of the committed input, L = eq(x[cb..n)): the vector a proof of dot product opens (2^cb elements to the host) */
int32_t zk_hyrax_combine_rows(zk_ctx *ctx, const uint64_t *x, uint32_t n, uint64_t *out_w);

/* ---- verifier side: wiring predicates over the resident gate lists (reference src/verifier.cpp:36-116) --------------------- */
/* Everything betaInitPhase1/2 + predicatePhase1/2 produce for layer `layer`: uni[0..1] (already multiplied by beta_v[0] when the
 * layer has a phase 2) and bin[0..2] (indexed by binGate::l). r_0 / r_1 / alpha / beta: the layer's incoming claim; r_u / r_v: this
 * layer's sumcheck points; r_u2 / r_v2: the points of layer + 2 (PADDING and DOT_PROD only, may be NULL otherwise). The tables
 * are rebuilt from these arguments in verifier-owned buffers; nothing of the prover's state is read except the circuit. */
int32_t zk_verifier_predicates(zk_ctx *ctx, int32_t layer, const uint64_t *r_0, const uint64_t *r_1, const uint64_t alpha[4],
                               const uint64_t beta[4], const uint64_t relu_rou[4], const uint64_t *r_u, const uint64_t *r_v,
                               const uint64_t *r_u2, const uint64_t *r_v2, uint64_t uni[8], uint64_t bin[12]);
/* Layer-0 check (reference src/verifier.cpp:304-325): out = sum over layers i = 1..n of
 * sum_j eq(r_u0, ori_id_u[j]) sig_u[i-1] eq(r_u[i], j) + the same for v. r_u / r_v: n + 1 pointers indexed by layer (entry 0 unused). */
int32_t zk_verifier_input_predicate(zk_ctx *ctx, const uint64_t *r_u0, const uint64_t *const *r_u, const uint64_t *const *r_v,
                                    const uint64_t *sig_u, const uint64_t *sig_v, uint32_t n, uint64_t out[4]);

/* The multi-scalar multiplications of the verifier's opening check (reference src/verifier.cpp:360 -> hyrax polyVerifier::verify):
 * out = sum scalars[i] * bases[i] (affine, C-ABI layout). bases_are_generators != 0: `bases` must be a prefix of the generator set of the
 * last zk_commit_input* call (ZK_ERR_STATE otherwise) and the resident tables serve it; == 0: arbitrary points (the commitment rows), through a
 * second, verifier-owned table set. */
int32_t zk_verifier_msm(zk_ctx *ctx, uint64_t out[12], const uint64_t *scalars, const uint64_t *bases, uint64_t n, int32_t bases_are_generators);

/* ---- witness of a generic layer (reference src/neuralNetwork.cpp:918-935, calcNormalLayer) -------------------------------- */
/* Layer 0 is built piecewise while the circuit is generated (weights, then the bit / sign / max witnesses of every RELU and pooling
 * layer): the context keeps a device copy; call this with every span the host has written since the last call (no holes:
 * offset <= entries held so far; offset 0 starts a new layer 0). */
int32_t zk_witness_input(zk_ctx *ctx, uint64_t offset, const uint64_t *values, uint64_t n);
/* out[g] = scale * ( sum over uni gates in_lu[u] * two_mul[sc]  +  sum over bin gates in_U[u] * in_V[v] * two_mul[sc] ), g < n_out.
 * Operands in layer 0 are read from the device copy above, operands in the previous layer from `prev` (host, n_prev entries);
 * uni.lu != 0 means "previous layer", bin.l as in reference src/circuit.h:31-32; prev == NULL: the previous layer is layer 0 itself
 * (layer 1). out is a host array. */
int32_t zk_witness_gates(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const zk_uni_gate *uni, uint64_t n_uni, const zk_bin_gate *bin,
                         uint64_t n_bin, const uint64_t *prev, uint64_t n_prev, const uint64_t *two_mul, uint32_t n_two_mul,
                         const uint64_t scale[4]);

/* ---- the next picture on a resident circuit (replaces the host loops of reference src/neuralNetwork.cpp:60-142,
 * 652-728, 899-977 for every picture after the first). The host generator records HOW layer 0's auxiliary witnesses follow from the
 * layer values (host/neuralNetwork.hpp: witnessProgram); the context keeps that program, the gate lists grouped by output and all
 * layer values in HBM and replays it for a new picture: evaluate a layer, derive bits / signs / maxima into layer 0, scan a range. ---- */
typedef struct { uint32_t src, dst; uint8_t src_layer, op, shift, pad_; } zk_witness_op;     /* op: 0 bit, 1 sign, 2 running max, 3 bit of a window sum */
typedef struct {
    int32_t what;            /* 0 = auxiliary operations, 1 = evaluate `layer`, 2 = range of `layer` */
    int32_t layer;
    uint64_t op_begin, op_end, win_begin;
    int32_t win, bits, scale_index, pad_;
} zk_witness_step;
/* After zk_upload_circuit, with the SAME layer descriptors (gate lists as uploaded, ori_id_u / ori_id_v). Validates every index. */
int32_t zk_witness_program_upload(zk_ctx *ctx, const zk_witness_op *ops, uint64_t n_ops, const uint32_t *windows, uint64_t n_windows,
                                  const zk_witness_step *steps, uint32_t n_steps, const zk_layer_desc *layers, int32_t n_layers);
/* Writes `picture` (n_picture field elements) over the start of layer 0 and recomputes every layer's values and layer 0's auxiliary
 * witnesses in HBM. ranges[2k], ranges[2k+1] = largest non-negative value / largest magnitude of a negative value of the k-th
 * range step (the host turns them into quantisation scales and compares with the circuit's); last_layer (may be NULL) receives the
 * first n_last values of the output layer. ZK_ERR_STATE if a value exceeded 63 bits (the int64 view of the reference would differ). */
int32_t zk_witness_rerun(zk_ctx *ctx, const uint64_t *picture, uint64_t n_picture, uint64_t *ranges, uint32_t n_ranges,
                         uint64_t *last_layer, uint64_t n_last);
/* Convolutions whose gate pattern was checked at upload are evaluated from their two tensors, in 64-bit integers when a device-side bound test
 * shows the sums cannot wrap (quantised activations and weights are small integers), in field arithmetic otherwise; both give the same field
 * elements. Test hook: out[0] = such convolutions, out[1] = those the last zk_witness_rerun evaluated in integers, out[2] = index in layer 0 of the first one's weights; force_field = 1 / 0 keeps
 * every convolution in the field / lets the bounds decide from now on, < 0 changes nothing. */
int32_t zk_witness_conv_paths(zk_ctx *ctx, int32_t force_field, uint64_t out[3]);

/* frees the device copy of layer 0 and the staging buffers of zk_witness_* (the witness is complete) */
int32_t zk_witness_release(zk_ctx *ctx);

/* ---- built-in profiler: HIP events around every launch of the selected kernel classes, on the context's stream ---- */
/* class_mask: bit i selects class i of zk_profile_report's list; 0 switches profiling off; ~0u selects all */
int32_t zk_profile_enable(zk_ctx *ctx, uint32_t class_mask);
/* drains pending events and writes a JSON object {"class": {"ms": total, "launches": n, "bytes": algorithmic}, ...};
 * reset != 0 clears the counters afterwards */
int32_t zk_profile_report(zk_ctx *ctx, char *buf, uint64_t cap, int32_t reset);

/* ---- kernel-level entry points (host arrays in/out; used by the parity tests and the micro-benches) ---- */
int32_t zk_k_fr_mul(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n);
int32_t zk_k_fr_add(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n);
int32_t zk_k_fr_sub(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n);
int32_t zk_k_eq_table(zk_ctx *ctx, uint64_t *out, int32_t n, const uint64_t *r0, const uint64_t *r1,
                      const uint64_t alpha[4], const uint64_t beta[4]);
int32_t zk_k_phi_table(zk_ctx *ctx, uint64_t *out, const uint64_t *rx, const uint64_t scale[4], int32_t n, int32_t inverse);
/* one quadratic sumcheck round on two n-entry tables: fold with r unless `first`, then round polynomial.
 * V / M are updated in place to the folded tables; returns the folded length in *n_out. */
int32_t zk_k_round_quadratic(zk_ctx *ctx, uint64_t *V, uint64_t *M, uint64_t n, const uint64_t r[4], int32_t first,
                             uint64_t out_abc[12], uint64_t *n_out);
int32_t zk_k_msm(zk_ctx *ctx, uint64_t out[12], const uint64_t *scalars, const uint64_t *bases, uint64_t n);
/* out[i] = in[i]^-1 in the base field Fp (6 words each, Montgomery form, R = 2^384; the inverse of 0 is 0): the inversion the table kernels use */
int32_t zk_k_fp_inv(zk_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t n);
/* row-cooperative base-field arithmetic (one element per 16-lane row; hip/fpc_dev.cuh -- the arithmetic of the commitment's reduction trees, in place of
 * mcl's Fp inside the absent hyrax-bls12-381): out[i] = a[i] b[i], out[n + i] = a[i] + b[i], out[2n + i] = a[i] - b[i]; 6 words per element, Montgomery form */
int32_t zk_k_fpc_ops(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n);
/* ... point arithmetic in that form: out[i] = p[i] + q[i] (infinity, p = q, p = -q handled), out[n + i] = 2 p[i]; Jacobian points of 18 words (Z = 0: infinity) */
int32_t zk_k_cl_add(zk_ctx *ctx, uint64_t *out, const uint64_t *p, const uint64_t *q, uint64_t n);
/* ... and the reduction tree of the MSM kernels: out[s] = sum of the n_seg points of segment s, runs of n_in points per 512-thread block and level */
int32_t zk_k_cl_tree(zk_ctx *ctx, uint64_t *out, const uint64_t *pts, uint64_t n_seg, uint64_t segs, uint32_t n_in);
/* the commitInput data path on caller data: `rows` Pedersen commitments of `cols` scalars each over `cols` bases */
int32_t zk_k_commit_rows(zk_ctx *ctx, uint64_t *out, const uint64_t *scalars, const uint64_t *bases, uint64_t rows, uint64_t cols);
/* device-resident micro-benchmarks: seconds per launch, averaged over `iters` launches with HIP events */
int32_t zk_bench_fr_mul(zk_ctx *ctx, uint64_t n_threads, uint32_t muls_per_thread, uint32_t iters, double *sec);
int32_t zk_bench_round_quadratic(zk_ctx *ctx, uint32_t log_n, uint32_t iters, double *sec_per_launch,
                                 double *algorithmic_bytes);
int32_t zk_bench_copy(zk_ctx *ctx, uint64_t bytes, uint32_t iters, double *sec);

#ifdef __cplusplus
}
#endif
#endif
