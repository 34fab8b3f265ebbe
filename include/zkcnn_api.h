/* Whole-proof driver C-ABI (host side): build a CNN circuit + witness, run the interactive
 * GKR protocol (verifier <-> prover) and return timings, acceptance and the canonical transcript.
 *
 * It replaces what the reference does in main(): reference src/main_demo_lenet.cpp:19-41 and
 * src/main_demo_vgg.cpp:20-43 (model ctor -> nn.create(p,false) -> verifier v(&p,p.C) -> v.verify()
 * -> 16-column result row), as a library call so that Python tests / bench.py (ctypes) and C/C++
 * callers share one entry point. The product library (libzkcnn_host.so) runs the prover on the
 * GPU through include/zkcnn_hip.h; oracle/liboracle.so exports the same three functions with the
 * `oracle_` prefix, backed by the CPU restatement (test infrastructure only).
 */
#ifndef ZKCNN_API_H
#define ZKCNN_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const char *model;     /* "lenet", "lenet.avg", "lenetCifar", "vgg11", "vgg16", "vgg:<tokens>", "custom:<spec>" */
    int32_t pic_x, pic_y, pic_channel;
    int32_t pic_cnt;       /* pictures folded into ONE circuit (reference pic_parallel) */
    uint64_t data_seed;    /* synthetic picture / weights / biases (reference data.tar.gz is absent) */
    uint64_t picture_seed; /* 0: the picture comes from the data_seed stream; else from its own stream, so that sessions with one data_seed share
                              their weights and differ in the picture (zkcnn_session_new_image switches a session to another picture_seed) */
} zkcnn_model_desc;

#define ZKCNN_MODE_VERIFY      0u  /* full verifier (reference behaviour): challenges from the operating system's CSPRNG, fresh random generators */
#define ZKCNN_MODE_DRIVE_ONLY  1u  /* same challenges and prover calls, verifier checks skipped */
#define ZKCNN_MODE_REUSE_GENS  2u  /* public commitment generators (hash-to-curve, nobody knows a discrete log) instead of fresh random multiples of G */
#define ZKCNN_MODE_TAMPER      4u  /* test hook: the verifier corrupts message number ((mode >> 8) & 0xffff) before checking it; flags continue at bit 24 */
#define ZKCNN_MODE_HOST_PRED   8u  /* verifier's wiring predicates on the host (reference src/verifier.cpp:89-116) instead of the GPU */
#define ZKCNN_MODE_FIAT_SHAMIR 32u /* non-interactive: challenges are a BLAKE2s chain over the statement and every message so far (the seed is ignored), derived by
                                     the verifier object on the host while the rounds run as in the interactive protocol (ZKCNN_MODE_FS_DEVICE: on the GPU);
                                     always on the public hash-to-curve generators, whose digest is part of the hashed statement */
#define ZKCNN_MODE_SEEDED      64u /* reproducible run for parity tests / benches: challenges (and, without REUSE_GENS, generator scalars) come from a
                                     xoshiro stream seeded with challenge_seed. NOT secure -- every challenge is predictable from the seed.
                                     zkcnn_session_verify accepts an interactive (non-Fiat-Shamir) transcript only with this bit: a replay against a
                                     known challenge stream is a debugging aid, never evidence that a statement is true */
#define ZKCNN_MODE_FULL_IPA  128u  /* inner-product argument down to length 1 (log2(m) rounds) instead of sending the last 256 scalars in the clear */
#define ZKCNN_MODE_ZK  (1u << 24)  /* zero-knowledge mode: blinded commitments, masked round polynomials, masked evaluation claims, proofs of dot product.
                                     It fixes the host tail itself (a phase's last <= 32 entries: the masks' share of the last round polynomial is added on the host):
                                     ZKCNN_MODE_HOST_TAIL next to it changes nothing, together with ZKCNN_MODE_GPU_TAIL the call is refused (-4) */
#define ZKCNN_MODE_HOST_ROUNDS (1u << 25)  /* every sumcheck round is a kernel launch driven from the host: no resident round kernel (interactive), no device-side
                                              rounds (Fiat-Shamir) -- A/B and parity of both */
#define ZKCNN_MODE_HOST_TAIL (1u << 26)    /* hybrid tail (off by default): once a phase's tables have <= 64 entries they travel to the host and its last
                                            ~6 rounds run there (a few hundred host multiplications instead of six latency-bound launches) */
#define ZKCNN_MODE_GPU_TAIL (1u << 28)     /* lanes of a lock-step batch: no hybrid tail either (their default: tables of <= 32 entries finish on the host,
                                            * include/zkcnn_hip.h: zk_set_host_tail) -- every round of every phase is a kernel launch */
#define ZKCNN_MODE_FS_DEVICE (1u << 27)    /* with ZKCNN_MODE_FIAT_SHAMIR: the GPU derives the challenges of the small and mid-size rounds itself
                                            (BLAKE2s chain on the device) instead of trading polynomials and challenges with the
                                            host, which hashes by default. Same transcript; slower since round 3 (the resident kernels' mailbox
                                            round trip costs less than the serial hash on four lanes): kept as an option and for the tests */
#define ZKCNN_MODE_CROSS_PRED 16u  /* both, and the verifier rejects if they differ (parity check of zk_verifier_*) */

typedef struct {
    int32_t accepted;          /* 1 = "Verification pass" + Hyrax opening ok, 0 = rejected, -1 = not checked */
    int32_t n_layers;
    uint64_t input_size;       /* layer-0 size (witness size column) */
    int32_t input_bits;
    int32_t n_rounds;          /* prover round-polynomial calls */
    int32_t n_messages;        /* sumcheck-phase messages received (round polynomials + claims); commitment messages follow */
    int32_t reserved_;
    double prove_s;            /* prover::proveTime()            (column PT) */
    double poly_prove_s;       /* prover::polyProverTime()       (column POLY_PT); TOT_PT = sum of the two */
    double verify_s;           /* verifier time                  (column VT) */
    double poly_verify_s;
    double proof_kb, poly_proof_kb;
    double witness_s;          /* circuit + witness generation (not prover time, as in the reference) */
    double upload_s;           /* host -> HBM transfer of circuit and witness (0 for the oracle) */
    double wall_s;             /* wall clock of the whole prove call */
    uint64_t transcript_len;   /* bytes of canonical transcript produced */
    uint64_t gate_cnt_uni, gate_cnt_bin, table_entries;   /* circuit statistics */
    char message[128];         /* failure reason, if any */
} zkcnn_result;

/* Builds the circuit and the witness (and, for the product library, uploads both to `device`).
 * Returns NULL on error. */
void *zkcnn_session_create(const zkcnn_model_desc *desc, int32_t device);
/* Calibrated session: the same, but under GIVEN quantisation scales (zkcnn_session_statement of an earlier session of the model) instead of
 * the ones this picture's ranges would give -- every picture of a calibrated model then has the same circuit. NULL if the picture's values
 * do not fit the scales. */
void *zkcnn_session_create_calibrated(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t device);
/* Product library only -- a second session on the SAME resident circuit, with a copy of `session`'s witness (in HBM) and NO host copy of the circuit:
 * nothing is generated, sorted or uploaded (a vgg11 session costs 3.5 s and 2.7 GB of host memory to build; a clone ~0.1 s and ~0.1 GB). Proves what
 * `session` proves until zkcnn_session_new_image gives it a picture of its own: how the lanes of a batch are made. Independent of `session` afterwards
 * (either may be destroyed first). The clone keeps no gate lists on the host: the verifier's wiring predicates run on the GPU; ZKCNN_MODE_HOST_PRED /
 * ZKCNN_MODE_CROSS_PRED are rejected for it. NULL on error (a verifier-only session cannot be cloned). Several threads may clone one session at the
 * same time; the session must not be proving or taking a new picture meanwhile. */
void *zkcnn_session_clone(void *session);
/* One proof. `transcript` may be NULL; at most `cap` bytes are written, the full length is reported. */
int32_t zkcnn_session_prove(void *session, uint64_t challenge_seed, uint32_t mode, uint8_t *transcript,
                            uint64_t cap, zkcnn_result *out);
/* Checks a serialized proof (the transcript zkcnn_session_prove returned) without running the prover: the verifier replays the
 * bytes against this session's circuit. Only Fiat-Shamir proofs (ZKCNN_MODE_FIAT_SHAMIR) are PROOFS off line; an interactive
 * transcript is rejected unless ZKCNN_MODE_SEEDED says "replay it against the seeded stream it was driven with" (parity / debugging).
 * out->accepted = 1 / 0, out->message = the reason for a rejection. */
int32_t zkcnn_session_verify(void *session, uint64_t challenge_seed, uint32_t mode, const uint8_t *proof, uint64_t len,
                             zkcnn_result *out);
/* The statement of a session's circuit besides the model descriptor: the quantisation scales (bits kept per layer) the circuit's
 * shape depends on. Returns their number (also when it exceeds `cap`; at most `cap` are written). */
int64_t zkcnn_session_statement(void *session, int32_t *scales, uint64_t cap);
/* A verifier-only session: the circuit rebuilt from the model descriptor (data_seed ignored) and a statement, without picture,
 * weights or witness; zkcnn_session_verify works on it, zkcnn_session_prove fails. Returns NULL on error (also on a statement
 * that does not fit the model). */
void *zkcnn_verifier_create(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t device);
/* Product library only -- the next picture on the session's resident circuit: `pixels` (channel, x, y order, n_pixels =
 * channel * x * y values; NULL: the synthetic picture of `picture_seed`) is quantised on the host and every layer value and auxiliary
 * witness is recomputed in HBM by replaying the witness program the circuit generator recorded -- no circuit generation, no upload of
 * values. The circuit's shape depends on the picture through its quantisation scales; returns
 *   0  done: the session now proves the new picture (same transcript as a calibrated session created for it from scratch: every value
 *      range fits the circuit's scale -- the picture's own scales are equal or larger),
 *   1  this picture's own range does not fit the input scale -- nothing was changed,
 *   2  some layer's activation range does not fit its scale -- refused; the witness of the picture the session proved before is put
 *      back (a second replay), so the session keeps proving THAT picture (create a new session for the refused one),
 *  <0  error. *ms (may be NULL) receives the wall-clock milliseconds of the call. */
int32_t zkcnn_session_new_image(void *session, uint64_t picture_seed, const double *pixels, uint64_t n_pixels, double *ms);
/* the pixel values of synthetic picture `picture_seed` for this session's model (cap >= channel * x * y); returns their number */
int64_t zkcnn_session_synthetic_picture(void *session, uint64_t picture_seed, double *pixels, uint64_t cap);
/* ---- lock-step batches (product library; the oracle exports oracle_batch_prove over its CPU sessions for the host-logic tests) ----
 * K sessions of ONE model on ONE GPU (same data_seed => same weights => one resident circuit; their pictures differ: picture_seed, or
 * zkcnn_session_new_image) become the lanes of a batch: one host thread drives their K verifier loops together and every sumcheck round of
 * the K proofs is ONE kernel launch (include/zkcnn_hip.h: zk_batch_*). The sessions stay owned by the caller, must outlive the batch and must
 * not be proved individually from other threads while it exists. n <= 8. NULL on error (a session on another device or circuit). */
void *zkcnn_batch_create(void *const *sessions, int32_t n);
/* n_batches batches at once -- sessions[0 .. counts[0]) the lanes of the first, the next counts[1] of the second, ... -- with every batch's stream created
 * before any lane is attached, so that the streams spread evenly over the hardware queues (streams that share one run their kernels one after the other:
 * the step time of B batches is the most loaded queue's). out[j] = the handles (zkcnn_batch_prove / _stats / _destroy as for zkcnn_batch_create). 0 = ok. */
int32_t zkcnn_batch_create_group(void *const *sessions, const int32_t *counts, int32_t n_batches, void **out);
/* One proof per lane: seeds[k], transcripts[k] (may be NULL) / caps[k], out[k] as in zkcnn_session_prove; `mode` is shared. Each lane's
 * transcript is byte for byte what zkcnn_session_prove(sessions[k], seeds[k], mode) returns. The per-lane timers in out[k] (prove_s ...)
 * include the time the thread spent in other lanes; *wall_s (may be NULL) is the wall clock of the whole batch. */
int32_t zkcnn_batch_prove(void *batch, const uint64_t *seeds, uint32_t mode, uint8_t *const *transcripts, const uint64_t *caps,
                          zkcnn_result *out, double *wall_s);
/* out[0] = fused launches, out[1] = lane launches they stood for, out[2] = flushes, out[3] = lanes, out[4] = passes of the driver loop */
int32_t zkcnn_batch_stats(void *batch, uint64_t out[5]);
void zkcnn_batch_destroy(void *batch);

/* Test hooks (both libraries): size of layer `layer` of the session's circuit (its layerType in *type), and overwriting one value of the
 * witness -- an INVALID witness on purpose: proofs must then be rejected, and a seeded GPU proof must still equal the oracle's. */
int64_t zkcnn_session_layer_size(void *session, int32_t layer, int32_t *type);
int32_t zkcnn_session_poke(void *session, int32_t layer, uint64_t index, const uint64_t value[4]);
void zkcnn_session_destroy(void *session);
/* The reference CLI's 16-column result row of the last prove call ("a, b, c, ..."), NUL terminated. */
int32_t zkcnn_session_row(void *session, char *buf, uint64_t cap);

/* Product library only, test hook (include/zkcnn_hip.h: zk_witness_conv_paths): out[0] = convolutions new_image evaluates from their tensors,
 * out[1] = those the last new_image evaluated in 64-bit integers, out[2] = index in layer 0 of the first one's weights; force_field = 1 / 0 / < 0: field arithmetic for all / by the bounds / unchanged. */
int32_t zkcnn_session_conv_paths(void *session, int32_t force_field, uint64_t out[3]);
/* Product library only: sumcheck rounds / phases the GPU has run by itself in Fiat-Shamir mode so far (include/zkcnn_hip.h: zk_fs_attach) */
int32_t zkcnn_session_fs_stats(void *session, uint64_t *rounds, uint64_t *phases);
/* Product library only: sumcheck rounds the hybrid tail has run on the host so far (include/zkcnn_hip.h: zk_set_host_tail) */
int32_t zkcnn_session_host_tail_rounds(void *session, uint64_t *rounds);

/* Product library only: how many direct-convolution layers of the session's circuit run the factored gate sums (include/zkcnn_hip.h: zk_conv_hint) */
int32_t zkcnn_session_structured_layers(void *session);
/* ... and DOT_PROD layers of FFT convolutions over several pictures whose phase-1 table is built in factored form (zk_factored_dot_layers) */
int32_t zkcnn_session_factored_dot_layers(void *session);
/* ... and DOT_PROD phases proved so far with the deferred evaluation of Y's dead region (zk_dot_deferred_phases) */
uint64_t zkcnn_session_dot_deferred_phases(void *session);

/* Product library only: HIP-event profiler of the GPU kernels (see zk_profile_enable / zk_profile_report in zkcnn_hip.h). */
int32_t zkcnn_session_profile(void *session, uint32_t class_mask);
int32_t zkcnn_session_profile_report(void *session, char *buf, uint64_t cap, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
