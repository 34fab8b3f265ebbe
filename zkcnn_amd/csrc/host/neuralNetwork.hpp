// CNN -> layered circuit + witness generator (host side, not part of the timed prover path).
// Public shape follows reference src/neuralNetwork.hpp:55-168 (constructor arguments, create(),
// conv_section / pool / full_conn topology tables that the model classes fill); the reference's
// data files are absent (data.tar.gz is a missing blob), so next to the file reader there is a
// seeded synthetic source that produces the same value counts in the same order.
#pragma once
#include <stdexcept>
#include <memory>
#include "circuit.h"
#include "utils.hpp"

enum convType { FFT, NAIVE, NAIVE_FAST };
enum poolType { AVG, MAX, NONE };
enum actType { RELU_ACT };

struct convKernel {
    convType ty;
    i64 channel_out, channel_in, size, stride_bl, padding, weight_start_id, bias_start_id;
    convKernel(convType t, i64 co, i64 ci, i64 ksize, i64 log_stride, i64 pad)
            : ty(t), channel_out(co), channel_in(ci), size(ksize), stride_bl(log_stride), padding(pad),
              weight_start_id(0), bias_start_id(0) {}
    convKernel(convType t, i64 co, i64 ci, i64 ksize) : convKernel(t, co, ci, ksize, 0, ksize >> 1) {}
};

struct fconKernel {
    i64 channel_out, channel_in, weight_start_id, bias_start_id;
    fconKernel(i64 co, i64 ci) : channel_out(co), channel_in(ci), weight_start_id(0), bias_start_id(0) {}
};

struct poolKernel {
    poolType ty;
    i64 size, stride_bl;
    poolKernel(poolType t, i64 sz, i64 log_stride) : ty(t), size(sz), stride_bl(log_stride) {}
};

// Source of the real-valued picture / weights / biases, consumed strictly in reference file order:
// picture (c,x,y); then per layer conv weight (co,ci,x,y), conv bias (co); ...; fc weight (co,ci), fc bias.
class dataSource {
public:
    enum kind { PICTURE, WEIGHT, BIAS };
    virtual ~dataSource() {}
    virtual double next(kind k, i64 fan_in) = 0;
};

// Optional offload of the two witness loops that belong to the FFT-convolution hot path:
// the batched NTT of the FFT / IFFT layers and the per-frequency channel contraction of the DOT_PROD layer.
// Without one the host loops run (that is what the CPU oracle session uses). Results are exact field
// elements either way.
class witnessAccel {
public:
    virtual ~witnessAccel() {}
    // count transforms of length 2^logn; forward: half-length inputs, full outputs; inverse: full inputs, half outputs
    virtual bool ntt(F *dst, const F *src, int logn, bool inverse, size_t count) = 0;
    virtual bool dotProd(F *out, size_t n_out, const F *in, size_t n_in, const binGate *gates, size_t n_gates, int fft_bl) = 0;
    // layer 0 grows while the circuit is generated: entries [offset, offset + n) were (re)written by the host
    virtual bool input(size_t offset, const F *values, size_t n) = 0;
    // value of every gate of a generic layer; operands in layer 0 come from the spans passed to input()
    virtual bool gates(F *out, size_t n_out, const uniGate *uni, size_t n_uni, const binGate *bin, size_t n_bin, const F *prev,
                       size_t n_prev, const F *two_mul, size_t n_two_mul, const F &scale) = 0;
};

// ---- the witness as a PROGRAM ("a new picture without the host round trip of val") ----
// The circuit's wiring never depends on the picture, only its quantisation scales do. A normal build therefore records, next to the
// circuit, how layer 0's auxiliary witnesses come out of the layer values: which bit / sign / running maximum / window sum of which
// value goes where, and after which layers the range of the activations fixes the next scale. A backend that keeps the circuit and
// the values resident (the HIP prover: zk_witness_program_upload / zk_witness_rerun) replays that program for the next picture; the
// host then only quantises the picture and checks that every recorded scale still comes out the same (else: rebuild).
struct witnessOp {
    enum kind : u8 { BIT = 0, SIGN = 1, MAX = 2, SUM_BIT = 3 };
    u32 src;         // BIT / SIGN / MAX: index in layer src_layer; SUM_BIT: number of the window whose entries are summed
    u32 dst;         // index in layer 0
    u8 src_layer, op, shift, pad_;
};
struct witnessStep {
    enum kind : i32 { AUX = 0, EVAL = 1, RANGE = 2 };
    i32 what;
    i32 layer;           // EVAL: the layer evaluated; RANGE: the layer scanned; AUX: the layer its operations read
    u64 op_begin, op_end;    // AUX: span of `ops`
    u64 win_begin;       // AUX with SUM_BIT: first entry of `windows` of window 0 of this step
    i32 win;             // entries per window
    i32 bits;            // RANGE: x_bit + w_bit of the accumulator; the scale it must reproduce is scales()[scale_index]
    i32 scale_index, pad_;
};
// what a direct-convolution layer (layerType::NCONV) was generated from: lets a prover that knows the pattern factor the layer's gate
// sums (include/zkcnn_hip.h: zk_conv_hint, same layout; the prover checks the hint against the gate list before it trusts it)
struct convHint {
    i32 layer;
    u32 pic_parallel, channel_out, channel_in, nx_in, ny_in, nx_out, ny_out, m, padding, log_stride, weight_start;
};

struct witnessProgram {
    vector<witnessOp> ops;
    vector<u32> windows;
    vector<witnessStep> steps;
    u64 picture_values = 0;      // entries [0, picture_values) of layer 0 are the picture (pic_parallel copies)
    void clear() { ops.clear(); windows.clear(); steps.clear(); picture_values = 0; }
};

class neuralNetwork {
public:
    neuralNetwork(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &i_filename,
                  const string &c_filename, const string &o_filename);
    virtual ~neuralNetwork() {}

    // replace the input file by a seeded synthetic stream (picture ~ U[0,1), weights/biases ~ U(-k,k),
    // k = 1/sqrt(fan_in)); must be called before create()
    // picture_seed != 0: the picture comes from its own stream (the weight stream still skips as many draws), so that sessions with the
    // same `seed` share weights and differ in the picture
    void useSyntheticData(u64 seed, u64 picture_seed = 0);
    // every value create() draws from the current source is also written to `filename` in the reference's data-file format (one decimal per line,
    // picture first, then weights and biases in layer order): the same statement can then be given to the reference's generator, or read back here
    void recordDataTo(const string &filename);
    // the pixel values synthetic stream `picture_seed` yields for this model's picture (what a caller would read from a file)
    vector<double> syntheticPicture(u64 picture_seed) const;

    // ---- next picture on a resident circuit ----
    const witnessProgram &program() const { return prog; }
    const vector<convHint> &convHints() const { return conv_hints; }
    // quantises `pixels` (channel, x, y order; one picture, replicated pic_parallel times like the reference does) with the scale the
    // circuit was built for; false if this picture's range needs a smaller scale than the circuit's (its values would not fit)
    bool quantisePicture(const vector<double> &pixels, vector<F> &out) const;
    // ranges[k] = (largest non-negative value, largest magnitude of a negative value) of the layer RANGE step k scanned, as the
    // resident backend found them: true iff every range FITS the circuit's scale (the scale this picture would get by itself is not smaller;
    // equal for the picture the circuit was built from, larger for one with a smaller range: it only loses precision)
    bool rangesFitScales(const vector<std::pair<u64, u64>> &ranges) const;
    // the classes the output layer of a replayed witness infers (const: clones of a session share this object, each keeps its own result)
    vector<int> inferenceFrom(const vector<F> &last_layer) const;
    void setWitnessAccel(witnessAccel *a) { accel = a; }

    // The circuit's shape depends on the data only through the quantisation scales (bits kept per layer): a build records them,
    // and a build in structure-only mode replays them to produce the SAME circuit without picture, weights or witness --
    // what a verifier that has nothing but the proof needs (host/replay.hpp, zkcnn_verifier_create).
    const vector<int> &scales() const { return scale_log; }
    bool scalesConsumed() const { return scale_pos == scale_log.size(); }
    void setStructureOnly(const vector<int> &recorded) { structure_only = true; scale_log = recorded; scale_pos = 0; }
    // Calibrated build: circuit AND witness for this picture under the given scales (those of scales() of an earlier build of the same model).
    // Same circuit for every picture; scaleOverflow() after create() says that this picture's values do not fit them (the witness is then invalid).
    void setFixedScales(const vector<int> &given) { use_fixed = true; fixed_scales = given; }
    bool scaleOverflow() const { return scale_overflow; }
    bool fixedScalesConsumed() const { return !use_fixed || fixed_pos == fixed_scales.size(); }

    // Fills pr.C (circuit) and pr.val (value of every gate). Works for any prover type exposing those two.
    template <class P>
    void create(P &pr, bool only_compute) { build(pr.C, pr.val, only_compute); }

    void build(layeredCircuit &C, vector<vector<F>> &val, bool only_compute);

    i64 layerCount() const { return SIZE; }
    i64 inputSize() const { return total_in_size; }
    const vector<int> &inferred() const { return infer_result; }

protected:
    vector<vector<convKernel>> conv_section;
    vector<poolKernel> pool;
    vector<fconKernel> full_conn;

    i64 pic_size_x, pic_size_y, pic_channel, pic_parallel;
    i64 SIZE;
    const i64 NCONV_FAST_SIZE, NCONV_SIZE, FFT_SIZE, AVE_POOL_SIZE, FC_SIZE, RELU_SIZE;
    const i64 Q = 9;               // quantisation width
    const i64 Q_BIT_SIZE = 220;    // two_mul holds 2^0..2^220 and their negatives
    i64 T, Q_MAX;

private:
    std::unique_ptr<dataSource> src;
    witnessAccel *accel = nullptr;
    string out_filename;
    vector<int> infer_result;

    poolType pool_ty;
    i64 pool_bl, pool_sz, pool_stride_bl, pool_stride, pool_layer_cnt, conv_layer_cnt;
    i64 nx_in, nx_out, ny_in, ny_out, m, channel_in, channel_out, log_stride, padding;
    i64 new_nx_in, new_ny_in, nx_padded_in, ny_padded_in;
    i64 total_in_size, total_para_size, total_relu_in_size, total_ave_in_size, total_max_in_size;
    int x_bit, w_bit, x_next_bit;

    bool structure_only = false;
    vector<int> scale_log;
    size_t scale_pos = 0;
    // calibrated build: the scales are GIVEN (a model's scales fixed once, from a calibration picture) instead of taken from this picture's
    // ranges; the build notes whether a range needed a smaller scale than the given one (the picture does not fit: scaleOverflow())
    vector<int> fixed_scales;
    size_t fixed_pos = 0;
    bool use_fixed = false, scale_overflow = false;
    int logged(int computed) {       // record in a normal build, replay in a structure-only build
        if (!structure_only && use_fixed) {
            if (fixed_pos >= fixed_scales.size()) throw std::runtime_error("calibrated build: too few quantisation scales");
            const int v = fixed_scales[fixed_pos++];
            if (v < 0 || v > 62) throw std::runtime_error("calibrated build: quantisation scale out of range");
            if (computed < v) scale_overflow = true;
            scale_log.push_back(v);
            return v;
        }
        if (!structure_only) { scale_log.push_back(computed); return computed; }
        if (scale_pos >= scale_log.size()) throw std::runtime_error("statement: too few quantisation scales");
        const int v = scale_log[scale_pos++];
        if (v < 0 || v > 62) throw std::runtime_error("statement: quantisation scale out of range");      // untrusted input (proof file)
        return v;
    }
    witnessProgram prog;
    vector<convHint> conv_hints;
    void logOp(witnessOp::kind k, i64 src_layer, i64 src, i64 dst, i64 shift);
    void logEval(i64 layer_id);
    vector<vector<F>> *vals;       // == &pr.val while building
    i64 in_dirty_lo = 0;           // layer-0 entries from here on are newer than the accelerator's copy
    size_t n_two_mul = 0;
    void touch0(i64 idx) { if (idx < in_dirty_lo) in_dirty_lo = idx; }
    const F *two_mul;

    void setTruncation();
    void planLayout();
    void setConv(i64 nx, i64 ny, const convKernel &conv);
    void setFC(const fconKernel &fc);
    void setPool(const poolKernel &p);
    i64 fftBits() const;
    i64 poolAuxSize() const;
    int nextScaleBits(i64 layer_id);
    int scaleFromRange(i64 range, int bits) const;

    // witness helpers
    void loadPicture(layer &L);
    void loadConvWeight(i64 first_id);
    void loadFcWeight(i64 first_id);
    void loadBias(i64 first_id);
    int quantBits(double mx, double mn) const;
    void putBit(i64 layer_id, i64 idx, i64 dst, i64 shift);
    void putFieldBit(const F &data, i64 dst, i64 shift, i64 src_layer, const vector<u32> &window, bool first_of_window);
    void putSign(i64 layer_id, i64 idx, i64 dst);
    void putMax(i64 layer_id, i64 idx, i64 dst);
    void evalGates(const layer &L, i64 layer_id);
    void evalDotProd(const layer &L, i64 layer_id);
    void evalTransform(const layer &L, i64 layer_id);

    // layer emitters
    void emitInput(layer &L);
    void emitPadding(layer &L, i64 &layer_id, i64 weight_base);
    void emitFFT(layer &L, i64 &layer_id);
    void emitDotProd(layer &L, i64 &layer_id);
    void emitIFFT(layer &L, i64 &layer_id);
    void emitAddBias(layer &L, i64 &layer_id, i64 bias_base);
    void emitConvFast(layer &L, i64 &layer_id, i64 weight_base, i64 bias_base);
    void emitConvMul(layer &L, i64 &layer_id, i64 weight_base);
    void emitConvAdd(layer &L, i64 &layer_id, i64 bias_base);
    void emitRelu(layer &L, i64 &layer_id, i64 block_len);
    void emitAvgPool(layer &L, i64 &layer_id);
    void emitMaxPool(layeredCircuit &C, i64 &layer_id);
    void emitFC(layer &L, i64 &layer_id, i64 first_fc_id, i64 bias_base);
    void reportInference(const layeredCircuit &C);
};
