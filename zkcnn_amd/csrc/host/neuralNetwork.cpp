// Circuit + witness generation. Gate emission ORDER inside every layer follows the reference
// (reference src/neuralNetwork.cpp:144-650) because layeredCircuit::initSubset numbers layer-0
// wires in first-use order and that numbering is visible in the sumcheck transcript.
#include "neuralNetwork.hpp"

namespace {

class fileSource : public dataSource {
public:
    explicit fileSource(const string &name) : in(name) {
        if (!in.is_open()) fprintf(stderr, "Can't find the input file!!!\n");
    }
    double next(kind, i64) override {
        double x = 0;
        in >> x;
        return x;
    }
private:
    std::ifstream in;
};

class synthSource : public dataSource {
public:
    synthSource(u64 seed, u64 picture_seed) : own_picture(picture_seed != 0) { g.seed(seed); if (own_picture) gp.seed(picture_seed); }
    double next(kind k, i64 fan_in) override {
        double u = g.nextUnit();
        if (k == PICTURE) return own_picture ? gp.nextUnit() : u;
        double bound = 1.0 / std::sqrt((double) std::max<i64>(fan_in, 1));
        return (2.0 * u - 1.0) * bound;
    }
private:
    zkff::Xoshiro g, gp;
    bool own_picture;
};

// writes every value the generator draws, in draw order, as a decimal that parses back to the same double: a data file in the reference's format
// (whitespace-separated numbers: src/neuralNetwork.cpp:805-893) of the stream behind it
class teeSource : public dataSource {
public:
    // (the file is opened by the caller BEFORE the source changes hands: a constructor that throws after taking `s` would destroy the data source)
    teeSource(std::unique_ptr<dataSource> s, FILE *file, const string &name) : inner(std::move(s)), f(file), path(name) {}
    ~teeSource() override {
        // a short write (full disk) would leave a data file that reads as a DIFFERENT statement: say so, loudly, where it can still be said
        if (f && (fclose(f) != 0 || failed)) fprintf(stderr, "recordDataTo: writing %s failed -- the file is incomplete\n", path.c_str());
    }
    double next(kind k, i64 fan_in) override {
        const double x = inner->next(k, fan_in);
        if (fprintf(f, "%.17g\n", x) < 0 || ferror(f)) {
            failed = true;
            throw std::runtime_error("recordDataTo: write to " + path + " failed");
        }
        return x;
    }
private:
    std::unique_ptr<dataSource> inner;
    FILE *f;
    string path;
    bool failed = false;
};

} // namespace

void neuralNetwork::recordDataTo(const string &filename) {
    if (!src) throw std::runtime_error("recordDataTo: no data source yet");
    FILE *f = fopen(filename.c_str(), "w");
    if (!f) throw std::runtime_error("recordDataTo: cannot open " + filename);       // (`src` is untouched)
    src.reset(new teeSource(std::move(src), f, filename));
}

neuralNetwork::neuralNetwork(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &i_filename,
                             const string &c_filename, const string &o_filename)
        : pic_size_x(psize_x), pic_size_y(psize_y), pic_channel(pchannel), pic_parallel(pparallel), SIZE(0),
          NCONV_FAST_SIZE(1), NCONV_SIZE(2), FFT_SIZE(5), AVE_POOL_SIZE(1), FC_SIZE(1), RELU_SIZE(1), T(0), Q_MAX(0),
          out_filename(o_filename), pool_ty(NONE), vals(nullptr), two_mul(nullptr) {
    (void) c_filename;      // opened but never read by the reference either (src/neuralNetwork.cpp:32-34)
    if (!i_filename.empty()) src.reset(new fileSource(i_filename));
}

void neuralNetwork::useSyntheticData(u64 seed, u64 picture_seed) { src.reset(new synthSource(seed, picture_seed)); }

vector<double> neuralNetwork::syntheticPicture(u64 picture_seed) const {
    synthSource s(0, picture_seed);
    vector<double> px((size_t) (pic_channel * pic_size_x * pic_size_y));
    for (double &x : px) x = s.next(dataSource::PICTURE, 1);
    return px;
}

// ---------------------------------------------------------------------------------------------
// sizing
// ---------------------------------------------------------------------------------------------
void neuralNetwork::setConv(i64 nx, i64 ny, const convKernel &conv) {
    nx_in = nx; ny_in = ny;
    padding = conv.padding;
    nx_padded_in = nx_in + 2 * padding;
    ny_padded_in = ny_in + 2 * padding;
    m = conv.size;
    channel_in = conv.channel_in;
    channel_out = conv.channel_out;
    log_stride = conv.stride_bl;
    nx_out = ((nx_padded_in - m) >> log_stride) + 1;
    ny_out = ((ny_padded_in - m) >> log_stride) + 1;
    new_nx_in = nx_out;
    new_ny_in = ny_out;
    conv_layer_cnt = conv.ty == FFT ? FFT_SIZE : conv.ty == NAIVE ? NCONV_SIZE : NCONV_FAST_SIZE;
}

void neuralNetwork::setFC(const fconKernel &fc) {
    nx_in = nx_out = ny_in = ny_out = m = 1;
    channel_in = fc.channel_in;
    channel_out = fc.channel_out;
}

void neuralNetwork::setPool(const poolKernel &p) {
    pool_sz = p.size;
    pool_bl = ceilPow2BitLength((u32) pool_sz);
    pool_stride_bl = p.stride_bl;
    pool_stride = 1LL << p.stride_bl;
    pool_layer_cnt = p.ty == MAX ? 1 + ceilPow2BitLength((u32) (sqr(p.size) + 1)) : AVE_POOL_SIZE;
    new_nx_in = ((nx_out - pool_sz) >> pool_stride_bl) + 1;
    new_ny_in = ((ny_out - pool_sz) >> pool_stride_bl) + 1;
}

i64 neuralNetwork::fftBits() const { return ceilPow2BitLength((u32) (nx_padded_in * ny_padded_in)) + 1; }

i64 neuralNetwork::poolAuxSize() const {
    if (pool_ty == AVG) return new_nx_in * new_ny_in * (pool_bl << 1) * channel_out * pic_parallel;
    assert(pool_ty == MAX);
    return new_nx_in * new_ny_in * sqr(pool_sz) * channel_out * pic_parallel * (Q_MAX - 1);
}

// Layer-0 layout: [picture x pic_parallel][conv w, conv b]...[fc w, fc b]... then the auxiliary
// witnesses (bit decompositions, maxima) are appended as the layers that need them are emitted.
void neuralNetwork::planLayout() {
    i64 conv_layers = 0, pool_layers = 0;
    total_para_size = total_relu_in_size = total_ave_in_size = total_max_in_size = 0;
    i64 pos = pic_size_x * pic_size_y * pic_channel * pic_parallel;

    new_nx_in = pic_size_x;
    new_ny_in = pic_size_y;
    for (size_t i = 0; i < conv_section.size(); ++i) {
        for (convKernel &ck : conv_section[i]) {
            setConv(new_nx_in, new_ny_in, ck);
            ck.weight_start_id = pos;
            pos += sqr(m) * channel_in * channel_out;
            ck.bias_start_id = pos;
            pos += channel_out;
        }
        conv_layers += (i64) conv_section[i].size() * (conv_layer_cnt + RELU_SIZE);
        if (i >= pool.size()) continue;
        setPool(pool[i]);
        pool_layers += pool_layer_cnt;
        if (pool[i].ty == MAX) conv_layers -= RELU_SIZE;     // max pooling absorbs the last ReLU
    }
    for (fconKernel &fc : full_conn) {
        setFC(fc);
        fc.weight_start_id = pos;
        pos += channel_out * channel_in;
        fc.bias_start_id = pos;
        pos += channel_out;
    }
    total_para_size = pos - pic_size_x * pic_size_y * pic_channel * pic_parallel;
    total_in_size = pos;
    SIZE = 1 + conv_layers + pool_layers + (FC_SIZE + RELU_SIZE) * (i64) full_conn.size();
    if (!full_conn.empty()) SIZE -= RELU_SIZE;
}

// ---------------------------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------------------------
void neuralNetwork::build(layeredCircuit &C, vector<vector<F>> &val, bool only_compute) {
    assert((src || structure_only) && "no data source: pass an input file or call useSyntheticData()");
    if (!structure_only) scale_log.clear();
    scale_pos = 0;
    fixed_pos = 0;
    scale_overflow = false;
    assert(pool.size() + 1 >= conv_section.size());
    planLayout();
    assert(SIZE > 0 && SIZE < 256);
    C.init((u8) Q_BIT_SIZE, (u8) SIZE);
    val.assign(SIZE, vector<F>());
    vals = &val;
    two_mul = C.two_mul.data();
    n_two_mul = C.two_mul.size();
    in_dirty_lo = 0;
    prog.clear();
    conv_hints.clear();
    prog.picture_values = (u64) (pic_size_x * pic_size_y * pic_channel * pic_parallel);

    i64 layer_id = 0;
    emitInput(C.circuit[layer_id++]);

    new_nx_in = pic_size_x;
    new_ny_in = pic_size_y;
    for (size_t i = 0; i < conv_section.size(); ++i) {
        auto &sec = conv_section[i];
        for (size_t j = 0; j < sec.size(); ++j) {
            const convKernel &conv = sec[j];
            setConv(new_nx_in, new_ny_in, conv);
            pool_ty = (i < pool.size() && j + 1 == sec.size()) ? pool[i].ty : NONE;
            x_bit = x_next_bit;
            if (conv.ty == FFT) {
                emitPadding(C.circuit[layer_id], layer_id, conv.weight_start_id);
                emitFFT(C.circuit[layer_id], layer_id);
                emitDotProd(C.circuit[layer_id], layer_id);
                emitIFFT(C.circuit[layer_id], layer_id);
                emitAddBias(C.circuit[layer_id], layer_id, conv.bias_start_id);
            } else if (conv.ty == NAIVE_FAST) {
                emitConvFast(C.circuit[layer_id], layer_id, conv.weight_start_id, conv.bias_start_id);
            } else {
                emitConvMul(C.circuit[layer_id], layer_id, conv.weight_start_id);
                emitConvAdd(C.circuit[layer_id], layer_id, conv.bias_start_id);
            }
            // re-quantise: keep Q bits of the accumulator, T = bits dropped
            x_next_bit = nextScaleBits(layer_id - 1);
            setTruncation();
            if (pool_ty != MAX)
                emitRelu(C.circuit[layer_id], layer_id, nx_out * ny_out * channel_out * pic_parallel);
        }
        if (i >= pool.size()) continue;
        setPool(pool[i]);
        if (pool[i].ty == AVG) emitAvgPool(C.circuit[layer_id], layer_id);
        else if (pool[i].ty == MAX) emitMaxPool(C, layer_id);
    }

    pool_ty = NONE;
    for (size_t i = 0; i < full_conn.size(); ++i) {
        const fconKernel &fc = full_conn[i];
        setFC(fc);
        x_bit = x_next_bit;
        emitFC(C.circuit[layer_id], layer_id, fc.weight_start_id, fc.bias_start_id);
        if (i + 1 == full_conn.size()) break;
        x_next_bit = nextScaleBits(layer_id - 1);
        setTruncation();
        emitRelu(C.circuit[layer_id], layer_id, channel_out * pic_parallel);
    }
    assert(SIZE == layer_id);

    total_in_size += total_max_in_size + total_ave_in_size + total_relu_in_size;
    initLayer(C.circuit[0], total_in_size, layerType::INPUT);
    assert((size_t) total_in_size == val[0].size());
    reportInference(C);
    if (!only_compute) C.initSubset();
    vals = nullptr;
}

// ---------------------------------------------------------------------------------------------
// witness helpers
// ---------------------------------------------------------------------------------------------
// largest shift s with (mx - mn) * 2^s <= 2^(Q-1) - 1 (reference src/neuralNetwork.cpp:822-825)
int neuralNetwork::quantBits(double mx, double mn) const {
    const int lim = (1 << (Q - 1)) - 1;
    int s = (int) (std::log(lim / (mx - mn)) / std::log(2));
    if ((int) ((mx - mn) * std::exp2(s)) > lim) --s;
    return s;
}

void neuralNetwork::loadPicture(layer &L) {
    auto &v0 = (*vals)[0];
    v0.assign(L.size, F_ZERO);
    touch0(0);
    if (structure_only) { x_next_bit = logged(0); return; }
    const i64 n = pic_channel * pic_size_x * pic_size_y;
    vector<double> dat(n);
    double mx = -10000, mn = 10000;
    for (i64 i = 0; i < n; ++i) {
        dat[i] = src->next(dataSource::PICTURE, 1);
        mx = std::max(mx, dat[i]);
        mn = std::min(mn, dat[i]);
    }
    x_next_bit = logged(quantBits(mx, mn));
    i64 pos = 0;
    for (i64 p = 0; p < pic_parallel; ++p)          // the single picture is replicated pic_parallel times
        for (i64 i = 0; i < n; ++i) v0[pos++] = F((i64) (dat[i] * std::exp2(x_next_bit)));
}

void neuralNetwork::loadConvWeight(i64 first_id) {
    if (structure_only) { w_bit = logged(0); return; }
    const i64 n = channel_out * channel_in * m * m;
    vector<double> dat(n);
    double mx = -10000, mn = 10000;
    for (i64 i = 0; i < n; ++i) {
        dat[i] = src->next(dataSource::WEIGHT, channel_in * m * m);
        mx = std::max(mx, dat[i]);
        mn = std::min(mn, dat[i]);
    }
    w_bit = logged(quantBits(mx, mn));
    auto &v0 = (*vals)[0];
    touch0(first_id);
    for (i64 i = 0; i < n; ++i) v0[first_id + i] = F((i64) (dat[i] * std::exp2(w_bit)));
}

void neuralNetwork::loadFcWeight(i64 first_id) {
    if (structure_only) { w_bit = logged(0); return; }
    const i64 n = channel_out * channel_in;
    vector<double> dat(n);
    double mx = -10000, mn = 10000;
    for (i64 i = 0; i < n; ++i) {
        dat[i] = src->next(dataSource::WEIGHT, channel_in);
        mx = std::max(mx, dat[i]);
        mn = std::min(mn, dat[i]);
    }
    w_bit = logged(quantBits(mx, mn));
    auto &v0 = (*vals)[0];
    touch0(first_id);
    for (i64 i = 0; i < n; ++i) v0[first_id + i] = F((i64) (dat[i] * std::exp2(w_bit)));
}

void neuralNetwork::loadBias(i64 first_id) {
    if (structure_only) return;
    auto &v0 = (*vals)[0];
    touch0(first_id);
    for (i64 co = 0; co < channel_out; ++co) {
        double b = src->next(dataSource::BIAS, channel_in * m * m);
        v0[first_id + co] = F((i64) (b * std::exp2(w_bit + x_bit)));
    }
}

// ---- recording of the witness program (neuralNetwork.hpp: witnessProgram) ----
void neuralNetwork::logOp(witnessOp::kind k, i64 src_layer, i64 src, i64 dst, i64 shift) {
    // consecutive operations form one step as long as they may run side by side: same source layer, and a running maximum (which
    // later operations of the same step would read) never shares a step with anything else
    const bool fresh = prog.steps.empty() || prog.steps.back().what != witnessStep::AUX || prog.steps.back().layer != (i32) src_layer ||
                       ((prog.ops[prog.steps.back().op_begin].op == witnessOp::MAX) != (k == witnessOp::MAX)) ||
                       ((prog.ops[prog.steps.back().op_begin].op == witnessOp::SUM_BIT) != (k == witnessOp::SUM_BIT));
    if (fresh) {
        witnessStep st;
        std::memset(&st, 0, sizeof(st));
        st.what = witnessStep::AUX;
        st.layer = (i32) src_layer;
        st.op_begin = st.op_end = prog.ops.size();
        st.win_begin = prog.windows.size();
        prog.steps.push_back(st);
    }
    witnessOp op = {(u32) src, (u32) dst, (u8) src_layer, (u8) k, (u8) shift, 0};
    prog.ops.push_back(op);
    prog.steps.back().op_end = prog.ops.size();
}
void neuralNetwork::logEval(i64 layer_id) {
    witnessStep st;
    std::memset(&st, 0, sizeof(st));
    st.what = witnessStep::EVAL;
    st.layer = (i32) layer_id;
    prog.steps.push_back(st);
}

void neuralNetwork::putBit(i64 layer_id, i64 idx, i64 dst, i64 shift) {
    if (structure_only) return;
    logOp(witnessOp::BIT, layer_id, idx, dst, shift);
    i64 mag = std::llabs((*vals)[layer_id].at(idx).getInt64());
    touch0(dst);
    (*vals)[0].at(dst) = F((i64) ((mag >> shift) & 1));
}
void neuralNetwork::putFieldBit(const F &data, i64 dst, i64 shift, i64 src_layer, const vector<u32> &window, bool first_of_window) {
    if (structure_only) return;
    {
        const size_t steps_before = prog.steps.size();
        logOp(witnessOp::SUM_BIT, src_layer, 0, dst, shift);
        witnessStep &st = prog.steps.back();
        if (prog.steps.size() != steps_before) st.win = (i32) window.size();
        if (first_of_window) prog.windows.insert(prog.windows.end(), window.begin(), window.end());
        prog.ops.back().src = (u32) ((prog.windows.size() - st.win_begin) / window.size() - 1);      // the window inserted last
    }
    i64 mag = std::llabs(data.getInt64());
    touch0(dst);
    (*vals)[0].at(dst) = F((i64) ((mag >> shift) & 1));
}
void neuralNetwork::putSign(i64 layer_id, i64 idx, i64 dst) {
    if (structure_only) return;
    logOp(witnessOp::SIGN, layer_id, idx, dst, 0);
    touch0(dst);
    (*vals)[0].at(dst) = (*vals)[layer_id].at(idx).isNegative() ? F_ONE : F_ZERO;
}
void neuralNetwork::putMax(i64 layer_id, i64 idx, i64 dst) {
    if (structure_only) return;
    logOp(witnessOp::MAX, layer_id, idx, dst, 0);
    const F &x = (*vals)[layer_id].at(idx);
    F clamped = x.isNegative() ? F_ZERO : x;
    touch0(dst);
    if (clamped > (*vals)[0].at(dst)) (*vals)[0].at(dst) = clamped;
}

// value of every gate of a generic layer (reference src/neuralNetwork.cpp:918-935)
void neuralNetwork::evalGates(const layer &L, i64 layer_id) {
    if (structure_only) return;
    logEval(layer_id);
    auto &val = *vals;
    auto &out = val[layer_id];
    out.assign(L.size, F_ZERO);
    if (accel && layer_id >= 1) {
        auto &v0 = val[0];
        const auto &prev = val[layer_id - 1];
        const i64 lo = std::min<i64>(in_dirty_lo, (i64) v0.size());
        if (accel->input((size_t) lo, v0.data() + lo, v0.size() - (size_t) lo) &&
            accel->gates(out.data(), out.size(), L.uni_gates.data(), L.uni_gates.size(), L.bin_gates.data(), L.bin_gates.size(),
                         layer_id > 1 ? prev.data() : nullptr, layer_id > 1 ? prev.size() : 0, two_mul, n_two_mul, L.scale)) {
            in_dirty_lo = (i64) v0.size();
            return;
        }
    }
    for (const uniGate &gt : L.uni_gates) {
        const F &x = val[gt.lu].at(gt.u);
        if (gt.sc == 0) out[gt.g] = out[gt.g] + x;
        else out[gt.g] = out[gt.g] + x * two_mul[gt.sc];
    }
    for (const binGate &gt : L.bin_gates) {
        F pr = val[gt.getLayerIdU((u8) layer_id)].at(gt.u) * val[gt.getLayerIdV((u8) layer_id)].at(gt.v);
        if (gt.sc == 0) out[gt.g] = out[gt.g] + pr;
        else out[gt.g] = out[gt.g] + pr * two_mul[gt.sc];
    }
    if (!(L.scale == F_ONE)) for (F &x : out) x = x * L.scale;
}

// per-frequency channel contraction (reference src/neuralNetwork.cpp:937-948)
void neuralNetwork::evalDotProd(const layer &L, i64 layer_id) {
    if (structure_only) return;
    logEval(layer_id);
    auto &val = *vals;
    auto &out = val[layer_id];
    out.assign(L.size, F_ZERO);
    const int fb = L.fft_bit_length;
    const u32 len = 1u << fb;
    const auto &prev = val[layer_id - 1];
    if (accel && accel->dotProd(out.data(), out.size() >> fb, prev.data(), prev.size() >> fb, L.bin_gates.data(), L.bin_gates.size(), fb))
        return;
    for (const binGate &gt : L.bin_gates) {
        F *o = &out[(size_t) gt.g << fb];
        const F *a = &prev[(size_t) gt.u << fb], *b = &prev[(size_t) gt.v << fb];
        for (u32 s = 0; s < len; ++s) o[s] = o[s] + a[s] * b[s];
    }
}

// FFT layer: each half-length block is zero-padded and transformed; IFFT layer: inverse transform
// and keep the first half (reference src/neuralNetwork.cpp:950-965)
void neuralNetwork::evalTransform(const layer &L, i64 layer_id) {
    if (structure_only) return;
    logEval(layer_id);
    auto &val = *vals;
    const size_t len = (size_t) 1 << L.fft_bit_length, lenh = len >> 1;
    auto &out = val[layer_id];
    const auto &prev = val[layer_id - 1];
    out.assign(L.size, F_ZERO);
    if (accel) {
        const bool inv = L.ty == layerType::IFFT;
        const size_t count = inv ? L.size / lenh : L.size / len;
        if (prev.size() >= count * (inv ? len : lenh) && accel->ntt(out.data(), prev.data(), L.fft_bit_length, inv, count)) return;
    }
    vector<F> arr(len);
    if (L.ty == layerType::FFT) {
        for (size_t c = 0, d = 0; d < L.size; c += lenh, d += len) {
            for (size_t j = 0; j < lenh; ++j) arr[j] = prev.at(c + j);
            for (size_t j = lenh; j < len; ++j) arr[j].clear();
            fft(arr, L.fft_bit_length, false);
            for (size_t j = 0; j < len; ++j) out[d + j] = arr[j];
        }
    } else {
        for (size_t c = 0, d = 0; c < L.size; c += lenh, d += len) {
            for (size_t j = 0; j < len; ++j) arr[j] = prev.at(d + j);
            fft(arr, L.fft_bit_length, true);
            for (size_t j = 0; j < lenh; ++j) out[c + j] = arr[j];
        }
    }
}

// T = bits dropped by the re-quantisation, Q_MAX = width of the value that is decomposed into bits. The scales may come from an
// untrusted statement (proof file): they size layers (block_len * Q_MAX rows) and index two_mul, so they are bounded here -- in
// every build, the NDEBUG asserts elsewhere are not a defence.
void neuralNetwork::setTruncation() {
    T = x_bit + w_bit - x_next_bit;
    Q_MAX = Q + T;
    if (x_bit < 0 || x_bit > 62 || w_bit < 0 || w_bit > 62 || x_next_bit < 0 || x_next_bit > 62 || T < 0 || Q_MAX < Q || Q_MAX > 64)
        throw std::runtime_error("statement: quantisation scale out of range");
}

// scale (in bits) of the next activation so that its range fits Q-1 bits
// (reference src/neuralNetwork.cpp:967-977)
int neuralNetwork::nextScaleBits(i64 layer_id) {
    if (structure_only) return logged(0);
    F mx = F_ZERO, mn = F_ZERO;
    for (const F &x : (*vals)[layer_id]) {
        if (!x.isNegative()) { if (x > mx) mx = x; }
        else { F nx = -x; if (nx > mn) mn = nx; }
    }
    i64 range = (mx + mn).getInt64();
    witnessStep st;
    std::memset(&st, 0, sizeof(st));
    st.what = witnessStep::RANGE;
    st.layer = (i32) layer_id;
    st.bits = x_bit + w_bit;
    st.scale_index = (i32) scale_log.size();
    prog.steps.push_back(st);
    return logged(scaleFromRange(range, x_bit + w_bit));
}
int neuralNetwork::scaleFromRange(i64 range, int bits) const {
    if (range <= 0) range = 1;          // a layer of zeros: the largest finite scale (the reference divides by zero here and casts the infinity)
    double real_scale = range / std::exp2(bits);
    return (int) std::log2(((1 << (Q - 1)) - 1) / real_scale);
}

bool neuralNetwork::quantisePicture(const vector<double> &pixels, vector<F> &out) const {
    const i64 n = pic_channel * pic_size_x * pic_size_y;
    if ((i64) pixels.size() != n || scale_log.empty()) return false;
    double mx = -10000, mn = 10000;
    for (double x : pixels) { mx = std::max(mx, x); mn = std::min(mn, x); }
    if (!(mx > mn) || quantBits(mx, mn) < scale_log[0]) return false;
    out.resize((size_t) (n * pic_parallel));
    size_t pos = 0;
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 i = 0; i < n; ++i) out[pos++] = F((i64) (pixels[i] * std::exp2(scale_log[0])));
    return true;
}

bool neuralNetwork::rangesFitScales(const vector<std::pair<u64, u64>> &ranges) const {
    size_t k = 0;
    for (const witnessStep &st : prog.steps) {
        if (st.what != witnessStep::RANGE) continue;
        if (k >= ranges.size() || st.scale_index < 0 || (size_t) st.scale_index >= scale_log.size()) return false;
        if ((ranges[k].first >> 62) || (ranges[k].second >> 62)) return false;
        const i64 range = (i64) (ranges[k].first + ranges[k].second);
        ++k;
        if (range <= 0 || scaleFromRange(range, st.bits) < scale_log[st.scale_index]) return false;
    }
    return k == ranges.size();
}

vector<int> neuralNetwork::inferenceFrom(const vector<F> &last_layer) const {
    vector<int> out;
    if (full_conn.empty()) return out;
    const int n_class = (int) full_conn.back().channel_out;
    for (int p = 0; p < pic_parallel; ++p) {
        int k = -1;
        F best = F_ZERO;
        for (int c = 0; c < n_class; ++c) {
            const F &t = last_layer.at(matIdx(p, c, n_class));
            if (!t.isNegative() && (k == -1 || best < t)) { k = c; best = t; }
        }
        out.push_back(k);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// layer emitters
// ---------------------------------------------------------------------------------------------
void neuralNetwork::emitInput(layer &L) {
    initLayer(L, total_in_size, layerType::INPUT);
    L.uni_gates.reserve(total_in_size);
    for (i64 i = 0; i < total_in_size; ++i) L.uni_gates.emplace_back((u32) i, 0u, (u8) 0, (u8) 0);
    loadPicture(L);
}

void neuralNetwork::emitPadding(layer &L, i64 &layer_id, i64 weight_base) {
    const i64 lenh = (1LL << fftBits()) >> 1;
    initLayer(L, lenh * channel_in * (pic_parallel + channel_out), layerType::PADDING);
    L.fft_bit_length = (i8) fftBits();

    // data: written reversed so that the linear convolution lands where ADD_BIAS reads it
    const i64 lo = -padding, x_end = nx_in + padding, y_end = ny_in + padding;
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 ci = 0; ci < channel_in; ++ci)
            for (i64 x = lo; x < x_end; ++x)
                for (i64 y = lo; y < y_end; ++y) {
                    if (!check(x, y, nx_in, ny_in)) continue;
                    i64 g = cubIdx(p, ci, matIdx(x_end - x - 1, y_end - y - 1, ny_padded_in), channel_in, lenh);
                    i64 u = tesIdx(p, ci, x, y, channel_in, nx_in, ny_in);
                    L.uni_gates.emplace_back((u32) g, (u32) u, (u8) (layer_id - 1), (u8) 0);
                }
    // kernels, from layer 0
    const i64 first = pic_parallel * channel_in * lenh;
    for (i64 co = 0; co < channel_out; ++co)
        for (i64 ci = 0; ci < channel_in; ++ci)
            for (i64 x = 0; x < nx_padded_in; ++x)
                for (i64 y = 0; y < ny_padded_in; ++y) {
                    if (!check(x, y, m, m)) continue;
                    i64 g = first + cubIdx(co, ci, matIdx(x, y, ny_padded_in), channel_in, lenh);
                    i64 u = weight_base + tesIdx(co, ci, x, y, channel_in, m, m);
                    L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) 0);
                }
    loadConvWeight(weight_base);
    evalGates(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitFFT(layer &L, i64 &layer_id) {
    initLayer(L, (1LL << fftBits()) * channel_in * (pic_parallel + channel_out), layerType::FFT);
    L.fft_bit_length = (i8) fftBits();
    evalTransform(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitDotProd(layer &L, i64 &layer_id) {
    initLayer(L, (1LL << fftBits()) * channel_out * pic_parallel, layerType::DOT_PROD);
    L.need_phase2 = true;
    L.fft_bit_length = (i8) fftBits();
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 ci = 0; ci < channel_in; ++ci)
                L.bin_gates.emplace_back((u32) matIdx(p, co, channel_out), (u32) matIdx(p, ci, channel_in),
                                         (u32) matIdx(pic_parallel + co, ci, channel_in), (u8) 0, (u8) 1);
    evalDotProd(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitIFFT(layer &L, i64 &layer_id) {
    const i64 lenh = (1LL << fftBits()) >> 1;
    initLayer(L, lenh * channel_out * pic_parallel, layerType::IFFT);
    L.fft_bit_length = (i8) fftBits();
    F::inv(L.scale, F((unsigned long long) (1ULL << L.fft_bit_length)));
    evalTransform(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitAddBias(layer &L, i64 &layer_id, i64 bias_base) {
    initLayer(L, nx_out * ny_out * channel_out * pic_parallel, layerType::ADD_BIAS);
    const i64 lenh = (1LL << fftBits()) >> 1;
    const i64 lo = -padding, x_end = nx_in + padding, y_end = ny_in + padding, st = 1LL << log_stride;
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 x = lo; x + m <= x_end; x += st)
                for (i64 y = lo; y + m <= y_end; y += st) {
                    i64 u = cubIdx(p, co, matIdx(x_end - x - 1, y_end - y - 1, ny_padded_in), channel_out, lenh);
                    i64 g = tesIdx(p, co, (x - lo) >> log_stride, (y - lo) >> log_stride, channel_out, nx_out, ny_out);
                    L.uni_gates.emplace_back((u32) g, (u32) (bias_base + co), (u8) 0, (u8) 0);
                    L.uni_gates.emplace_back((u32) g, (u32) u, (u8) (layer_id - 1), (u8) 0);
                }
    loadBias(bias_base);
    evalGates(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitConvFast(layer &L, i64 &layer_id, i64 weight_base, i64 bias_base) {
    initLayer(L, nx_out * ny_out * channel_out * pic_parallel, layerType::NCONV);
    L.need_phase2 = true;
    const i64 lo = -padding, x_end = nx_in + padding, y_end = ny_in + padding, st = 1LL << log_stride;
    const u8 lcode = (u8) (2 * (layer_id > 1));
    L.bin_gates.reserve((size_t) (pic_parallel * channel_out * channel_in * nx_out * ny_out * m * m));
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 ci = 0; ci < channel_in; ++ci)
                for (i64 x = lo; x + m <= x_end; x += st)
                    for (i64 y = lo; y + m <= y_end; y += st) {
                        i64 g = tesIdx(p, co, (x - lo) >> log_stride, (y - lo) >> log_stride, channel_out, nx_out, ny_out);
                        if (ci == 0 && ~bias_base)
                            L.uni_gates.emplace_back((u32) g, (u32) (bias_base + co), (u8) 0, (u8) 0);
                        for (i64 wx = x; wx < x + m; ++wx)
                            for (i64 wy = y; wy < y + m; ++wy) {
                                if (!check(wx, wy, nx_in, ny_in)) continue;
                                i64 u = tesIdx(p, ci, wx, wy, channel_in, nx_in, ny_in);
                                i64 v = weight_base + tesIdx(co, ci, wx - x, wy - y, channel_in, m, m);
                                L.bin_gates.emplace_back((u32) g, (u32) u, (u32) v, (u8) 0, lcode);
                            }
                    }
    {
        convHint h = {(i32) layer_id, (u32) pic_parallel, (u32) channel_out, (u32) channel_in, (u32) nx_in, (u32) ny_in, (u32) nx_out, (u32) ny_out,
                      (u32) m, (u32) padding, (u32) log_stride, (u32) weight_base};
        conv_hints.push_back(h);
    }
    loadConvWeight(weight_base);
    if (~bias_base) loadBias(bias_base);
    evalGates(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitConvMul(layer &L, i64 &layer_id, i64 weight_base) {
    const i64 lo = -padding, x_end = nx_in + padding, y_end = ny_in + padding, st = 1LL << log_stride;
    const u8 lcode = (u8) (2 * (layer_id > 1));
    i64 g = 0;
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 ci = 0; ci < channel_in; ++ci)
                for (i64 x = lo; x + m <= x_end; x += st)
                    for (i64 y = lo; y + m <= y_end; y += st)
                        for (i64 wx = x; wx < x + m; ++wx)
                            for (i64 wy = y; wy < y + m; ++wy) {
                                if (!check(wx, wy, nx_in, ny_in)) continue;
                                i64 u = tesIdx(p, ci, wx, wy, channel_in, nx_in, ny_in);
                                i64 v = weight_base + tesIdx(co, ci, wx - x, wy - y, channel_in, m, m);
                                L.bin_gates.emplace_back((u32) g++, (u32) u, (u32) v, (u8) 0, lcode);
                            }
    initLayer(L, g, layerType::NCONV_MUL);
    L.need_phase2 = true;
    loadConvWeight(weight_base);
    evalGates(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitConvAdd(layer &L, i64 &layer_id, i64 bias_base) {
    initLayer(L, nx_out * ny_out * channel_out * pic_parallel, layerType::NCONV_ADD);
    const i64 lo = -padding, x_end = nx_in + padding, y_end = ny_in + padding, st = 1LL << log_stride;
    i64 u = 0;
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 ci = 0; ci < channel_in; ++ci)
                for (i64 x = lo; x + m <= x_end; x += st)
                    for (i64 y = lo; y + m <= y_end; y += st) {
                        i64 g = tesIdx(p, co, (x - lo) >> log_stride, (y - lo) >> log_stride, channel_out, nx_out, ny_out);
                        if (ci == 0 && ~bias_base)
                            L.uni_gates.emplace_back((u32) g, (u32) (bias_base + co), (u8) 0, (u8) 0);
                        for (i64 wx = x; wx < x + m; ++wx)
                            for (i64 wy = y; wy < y + m; ++wy)
                                if (check(wx, wy, nx_in, ny_in))
                                    L.uni_gates.emplace_back((u32) g, (u32) u++, (u8) (layer_id - 1), (u8) 0);
                    }
    if (~bias_base) loadBias(bias_base);
    evalGates(L, layer_id);
    ++layer_id;
}

// ReLU with truncation. Three row blocks over `block_len` activations x:
//   [0, blk)            y   = (1 - sign) * sum_{s=1}^{Q-1} bit_s 2^{Q-1-s}
//   [blk, 2blk)         0   = -x + 2 x sign + sum_{s=1}^{Q_MAX-1} bit_s 2^{Q_MAX-1-s}
//   [2blk, 2blk+blk*Q_MAX)  0 = b*b - b for every auxiliary bit (sign included)
// Conv and fully-connected variants emit the same gates (reference src/neuralNetwork.cpp:344-439).
void neuralNetwork::emitRelu(layer &L, i64 &layer_id, i64 block_len) {
    initLayer(L, block_len * (2 + Q_MAX), layerType::RELU);
    L.need_phase2 = true;
    L.zero_start_id = (u32) block_len;

    auto &v0 = (*vals)[0];
    const i64 first_dcmp_id = (i64) v0.size();
    touch0(first_dcmp_id);
    v0.resize(v0.size() + block_len * Q_MAX, F_ZERO);
    total_relu_in_size += block_len * Q_MAX;
    const u8 lcode = (u8) (2 * (layer_id > 1));

    for (i64 g = 0; g < block_len; ++g) {
        const i64 sign_u = first_dcmp_id + g * Q_MAX;
        for (i64 s = 1; s < Q; ++s) {
            L.uni_gates.emplace_back((u32) g, (u32) (sign_u + s), (u8) 0, (u8) (Q - 1 - s));
            L.bin_gates.emplace_back((u32) g, (u32) sign_u, (u32) (sign_u + s), (u8) (Q - s + Q_BIT_SIZE), (u8) 0);
        }
    }
    for (i64 u = 0; u < block_len; ++u) {
        const i64 g = block_len + u, sign_v = first_dcmp_id + u * Q_MAX;
        L.uni_gates.emplace_back((u32) g, (u32) u, (u8) (layer_id - 1), (u8) (Q_BIT_SIZE + 1));
        L.bin_gates.emplace_back((u32) g, (u32) u, (u32) sign_v, (u8) 1, lcode);
        putSign(layer_id - 1, u, sign_v);
        for (i64 s = 1; s < Q_MAX; ++s) {
            L.uni_gates.emplace_back((u32) g, (u32) (sign_v + s), (u8) 0, (u8) (Q_MAX - s - 1));
            putBit(layer_id - 1, u, sign_v + s, Q_MAX - s - 1);
        }
    }
    for (i64 k = 0; k < block_len * Q_MAX; ++k) {
        const i64 g = 2 * block_len + k, u = first_dcmp_id + k;
        L.bin_gates.emplace_back((u32) g, (u32) u, (u32) u, (u8) 0, (u8) 0);
        L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) (Q_BIT_SIZE + 1));
    }
    evalGates(L, layer_id);
    ++layer_id;
}

void neuralNetwork::emitAvgPool(layer &L, i64 &layer_id) {
    pool_ty = AVG;
    const i64 zero_start = new_nx_in * new_ny_in * channel_out * pic_parallel;
    const i64 dbl = pool_bl << 1;
    initLayer(L, zero_start + poolAuxSize(), layerType::AVG_POOL);
    F::inv(L.scale, F((long long) sqr(pool_sz)));
    L.zero_start_id = (u32) zero_start;
    L.need_phase2 = true;

    auto &val = *vals;
    const i64 first_gate_id = (i64) val[0].size();
    touch0(first_gate_id);
    val[0].resize(val[0].size() + zero_start * dbl, F_ZERO);
    total_ave_in_size += zero_start * dbl;

    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co)
            for (i64 x = 0; x + pool_sz <= nx_out; x += pool_stride)
                for (i64 y = 0; y + pool_sz <= ny_out; y += pool_stride) {
                    i64 g = tesIdx(p, co, x >> pool_stride_bl, y >> pool_stride_bl, channel_out, new_nx_in, new_ny_in);
                    F sum = F_ZERO;
                    vector<u32> window;
                    for (i64 wx = x; wx < x + pool_sz; ++wx)
                        for (i64 wy = y; wy < y + pool_sz; ++wy) {
                            i64 u = tesIdx(p, co, wx, wy, channel_out, nx_out, ny_out);
                            L.uni_gates.emplace_back((u32) g, (u32) u, (u8) (layer_id - 1), (u8) 0);
                            if (!structure_only) sum = sum + val[layer_id - 1][u];
                            window.push_back((u32) u);
                        }
                    // subtract the remainder bits so that the division by pool_sz^2 is exact
                    for (i64 k = 0; k < dbl; ++k) {
                        i64 idx = matIdx(g, k, dbl), u = first_gate_id + idx, g_bit = zero_start + idx;
                        L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) (dbl - k + Q_BIT_SIZE));
                        putFieldBit(sum, u, dbl - k - 1, layer_id - 1, window, k == 0);
                        L.bin_gates.emplace_back((u32) g_bit, (u32) u, (u32) u, (u8) 0, (u8) 0);
                        L.uni_gates.emplace_back((u32) g_bit, (u32) u, (u8) 0, (u8) (Q_BIT_SIZE + 1));
                    }
                }
    evalGates(L, layer_id);
    ++layer_id;
}

// Max pooling over pool_sz^2 windows, fused with the ReLU of the preceding conv
// (reference src/neuralNetwork.cpp:486-627). Layer A: d_k = max - x_k and the bit decomposition
// of max; then a product tree over the d_k (one of them must vanish), the bit decompositions of
// every d_k (non-negativity), and in the last layer the truncated output tensor.
void neuralNetwork::emitMaxPool(layeredCircuit &C, i64 &layer_id) {
    pool_ty = MAX;
    auto &val = *vals;
    const i64 tot_new = new_nx_in * new_ny_in * channel_out * pic_parallel;
    const i64 ksq = sqr(pool_sz);

    const i64 first_dcmp_id = (i64) val[0].size();
    touch0(first_dcmp_id);
    const i64 dcmp_cnt = poolAuxSize();
    val[0].resize(val[0].size() + dcmp_cnt, F_ZERO);
    const i64 first_max_id = (i64) val[0].size();
    val[0].resize(val[0].size() + tot_new, F_ZERO);
    const i64 first_max_dcmp_id = (i64) val[0].size();
    val[0].resize(val[0].size() + tot_new * (Q_MAX - 1), F_ZERO);
    total_max_in_size += dcmp_cnt + tot_new + tot_new * (Q_MAX - 1);

    {
        layer &L = C.circuit[layer_id];
        initLayer(L, tot_new * ksq + tot_new, layerType::MAX_POOL);
        L.zero_start_id = (u32) (tot_new * ksq);
        for (i64 p = 0; p < pic_parallel; ++p)
            for (i64 co = 0; co < channel_out; ++co)
                for (i64 x = 0; x + pool_sz <= nx_out; x += pool_stride)
                    for (i64 y = 0; y + pool_sz <= ny_out; y += pool_stride) {
                        i64 i_max = tesIdx(p, co, x >> pool_stride_bl, y >> pool_stride_bl, channel_out, new_nx_in, new_ny_in);
                        i64 u_max = first_max_id + i_max;
                        for (i64 wx = x; wx < x + pool_sz; ++wx)
                            for (i64 wy = y; wy < y + pool_sz; ++wy) {
                                i64 g = cubIdx(i_max, wx - x, wy - y, pool_sz, pool_sz);
                                i64 u_g = tesIdx(p, co, wx, wy, channel_out, nx_out, ny_out);
                                L.uni_gates.emplace_back((u32) g, (u32) u_max, (u8) 0, (u8) 0);
                                L.uni_gates.emplace_back((u32) g, (u32) u_g, (u8) (layer_id - 1), (u8) (Q_BIT_SIZE + 1));
                                putMax(layer_id - 1, u_g, u_max);
                            }
                    }
        for (i64 i = 0; i < tot_new; ++i) {
            i64 g = L.zero_start_id + i, u = first_max_id + i;
            L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) (Q_BIT_SIZE + 1));
            for (i64 b = 0; b < Q_MAX - 1; ++b) {
                i64 ub = first_max_dcmp_id + matIdx(i, b, Q_MAX - 1);
                L.uni_gates.emplace_back((u32) g, (u32) ub, (u8) 0, (u8) (Q_MAX - 2 - b));
                putBit(0, u, ub, Q_MAX - 2 - b);
            }
        }
        evalGates(L, layer_id);
        ++layer_id;
    }

    i64 contain_max_ly = 1, ksize = ksq;
    while (!(ksize & 1)) { ksize >>= 1; ++contain_max_ly; }
    ksize = ksq;
    for (i64 i = 1; i < pool_layer_cnt; ++i) {
        layer &L = C.circuit[layer_id];
        const bool first = i == 1, last = i == pool_layer_cnt - 1;
        const i64 halfk = (ksize + 1) >> 1;
        i64 size = tot_new * (halfk + (first ? ksize : 0)) + (last ? tot_new * Q_MAX + tot_new * ksq * (Q_MAX - 1) : 0);
        initLayer(L, size, layerType::MAX_POOL);
        L.need_phase2 = true;
        const u8 lprev = (u8) (layer_id > 1), lmix = (u8) (2 * (layer_id > 1));

        i64 before_mul = 0;
        if (last) {     // the pooled, truncated output tensor
            before_mul = tot_new;
            for (i64 g = 0; g < tot_new; ++g)
                for (i64 j = 0; j < Q - 1; ++j)
                    L.uni_gates.emplace_back((u32) g, (u32) (first_max_dcmp_id + matIdx(g, j, Q_MAX - 1)), (u8) 0,
                                             (u8) (Q - 2 - j));
        }
        for (i64 cnt = 0; cnt < tot_new; ++cnt)
            for (i64 j = 0; (j << 1) < ksize; ++j) {
                i64 g = before_mul + matIdx(cnt, j, halfk);
                i64 u = matIdx(cnt, j << 1, ksize);
                if ((j << 1 | 1) < ksize)
                    L.bin_gates.emplace_back((u32) g, (u32) u, (u32) matIdx(cnt, j << 1 | 1, ksize), (u8) 0, lprev);
                else if (i == contain_max_ly)
                    L.bin_gates.emplace_back((u32) g, (u32) u, (u32) (first_max_id + cnt), (u8) 0, lmix);
                else
                    L.uni_gates.emplace_back((u32) g, (u32) u, (u8) (layer_id - 1), (u8) 0);
            }
        if (first) {
            const i64 minus_cnt = tot_new * ksize, minus_new_cnt = tot_new * halfk;
            L.zero_start_id = (u32) minus_new_cnt;
            for (i64 v = 0; v < minus_cnt; ++v) {
                i64 g = minus_new_cnt + v;
                L.uni_gates.emplace_back((u32) g, (u32) v, (u8) (layer_id - 1), (u8) (Q_BIT_SIZE + 1));
                for (i64 b = 0; b < Q_MAX - 1; ++b) {
                    i64 u = first_dcmp_id + matIdx(v, b, Q_MAX - 1);
                    L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) (Q_MAX - 2 - b));
                    putBit(layer_id - 1, v, u, Q_MAX - 2 - b);
                }
            }
        } else if (last) {
            const i64 minus_cnt = tot_new * ksq;
            L.zero_start_id = (u32) before_mul;
            for (i64 j = 0; j < minus_cnt; ++j) {
                i64 g = before_mul + tot_new + j, u = first_dcmp_id + j;
                L.bin_gates.emplace_back((u32) g, (u32) u, (u32) u, (u8) 0, (u8) 0);
                L.uni_gates.emplace_back((u32) g, (u32) u, (u8) 0, (u8) (Q_BIT_SIZE + 1));
            }
        }
        ksize = halfk;
        evalGates(L, layer_id);
        ++layer_id;
    }
}

void neuralNetwork::emitFC(layer &L, i64 &layer_id, i64 first_fc_id, i64 bias_base) {
    initLayer(L, channel_out * pic_parallel, layerType::FCONN);
    L.need_phase2 = true;
    const u8 lcode = (u8) (2 * (layer_id > 1));
    L.bin_gates.reserve((size_t) (pic_parallel * channel_out * channel_in));
    for (i64 p = 0; p < pic_parallel; ++p)
        for (i64 co = 0; co < channel_out; ++co) {
            i64 g = matIdx(p, co, channel_out);
            L.uni_gates.emplace_back((u32) g, (u32) (bias_base + co), (u8) 0, (u8) 0);
            for (i64 ci = 0; ci < channel_in; ++ci)
                L.bin_gates.emplace_back((u32) g, (u32) matIdx(p, ci, channel_in),
                                         (u32) (first_fc_id + matIdx(co, ci, channel_in)), (u8) 0, lcode);
        }
    loadFcWeight(first_fc_id);
    loadBias(bias_base);
    evalGates(L, layer_id);
    ++layer_id;
}

// arg-max class per picture over the non-negative logits (reference src/neuralNetwork.cpp:994-1016)
void neuralNetwork::reportInference(const layeredCircuit &C) {
    infer_result.clear();
    if (!full_conn.empty() && !structure_only) {
        const int n_class = (int) full_conn.back().channel_out;
        const auto &outv = (*vals)[SIZE - 1];
        for (int p = 0; p < pic_parallel; ++p) {
            int k = -1;
            F best = F_ZERO;
            for (int c = 0; c < n_class; ++c) {
                const F &t = outv.at(matIdx(p, c, n_class));
                if (!t.isNegative() && (k == -1 || best < t)) { k = c; best = t; }
            }
            infer_result.push_back(k);
        }
        if (!out_filename.empty()) {
            std::ofstream out(out_filename);
            for (int k : infer_result) out << k << std::endl;
        }
    }
    const layer &L0 = C.circuit[0];
    output_tb[WS_OUT_ID] = std::to_string(L0.size) + "(2^" + std::to_string((int) ceilPow2BitLength(L0.size)) + ")";
}
