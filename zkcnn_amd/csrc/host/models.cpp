#include "models.hpp"

namespace {
struct stage { vector<i64> conv_out; };   // output channels of the convolutions before one pooling

i64 pooled(i64 n, const poolKernel &p) { return ((n - p.size) >> p.stride_bl) + 1; }
}

vgg::vgg(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &i_filename, const string &c_filename,
         const string &o_filename, const string &network)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, i_filename, c_filename, o_filename) {
    assert(psize_x == psize_y && psize_x == 32);
    string tokens = network;
    {
        std::ifstream f(network);
        if (f.is_open()) { std::stringstream ss; ss << f.rdbuf(); tokens = ss.str(); }
    }
    const convType ty = defaultConvType(3, pparallel);
    conv_section.resize(5);
    std::istringstream in(tokens);
    string tok;
    i64 ch_in = pic_channel, nx = pic_size_x, ny = pic_size_y;
    size_t sec = 0;
    while (in >> tok) {
        if (tok[0] == 'M' || tok[0] == 'A') {
            pool.emplace_back(tok[0] == 'M' ? MAX : AVG, 2, 1);
            nx = pooled(nx, pool.back());
            ny = pooled(ny, pool.back());
            ++sec;
        } else {
            i64 ch_out = std::stoi(tok);
            if (sec >= conv_section.size()) conv_section.resize(sec + 1);
            conv_section[sec].emplace_back(ty, ch_out, ch_in, 3);
            ch_in = ch_out;
        }
    }
    full_conn.emplace_back(512, nx * ny * ch_in);
    full_conn.emplace_back(512, 512);
    full_conn.emplace_back(10, 512);
}

static void vggStages(vector<vector<convKernel>> &conv_section, vector<poolKernel> &pool, vector<fconKernel> &full_conn,
                      const vector<stage> &stages, i64 pic, i64 pic_channel, i64 pparallel, poolType pool_ty) {
    const convType ty = defaultConvType(3, pparallel);
    conv_section.resize(stages.size());
    i64 ch_in = pic_channel, n = pic;
    for (size_t s = 0; s < stages.size(); ++s) {
        for (i64 co : stages[s].conv_out) {
            conv_section[s].emplace_back(ty, co, ch_in, 3);
            ch_in = co;
        }
        pool.emplace_back(pool_ty, 2, 1);
        n = pooled(n, pool.back());
    }
    if (pic == 224) {
        full_conn.emplace_back(4096, n * n * ch_in);
        full_conn.emplace_back(4096, 4096);
        full_conn.emplace_back(1000, 4096);
    } else {
        assert(pic == 32);
        full_conn.emplace_back(512, n * n * ch_in);
        full_conn.emplace_back(512, 512);
        full_conn.emplace_back(10, 512);
    }
}

vgg16::vgg16(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
             const string &c_filename, const string &o_filename)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, i_filename, c_filename, o_filename) {
    assert(psize_x == psize_y);
    vggStages(conv_section, pool, full_conn,
              {{{64, 64}}, {{128, 128}}, {{256, 256, 256}}, {{512, 512, 512}}, {{512, 512, 512}}},
              pic_size_x, pic_channel, pparallel, pool_ty);
}

vgg11::vgg11(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
             const string &c_filename, const string &o_filename)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, i_filename, c_filename, o_filename) {
    assert(psize_x == psize_y);
    vggStages(conv_section, pool, full_conn, {{{64}}, {{128}}, {{256, 256}}, {{512, 512}}, {{512, 512}}},
              pic_size_x, pic_channel, pparallel, pool_ty);
}

lenet::lenet(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
             const string &c_filename, const string &o_filename)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, i_filename, c_filename, o_filename) {
    const convType ty = defaultConvType(5, pparallel);
    const i64 pad0 = (psize_x == 28 && psize_y == 28) ? 2 : 0;      // MNIST 28x28 is padded to 32x32
    conv_section.resize(2);
    conv_section[0].emplace_back(ty, 6, pchannel, 5, 0, pad0);
    pool.emplace_back(pool_ty, 2, 1);
    conv_section[1].emplace_back(ty, 16, 6, 5, 0, 0);
    pool.emplace_back(pool_ty, 2, 1);
    full_conn.emplace_back(120, 400);
    full_conn.emplace_back(84, 120);
    full_conn.emplace_back(10, 84);
}

lenetCifar::lenetCifar(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty,
                       const string &i_filename, const string &c_filename, const string &o_filename)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, i_filename, c_filename, o_filename) {
    const convType ty = defaultConvType(5, pparallel);
    conv_section.resize(3);
    conv_section[0].emplace_back(ty, 6, pchannel, 5, 0, 0);
    pool.emplace_back(pool_ty, 2, 1);
    conv_section[1].emplace_back(ty, 16, 6, 5, 0, 0);
    pool.emplace_back(pool_ty, 2, 1);
    conv_section[2].emplace_back(ty, 120, 16, 5, 0, 0);
    full_conn.emplace_back(84, 120);
    full_conn.emplace_back(10, 84);
}

customNet::customNet(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &spec)
        : neuralNetwork(psize_x, psize_y, pchannel, pparallel, "", "", "") {
    std::istringstream in(spec);
    string tok;
    i64 ch = pchannel, nx = psize_x, ny = psize_y;
    conv_section.emplace_back();
    bool flat = false;
    while (in >> tok) {
        if (tok[0] == 'C') {
            assert(!flat);
            i64 co = 0, k = 3, pad = 1;
            char alg = 's';
            int got = sscanf(tok.c_str(), "C%lld:%lld:%lld:%c", &co, &k, &pad, &alg);
            assert(got >= 3);
            (void) got;
            convType ty = alg == 'f' ? FFT : alg == 'n' ? NAIVE : NAIVE_FAST;
            if (conv_section.size() == pool.size()) conv_section.emplace_back();
            conv_section.back().emplace_back(ty, co, ch, k, 0, pad);
            ch = co;
            nx = nx + 2 * pad - k + 1;
            ny = ny + 2 * pad - k + 1;
        } else if (tok[0] == 'M' || tok[0] == 'A') {
            assert(!flat && conv_section.size() == pool.size() + 1 && !conv_section.back().empty());
            pool.emplace_back(tok[0] == 'M' ? MAX : AVG, 2, 1);
            nx = pooled(nx, pool.back());
            ny = pooled(ny, pool.back());
        } else if (tok[0] == 'F') {
            i64 co = std::stoll(tok.substr(1));
            full_conn.emplace_back(co, flat ? ch : nx * ny * ch);
            ch = co;
            flat = true;
        }
    }
    if (conv_section.back().empty()) conv_section.pop_back();
}

neuralNetwork *makeModel(const string &name, i64 px, i64 py, i64 pc, i64 pp) {
    if (name == "lenet") return new lenet(px, py, pc, pp, MAX, "", "", "");
    if (name == "lenet.avg") return new lenet(px, py, pc, pp, AVG, "", "", "");
    if (name == "lenetCifar") return new lenetCifar(px, py, pc, pp, MAX, "", "", "");
    if (name == "vgg11") return new vgg11(px, py, pc, pp, MAX, "", "", "");
    if (name == "vgg16") return new vgg16(px, py, pc, pp, MAX, "", "", "");
    if (name.compare(0, 4, "vgg:") == 0) return new vgg(px, py, pc, pp, "", "", "", name.substr(4));
    if (name.compare(0, 7, "custom:") == 0) return new customNet(px, py, pc, pp, name.substr(7));
    return nullptr;
}
