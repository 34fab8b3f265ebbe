// Network topologies (constructor = shape tables only). lenet / vgg / vgg11 / vgg16 / lenetCifar
// have the shapes of reference src/models.cpp:12-206; `customNet` is this repo's small-case
// builder for tests (the reference's singleConv / ccnn helpers have no caller and are not rebuilt).
#pragma once
#include "neuralNetwork.hpp"

// conv algorithm rule shared by every reference model: FFT iff kernel > 3 or batch > 1
inline convType defaultConvType(i64 kernel_size, i64 pparallel) {
    return (kernel_size > 3 || pparallel > 1) ? FFT : NAIVE_FAST;
}

class vgg : public neuralNetwork {
public:
    // `network` is either a file name or the token list itself, e.g. "64 M 128 M 256 256 M 512 512 M 512 512 M"
    vgg(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &i_filename, const string &c_filename,
        const string &o_filename, const string &network);
};

class vgg16 : public neuralNetwork {
public:
    vgg16(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
          const string &c_filename, const string &o_filename);
};

class vgg11 : public neuralNetwork {
public:
    vgg11(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
          const string &c_filename, const string &o_filename);
};

class lenet : public neuralNetwork {
public:
    lenet(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
          const string &c_filename, const string &o_filename);
};

class lenetCifar : public neuralNetwork {
public:
    lenetCifar(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, poolType pool_ty, const string &i_filename,
               const string &c_filename, const string &o_filename);
};

// Small configurable net. Spec tokens (space separated):
//   C<co>:<k>:<pad>:<f|s|n>   convolution, algorithm f = FFT, s = NAIVE_FAST, n = NAIVE (mul+add)
//   M / A                     2x2 stride-2 max / average pooling (closes the current conv section)
//   F<co>                     fully connected layer
class customNet : public neuralNetwork {
public:
    customNet(i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel, const string &spec);
};

// factory used by the C-ABI driver: "lenet", "lenet.avg", "lenetCifar", "vgg11", "vgg16", "vgg:<tokens>", "custom:<spec>"
neuralNetwork *makeModel(const string &name, i64 psize_x, i64 psize_y, i64 pchannel, i64 pparallel);
