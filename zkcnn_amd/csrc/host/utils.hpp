// Host-side field helpers used by the verifier and the witness generator: eq ("beta") tables,
// the DFT-matrix MLE table, NTT, root of unity, index helpers. Same entry points as the reference
// (reference src/utils.hpp:15-47) so verifier/neuralNetwork code reads the same; own implementation.
#pragma once
#include "circuit.h"

char ceilPow2BitLength(u32 n);
char floorPow2BitLength(u32 n);

// beta_g[i] = alpha * eq(r_0, i) + beta * eq(r_1, i),  i < 2^gLength (little-endian bits)
void initBetaTable(vector<F> &beta_g, u8 gLength, const vector<F>::const_iterator &r_0,
                   const vector<F>::const_iterator &r_1, const F &alpha, const F &beta);
// beta_g[i] = init * eq(r, i)
void initBetaTable(vector<F> &beta_g, u8 gLength, const vector<F>::const_iterator &r, const F &init);

// phi_g[u] = scale * sum_g eq(rx, g) * w^{+-g*u}: MLE (in the row index) of the DFT matrix at rx
void phiGInit(vector<F> &phi_g, const vector<F>::const_iterator &rx, const F &scale, int n, bool isIFFT);

// in-place length-2^logn NTT (flag = inverse, includes the 1/len factor)
void fft(vector<F> &arr, int logn, bool flag);

F getRootOfUnit(int n);

inline bool check(long x, long y, long nx, long ny) { return 0 <= x && x < nx && 0 <= y && y < ny; }
inline long matIdx(long x, long y, long n) { return x * n + y; }
inline long cubIdx(long x, long y, long z, long n, long m) { return (x * n + y) * m + z; }
inline long tesIdx(long w, long x, long y, long z, long n, long m, long l) { return ((w * n + x) * m + y) * l + z; }
inline long sqr(long x) { return x * x; }

void initLayer(layer &circuit, long size, layerType ty);
