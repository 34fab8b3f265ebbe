// Kernels of the resident witness program (zk_witness_program_upload / zk_witness_rerun): the auxiliary witnesses of
// layer 0 -- bits, signs, running maxima, bits of window sums (reference src/neuralNetwork.cpp:652-728: prepareDecmpBit, prepareFieldBit,
// prepareSignBit, prepareMax) -- computed from layer values that never leave HBM, and the activation ranges that fix the quantisation
// scales (reference src/neuralNetwork.cpp:967-977). Integer work on 32-byte field elements; HBM / latency bound, a few launches per layer.
#pragma once
#include "fr_dev.cuh"

struct wit_op { uint32_t src, dst; uint8_t src_layer, op, shift, pad_; };      // == zk_witness_op == host witnessOp
enum { WIT_BIT = 0, WIT_SIGN = 1, WIT_MAX = 2, WIT_SUM_BIT = 3 };
enum { WIT_FLAG_WIDE = 1u };                                                    // a value did not fit 63 bits: the host's int64 view would differ

// signed representative of a field element: |x| (low 64 bits), its sign, and whether |x| needs more than 63 bits
__device__ __forceinline__ uint64_t fr_signed_u64(const fr_t &mont, bool &neg, bool &wide) {
    const uint32_t half[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};   // (r - 1) / 2
    fr_t one_raw = fr_zero();
    one_raw.v[0] = 1;
    fr_t c = fr_mul(mont, one_raw);                  // out of Montgomery form
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t x = (uint64_t) half[i] - c.v[i] - borrow;
        borrow = (x >> 32) & 1;
    }
    neg = borrow != 0;
    if (neg) {
        const uint32_t m[8] = FR_MOD_INIT;
        uint64_t b2 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t x = (uint64_t) m[i] - c.v[i] - b2;
            c.v[i] = (uint32_t) x;
            b2 = (x >> 32) & 1;
        }
    }
    wide = (c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7] | (c.v[1] >> 31)) != 0;
    return ((uint64_t) c.v[1] << 32) | c.v[0];
}

// one thread per operation of a step; `ops` points at the step's first operation. A run of MAX operations with one destination is
// folded by the thread of its first operation.
__global__ void __launch_bounds__(ZK_BLOCK) k_witness_aux(fr_t *val0, const fr_t *src, const wit_op *ops, uint64_t n, const uint32_t *windows,
                                                          uint32_t win, uint32_t *flags) {
    for (uint64_t idx = blockIdx.x * (uint64_t) ZK_BLOCK + threadIdx.x; idx < n; idx += (uint64_t) gridDim.x * ZK_BLOCK) {
        const wit_op op = ops[idx];
        bool neg, wide;
        if (op.op == WIT_MAX) {
            if (idx > 0 && ops[idx - 1].dst == op.dst) continue;
            uint64_t best = 0;
            fr_t best_v = fr_zero();
            for (uint64_t j = idx; j < n && ops[j].dst == op.dst; ++j) {
                const fr_t x = fr_load(src + ops[j].src);
                const uint64_t mag = fr_signed_u64(x, neg, wide);
                if (wide) atomicOr(flags, WIT_FLAG_WIDE);
                if (!neg && mag > best) { best = mag; best_v = x; }
            }
            fr_store(val0 + op.dst, best_v);
            continue;
        }
        fr_t x;
        if (op.op == WIT_SUM_BIT) {
            const uint32_t *w = windows + (size_t) op.src * win;
            x = fr_load(src + w[0]);
            for (uint32_t k = 1; k < win; ++k) x = fr_add(x, fr_load(src + w[k]));
        } else {
            x = fr_load(src + op.src);
        }
        const uint64_t mag = fr_signed_u64(x, neg, wide);
        if (wide) atomicOr(flags, WIT_FLAG_WIDE);
        const bool bit = op.op == WIT_SIGN ? neg : ((mag >> op.shift) & 1) != 0;
        fr_store(val0 + op.dst, bit ? fr_one() : fr_zero());
    }
}

// out[0] = largest non-negative value, out[1] = largest magnitude of a negative value among v[0..n) (both zeroed by the caller)
__global__ void __launch_bounds__(ZK_BLOCK) k_witness_range(unsigned long long *out, const fr_t *v, uint64_t n, uint32_t *flags) {
    unsigned long long mx = 0, mn = 0;
    bool any_wide = false;
    for (uint64_t i = blockIdx.x * (uint64_t) ZK_BLOCK + threadIdx.x; i < n; i += (uint64_t) gridDim.x * ZK_BLOCK) {
        bool neg, wide;
        const uint64_t mag = fr_signed_u64(fr_load(v + i), neg, wide);
        any_wide |= wide;
        if (neg) mn = mag > mn ? mag : mn;
        else mx = mag > mx ? mag : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long a = __shfl_down(mx, d), b = __shfl_down(mn, d);
        mx = a > mx ? a : mx;
        mn = b > mn ? b : mn;
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx) atomicMax(out, mx);
        if (mn) atomicMax(out + 1, mn);
    }
    if (any_wide) atomicOr(flags, WIT_FLAG_WIDE);
}

// out[y] = 1 + index of the last non-zero entry of segment y (0 if it is all zero); out zeroed by the caller. grid (blocks, segments)
struct wit_segment { const fr_t *p; uint64_t n; };
__global__ void __launch_bounds__(ZK_BLOCK) k_last_nonzero(unsigned long long *out, const wit_segment *seg) {
    const wit_segment s = seg[blockIdx.y];
    unsigned long long last = 0;
    for (uint64_t i = blockIdx.x * (uint64_t) ZK_BLOCK + threadIdx.x; i < s.n; i += (uint64_t) gridDim.x * ZK_BLOCK)
        if (!fr_is_zero(fr_load(s.p + i))) last = i + 1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_down(last, d);
        last = o > last ? o : last;
    }
    if ((threadIdx.x & 63) == 0 && last) atomicMax(out + blockIdx.y, last);
}
