// Plain types shared by the kernel translation units of libzkcnn_hip.so.
#pragma once
#include "fr_dev.cuh"

#define ZK_BLOCK 256
#define ZK_MAX_VARS 30

struct fr_vec {                       // challenge vector passed by value in the kernel argument block
    fr_t v[ZK_MAX_VARS];
};

struct __align__(16) gate_rec {
    uint32_t g;        // output gate (index into beta_g)
    uint32_t key;      // destination table entry (sorted, non-decreasing)
    uint32_t aux;      // phase 1: resolved index of the OTHER operand's value; phase 2: u (index into beta_u)
    uint32_t meta;     // bits 0..8: two_mul index; bit 9: has value operand (bin gate, phase 1);
                       // bit 10: operand lives in the previous layer (else layer 0); bit 11: padding record
};
#define GATE_SC(m) ((m) & 0x1ffu)
#define GATE_HAS_VAL(m) (((m) >> 9) & 1u)
#define GATE_IN_PREV(m) (((m) >> 10) & 1u)
#define GATE_DUMMY(m) (((m) >> 11) & 1u)     // padding record: contributes nothing (runs of equal keys are padded to multiples of GATE_GROUP)
#define GATE_GROUP 4u
#define GATE_NOKEY 0xffffffffu

// a direct-convolution layer whose gate list was checked against the pattern of conv_kernels.cuh (sizes, and the bit widths of the index fields)
struct conv_desc {
    uint32_t pp, CO, CI, nxi, nyi, nxo, nyo, m, pad, ls, wstart;
    int32_t bx_i, by_i, bc_i, bx_o, by_o, bc_o;
};
