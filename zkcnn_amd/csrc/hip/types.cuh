// Plain types shared by the kernel translation units of libzkcnn_hip.so.
#pragma once
#include "fr_dev.cuh"

#define ZK_BLOCK 256
#define ZK_MAX_VARS 30
// A launch of few workgroups is a link of a proof's dependent chain (a small round, a phase's initialisation): its waves ask for issue priority
// over the bulk kernels of OTHER proofs that share their SIMDs (multi-scalar multiplications, streaming rounds). Wide launches are bulk themselves.
#define ZK_LATENCY_GRID 128u
#define ZK_LATENCY_PRIO() do { if (gridDim.x <= ZK_LATENCY_GRID) __builtin_amdgcn_s_setprio(3); } while (0)

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

// ---- layer-0 combine (k_eq_halves_multi / k_liu_gather) and the small eq tables of the structured convolutions ----
struct liu_table { fr_vec r; fr_t init; int32_t n, fh, sh, pad_; };      // one per (layer, side) that touches layer 0
#define LIU_HALF_STRIDE 8192u                                               // entries per half table (bit length of a subset table <= 26: vgg16 over 32 pictures)
struct liu_entry { uint32_t h, t; };                                        // index h inside table (t & 0xffffff); t >> 24 = bits of the table's low half
#define CONV_TAB_STRIDE 4096u           // entries per small eq table
// table numbers inside the context's small-table buffer
// (the tables are built by the phase's k_prep launch: prep_kernels.cuh, prep_small_block -- eq(r[0..n), .) * init for n <= 12)
enum { CT_S0 = 0, CT_A0, CT_P0, CT_S1, CT_A1, CT_P1, CT_D, CT_C, CT_PU, CT_COUNT };

