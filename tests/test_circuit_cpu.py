"""`layeredCircuit::initSubset` (zkcnn_amd/csrc/host/circuit.cpp) against an INDEPENDENT restatement of reference src/circuit.cpp:4-88 in numpy.

The round-3 review's point: the CPU oracle compiles the product's own circuit.cpp, so a bug in the subset renumbering -- the order in which a layer's
references into layer 0 get their dense numbers (first use; uni gates before bin gates; u and v numbered apart), the table bit lengths derived from
the subset sizes, which operand tables exist -- is invisible to every transcript comparison. Here the circuit is taken as the generator + initSubset
left it (`oracle_session_layer_dump`), the gates' raw layer-0 indices are recovered through ori_id_u / ori_id_v, and the reference's algorithm,
restated from its source with array operations (no code shared with circuit.cpp), must reproduce every subset map, every renumbered operand, every
size and bit length. Direct and FFT convolutions, both poolings, several pictures per circuit, LeNet5, vgg11 at quarter width (8e6 gates)."""
import ctypes
import math

import numpy as np
import pytest

import zkcnn_amd
from tests import oracle_ffi

FFT, IFFT, DOT_PROD = 1, 2, 9            # layerType values (reference src/circuit.h:35-37)

MODELS = [
    ("custom:F8 F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1),
    ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    ("custom:C3:3:1:f A C2:3:1:f M F6 F3", (8, 8, 1), 3),
    ("lenet", (32, 32, 1), 1),
    ("lenetCifar", (32, 32, 3), 1),
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
]


def ceil_pow2_bit_length(n):
    """reference src/utils.cpp:23-25: ceil(log(n) / log(2.)) in double arithmetic; -1 for n = 0"""
    return -1 if n == 0 else int(math.ceil(math.log(n) / math.log(2.0)))


def first_use_numbering(refs):
    """dense numbers in first-use order for a sequence of raw indices: (ori_id, number of every element)"""
    if refs.size == 0:
        return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32)
    uniq, first, inverse = np.unique(refs, return_index=True, return_inverse=True)
    rank = np.empty(uniq.size, dtype=np.uint32)
    rank[np.argsort(first, kind="stable")] = np.arange(uniq.size, dtype=np.uint32)      # the value first seen earliest gets number 0
    ori = np.empty(uniq.size, dtype=np.uint32)
    ori[rank] = uniq
    return ori, rank[inverse]


def dump(lib, h, i):
    meta = (ctypes.c_int64 * 18)()
    assert lib.oracle_session_layer_dump(ctypes.c_void_p(h), ctypes.c_int32(i), meta, None, None, None, None) == 0
    m = list(meta)
    uni = np.zeros((m[2], 4), dtype=np.uint32)
    bn = np.zeros((m[3], 5), dtype=np.uint32)
    ou = np.zeros(m[4], dtype=np.uint32)
    ov = np.zeros(m[6], dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None      # noqa: E731
    assert lib.oracle_session_layer_dump(ctypes.c_void_p(h), ctypes.c_int32(i), meta, p(uni), p(bn), p(ou), p(ov)) == 0
    return m, uni, bn, ou, ov


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_init_subset_against_an_independent_restatement(built, model, pic, pp):
    lib = oracle_ffi.load().lib
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        n_layers = o.prove(seed=1, mode=zkcnn_amd.MODE_DRIVE_ONLY, want_transcript=False)[0].n_layers
        prev = dump(lib, o.h, 0)[0]
        for i in range(1, n_layers):
            m, uni, bn, ou, ov = dump(lib, o.h, i)
            ty, size = m[0], m[1]
            # raw layer-0 indices of the operands, as the generator emitted them (initSubset overwrote them with subset numbers)
            uni_l0 = uni[:, 2] == 0
            bin_u_l0 = bn[:, 4] == 0                      # binGate::getLayerIdU: layer 0 iff l == 0      (reference src/circuit.h:31)
            bin_v_l0 = (bn[:, 4] & 1) == 0                # binGate::getLayerIdV: layer 0 iff l is even   (reference src/circuit.h:32)
            assert (uni[uni_l0, 1] < m[4]).all() and (bn[bin_u_l0, 1] < m[4]).all() and (bn[bin_v_l0, 2] < m[6]).all()
            raw_u = np.concatenate([ou[uni[uni_l0, 1]], ou[bn[bin_u_l0, 1]]])       # visit order: every uni gate, then every bin gate (:15-27, :29-38)
            raw_v = ov[bn[bin_v_l0, 2]]
            # ---- the reference's algorithm on the raw gates ----
            want_ou, num_u = first_use_numbering(raw_u)
            want_ov, num_v = first_use_numbering(raw_v)
            assert np.array_equal(want_ou, ou), f"layer {i}: ori_id_u is not the first-use order of the layer's layer-0 operands"
            assert np.array_equal(want_ov, ov), f"layer {i}: ori_id_v"
            assert np.array_equal(num_u, np.concatenate([uni[uni_l0, 1], bn[bin_u_l0, 1]])) and np.array_equal(num_v, bn[bin_v_l0, 2])
            assert np.unique(ou).size == ou.size and np.unique(ov).size == ov.size            # a subset holds every wire once
            assert m[4] == ou.size and m[6] == ov.size
            assert m[9] == ceil_pow2_bit_length(ou.size) and m[11] == ceil_pow2_bit_length(ov.size)
            has_u = ty in (FFT, IFFT) or bool((~uni_l0).any()) or bool((~bin_u_l0).any())       # :12, :26, :47
            has_v = bool((~bin_v_l0).any())                                                     # :48
            fb = m[15]
            if not has_u:
                want_u1 = (0, -1)
            elif ty == FFT:
                want_u1 = (1 << (fb - 1), fb - 1)                                                # :56-58
            elif ty == IFFT:
                want_u1 = (1 << fb, fb)
            else:
                want_u1 = (prev[1], prev[8])
            if not has_v:
                want_v1 = (0, -1)
            elif ty == DOT_PROD:
                want_v1 = (prev[1] >> fb, prev[8] - fb)                                          # :74-76
            else:
                want_v1 = (prev[1], prev[8])
            assert (m[5], m[10]) == want_u1, f"layer {i}: u table of the previous layer"
            assert (m[7], m[12]) == want_v1, f"layer {i}: v table of the previous layer"
            # layer::updateSize (reference src/circuit.h:70-77)
            assert m[13] == max(m[9], m[10]) and m[14] == (max(m[11], m[12]) if m[16] else 0)
            # operands in the previous layer stay raw and inside it; outputs inside this layer
            assert (uni[~uni_l0, 1] < max(prev[1], 1)).all() and (uni[:, 0] < size).all() and (bn[:, 0] < max(size, 1)).all()
            prev = m
