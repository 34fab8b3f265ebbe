// Shared aliases for the host side. Mirrors the names the reference sources use
// (reference src/global_var.hpp:12-48): F / G macros, F_ONE / F_ZERO, the 16 result columns.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <hyrax-bls12-381/polyCommit.hpp>

#define F Fr
#define G G1
#define F_ONE (Fr::one())
#define F_ZERO (Fr(0LL))
#define F_BYTE_SIZE (Fr::getByteSize())

// result row layout, same column ids as the reference CLI prints
enum outputColumn {
    MO_INFO_OUT_ID = 0, PSIZE_OUT_ID, KSIZE_OUT_ID, PCNT_OUT_ID, CONV_TY_OUT_ID, QS_OUT_ID, WS_OUT_ID,
    PT_OUT_ID, VT_OUT_ID, PS_OUT_ID, POLY_PT_OUT_ID, POLY_VT_OUT_ID, POLY_PS_OUT_ID,
    TOT_PT_OUT_ID, TOT_VT_OUT_ID, TOT_PS_OUT_ID, OUT_COLUMN_CNT
};

using std::string;
using std::vector;

// one row per process, filled by neuralNetwork / verifier (defined in utils.cpp)
extern thread_local vector<string> output_tb;     // per thread: sessions on different host threads do not share the row

template <typename T>
inline string to_string_wp(const T value, const int digits = 4) {
    std::ostringstream os;
    os.setf(std::ios::fixed);
    os.precision(digits);
    os << value;
    return os.str();
}
