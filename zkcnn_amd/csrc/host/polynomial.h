// Coefficient-form round polynomials exchanged between prover and verifier
// (same public shape as reference src/polynomial.h:10-48: linear a*x+b, quadratic a*x^2+b*x+c,
// cubic a*x^3+b*x^2+c*x+d). Header-only; the unused degree-4/5 types of the reference are dropped.
#pragma once
#include "global_var.hpp"

struct linear_poly {
    F a, b;
    linear_poly() : a(F_ZERO), b(F_ZERO) {}
    linear_poly(const F &slope, const F &cst) : a(slope), b(cst) {}
    linear_poly(const F &cst) : a(F_ZERO), b(cst) {}
    F eval(const F &x) const { return a * x + b; }
    void clear() { a.clear(); b.clear(); }
    linear_poly operator+(const linear_poly &o) const { return linear_poly(a + o.a, b + o.b); }
    linear_poly operator*(const F &k) const { return linear_poly(a * k, b * k); }
};

struct quadratic_poly {
    F a, b, c;
    quadratic_poly() : a(F_ZERO), b(F_ZERO), c(F_ZERO) {}
    quadratic_poly(const F &x2, const F &x1, const F &x0) : a(x2), b(x1), c(x0) {}
    F eval(const F &x) const { return (a * x + b) * x + c; }
    void clear() { a.clear(); b.clear(); c.clear(); }
    quadratic_poly operator+(const quadratic_poly &o) const { return quadratic_poly(a + o.a, b + o.b, c + o.c); }
    quadratic_poly operator+(const linear_poly &o) const { return quadratic_poly(a, b + o.a, c + o.b); }
    quadratic_poly operator*(const F &k) const { return quadratic_poly(a * k, b * k, c * k); }
};

struct cubic_poly {
    F a, b, c, d;
    cubic_poly() : a(F_ZERO), b(F_ZERO), c(F_ZERO), d(F_ZERO) {}
    cubic_poly(const F &x3, const F &x2, const F &x1, const F &x0) : a(x3), b(x2), c(x1), d(x0) {}
    F eval(const F &x) const { return ((a * x + b) * x + c) * x + d; }
    void clear() { a.clear(); b.clear(); c.clear(); d.clear(); }
    cubic_poly operator+(const cubic_poly &o) const { return cubic_poly(a + o.a, b + o.b, c + o.c, d + o.d); }
    cubic_poly operator*(const F &k) const { return cubic_poly(a * k, b * k, c * k, d * k); }
};

inline quadratic_poly operator*(const linear_poly &p, const linear_poly &q) {
    return quadratic_poly(p.a * q.a, p.a * q.b + p.b * q.a, p.b * q.b);
}
inline cubic_poly operator*(const quadratic_poly &p, const linear_poly &q) {
    return cubic_poly(p.a * q.a, p.a * q.b + p.b * q.a, p.b * q.b + p.c * q.a, p.c * q.b);
}
