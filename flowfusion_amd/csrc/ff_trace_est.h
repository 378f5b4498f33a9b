// ff_trace_est.h -- the Hutch++ and XTrace divergence estimators on one sample's Jacobian, shared by the gfx950 kernel
// (ff_adaptive.hip: trace_estimate_kernel) and the host entry point the CPU tests call (ff_api.cpp: ff_trace_estimate_host).
//
// What is restated here, and from where: ScoreModel.forward's two estimator branches, flowfusion/diffusion.py:336-400
// (Hutch++: sketch Y = A S, Q = qr(Y), sum_i q_i^T A q_i + mean_g u^T A u with u = (I - Q Q^T) g) and :402-481 (XTrace:
// Y = A O, (Q, R) = qr(Y), the leave-one-out combination :457-481), with A v = J^T v.  The reference obtains the
// products by reverse mode and factorises with `torch.linalg.qr(mode="reduced")`; here the whole matrix A = J^T is on
// hand (ff_ode_args.jac_out) and the QR is LAPACK's Householder scheme (sgeqr2 + sorg2r: beta = -sign(alpha) |x|,
// tau = (beta - alpha) / beta, H = I when the column is already zero below the diagonal), which is what
// `torch.linalg.qr` runs on the CPU -- the reference's estimate depends on those conventions where a sketch is rank
// deficient (two +-1 probes equal up to sign), see flowfusion_amd/trace_estimators.py `thin_qr`, the torch statement of
// the same (kept for right-hand sides evaluated outside the library and as the comparator of the tests).
// Everything is fp32; sums run in index order (torch's matmuls order them differently: agreement is to rounding).
//
// One work item = one (evaluation row, sample).  Everything an item touches is addressed through a base pointer and a
// stride -- its matrix `A(i, k) = A[(i D + k) astride]`, its probes, its scratch `W(i) = ws[i stride]` -- so that the same
// arithmetic runs from three layouts: plain arrays on the host (strides 1), global memory with the scratch laid out
// item-fastest (generic device kernel), and LDS tiles transposed to item-fastest with an odd pitch (fast device kernel:
// consecutive lanes on consecutive banks).  `DC` > 0 fixes the dimension at compile time (loops unroll, loads batch).
#pragma once
#include <math.h>
#include <stdint.h>
#include "flowfusion_amd.h"
#include "ff_layout.h"

namespace ff {
namespace trace {

// floats of workspace per work item
FF_HD size_t workspace_per_item(int kind, int D, int r)
{
    const size_t d = (size_t)D, k = (size_t)r;
    return kind == FF_TRACE_XTRACE ? d * k + 2 * k + 6 * k * k : d * k + 2 * k + d;
}

struct Item {
    const float* A;          // A = J^T: element (i, k) at A[(i * D + k) * astride]
    size_t astride;
    const float* p0;         // probes S (Hutch++) or O (XTrace): element (c, i) at p0[c * pstride0 + i * pk0]
    size_t pstride0, pk0;
    const float* p1;         // probes G (Hutch++): element (c, i) at p1[c * pstride1 + i * pk1]
    size_t pstride1, pk1;
    float* ws;               // workspace base of this item
    size_t stride;           // floats between consecutive workspace words of this item
    int D, r, m;
};

#define FF_TW(i) it.ws[(size_t)(i) * it.stride]
#define FF_TA(i, k) it.A[(size_t)((i) * D + (k)) * it.astride]
#define FF_P0(c, i) it.p0[(size_t)(c) * it.pstride0 + (size_t)(i) * it.pk0]

// Householder QR of the D x r matrix held at Q(i, c) = W(i * r + c); on return Q holds the reduced Q factor, `tau` the r
// reflector scalars at W(tau0 + j), and -- if R0 >= 0 -- R (r x r, upper, row-major) at W(R0 + a * r + b).
template <int DC>
FF_HD void thin_qr(const Item& it, int tau0, int R0)
{
#pragma clang fp contract(off)
    const int D = DC ? DC : it.D, r = it.r;
#define FF_Q(i, c) FF_TW((i) * r + (c))
    for (int j = 0; j < r; ++j) {
        const float alpha = FF_Q(j, j);
        float ss = 0.f;
        for (int i = j + 1; i < D; ++i) ss += FF_Q(i, j) * FF_Q(i, j);
        const float xnorm = sqrtf(ss);
        float tau = 0.f, beta = alpha;
        if (xnorm > 0.f) {
            beta = -copysignf(sqrtf(alpha * alpha + xnorm * xnorm), alpha);
            tau = (beta - alpha) / beta;
            const float denom = alpha - beta;
            for (int i = j + 1; i < D; ++i) FF_Q(i, j) = FF_Q(i, j) / denom;         // v (v_j = 1 implied)
        }
        for (int c = j + 1; c < r; ++c) {                                        // H_j on the remaining columns
            float s = FF_Q(j, c);
            for (int i = j + 1; i < D; ++i) s += FF_Q(i, j) * FF_Q(i, c);
            FF_Q(j, c) = FF_Q(j, c) - tau * s;
            for (int i = j + 1; i < D; ++i) FF_Q(i, c) = FF_Q(i, c) - (tau * FF_Q(i, j)) * s;
        }
        FF_Q(j, j) = beta;
        FF_TW(tau0 + j) = tau;
    }
    if (R0 >= 0)
        for (int a = 0; a < r; ++a)
            for (int b = 0; b < r; ++b) FF_TW(R0 + a * r + b) = b >= a ? FF_Q(a, b) : 0.f;
    // the explicit Q, in place (sorg2r): column j of H_0 .. H_{r-1} applied to the first r unit vectors
    for (int j = r - 1; j >= 0; --j) {
        const float tau = FF_TW(tau0 + j);
        for (int c = j + 1; c < r; ++c) {
            float s = 0.f;                                                       // row j of a later column is zero so far
            for (int i = j + 1; i < D; ++i) s += FF_Q(i, j) * FF_Q(i, c);
            FF_Q(j, c) = -(tau * s);
            for (int i = j + 1; i < D; ++i) FF_Q(i, c) = FF_Q(i, c) - (tau * FF_Q(i, j)) * s;
        }
        for (int i = j + 1; i < D; ++i) FF_Q(i, j) = -(tau * FF_Q(i, j));
        FF_Q(j, j) = 1.f - tau;
        for (int i = 0; i < j; ++i) FF_Q(i, j) = 0.f;
    }
}

// One product with the item's matrix: out(i) = sum_k A(i, k) v(k) handed to `sink(i, value)` row by row.  With a
// compile-time dimension the operand vector is first copied into registers (each of its entries is used D times) and the
// dot products unroll; with a run-time dimension it is read in place.
template <int DC, class V, class S>
FF_HD void matvec(const Item& it, V&& v, S&& sink)
{
    const int D = DC ? DC : it.D;
    if constexpr (DC > 0) {
        float vr[DC];
#pragma unroll
        for (int k = 0; k < DC; ++k) vr[k] = v(k);
        for (int i = 0; i < DC; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < DC; ++k) s += FF_TA(i, k) * vr[k];
            sink(i, s);
        }
    } else {
        for (int i = 0; i < D; ++i) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) s += FF_TA(i, k) * v(k);
            sink(i, s);
        }
    }
}

// Y = A P: the sketch, written over the Q storage
template <int DC>
FF_HD void sketch(const Item& it)
{
    const int r = it.r;
    for (int c = 0; c < r; ++c)
        matvec<DC>(it, [&](int k) { return FF_P0(c, k); }, [&](int i, float s) { FF_Q(i, c) = s; });
}

// flowfusion/diffusion.py:336-400
template <int DC>
FF_HD float hutchpp(const Item& it)
{
    const int D = DC ? DC : it.D, r = it.r, m = it.m;
    const int tau0 = D * r, w0 = tau0 + r, u0 = w0 + r;
    sketch<DC>(it);
    thin_qr<DC>(it, tau0, -1);
    // sum_c q_c^T A q_c                                                         (:381-386)
    float trace_range = 0.f;
    for (int c = 0; c < r; ++c)
        matvec<DC>(it, [&](int k) { return FF_Q(k, c); }, [&](int i, float s) { trace_range += FF_Q(i, c) * s; });
    // u = (I - Q Q^T) g for every g, then u^T A u                               (:388-398)
    float trace_rest = 0.f;
    for (int g = 0; g < m; ++g) {
        const float* gv = it.p1 + (size_t)g * it.pstride1;
        for (int c = 0; c < r; ++c) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i) s += FF_Q(i, c) * gv[(size_t)i * it.pk1];
            FF_TW(w0 + c) = s;
        }
        for (int i = 0; i < D; ++i) {
            float s = 0.f;
            for (int c = 0; c < r; ++c) s += FF_Q(i, c) * FF_TW(w0 + c);
            FF_TW(u0 + i) = gv[(size_t)i * it.pk1] - s;
        }
        matvec<DC>(it, [&](int k) { return FF_TW(u0 + k); }, [&](int i, float s) { trace_rest += FF_TW(u0 + i) * s; });
    }
    return trace_range + trace_rest / (float)m;                                  // (:400)
}

// flowfusion/diffusion.py:402-481
template <int DC>
FF_HD float xtrace(const Item& it)
{
    const int D = DC ? DC : it.D, k = it.r;
    const int tau0 = D * k, z0 = tau0 + k, R0 = z0 + k, H0 = R0 + k * k, Wm0 = H0 + k * k, T0 = Wm0 + k * k,
              S0 = T0 + k * k, X0 = S0 + k * k;
    const int r = k;                                                             // (FF_Q's column count)
    sketch<DC>(it);
    thin_qr<DC>(it, tau0, R0);
#define FF_M(base, a, b) FF_TW((base) + (a) * k + (b))
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) { FF_M(H0, a, b) = 0.f; FF_M(T0, a, b) = 0.f; }
    // Z = A Q row by row; H = Q^T Z (:451), T = Z^T O (:455), W = Q^T O (:453)
    // (column c of Z at a time: H(:, c) = Q^T z, T(c, :) = z^T O)
    for (int c = 0; c < k; ++c)
        matvec<DC>(it, [&](int j) { return FF_Q(j, c); }, [&](int i, float z) {
            for (int a = 0; a < k; ++a) {
                FF_M(H0, a, c) += FF_Q(i, a) * z;
                FF_M(T0, c, a) += z * FF_P0(a, i);
            }
        });
    (void)z0;
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            float s = 0.f;
            for (int i = 0; i < D; ++i) s += FF_Q(i, a) * FF_P0(b, i);
            FF_M(Wm0, a, b) = s;
        }
    // S^T = R^-1 by back substitution (:457), rows to unit length (:459); S(a, i) = Rinv(i, a) is stored directly
    for (int col = 0; col < k; ++col) {
        for (int a = k - 1; a >= 0; --a) {
            float s = a == col ? 1.f : 0.f;
            for (int b = a + 1; b < k; ++b) s -= FF_M(R0, a, b) * FF_M(S0, col, b);   // Rinv(b, col) sits at S(col, b)
            FF_M(S0, col, a) = s / FF_M(R0, a, a);
        }
    }
    // (row i of Rinv = S(:, i): the norm runs over the first index)
    for (int i = 0; i < k; ++i) {
        float ss = 0.f;
        for (int a = 0; a < k; ++a) ss += FF_M(S0, a, i) * FF_M(S0, a, i);
        const float nrm = sqrtf(ss);
        for (int a = 0; a < k; ++a) FF_M(S0, a, i) = FF_M(S0, a, i) / nrm;
    }
    float tr_h = 0.f;
    for (int a = 0; a < k; ++a) tr_h += FF_M(H0, a, a);                          // (:463)
    float total = 0.f;
    for (int i = 0; i < k; ++i) {
        float sw = 0.f;
        for (int a = 0; a < k; ++a) sw += FF_M(S0, a, i) * FF_M(Wm0, a, i);
        for (int a = 0; a < k; ++a) FF_M(X0, a, i) = FF_M(Wm0, a, i) - sw * FF_M(S0, a, i);     // (:467)
        float shs = 0.f, xhx = 0.f, sr = 0.f, tx = 0.f;
        for (int a = 0; a < k; ++a) {
            float hs = 0.f, hx = 0.f;
            for (int b = 0; b < k; ++b) {
                hs += FF_M(H0, a, b) * FF_M(S0, b, i);
                hx += FF_M(H0, a, b) * FF_M(X0, b, i);
            }
            shs += FF_M(S0, a, i) * hs;                                          // (:469)
            xhx += FF_M(X0, a, i) * hx;                                          // (:471)
            sr += FF_M(S0, a, i) * FF_M(R0, a, i);                               // (:475)
            tx += FF_M(T0, a, i) * FF_M(X0, a, i);                               // (:477)
        }
        total += tr_h - shs + sw * sr - tx + xhx;                                // (:479)
    }
    return total / (float)k;                                                     // (:481)
#undef FF_M
}

#undef FF_Q
#undef FF_TW
#undef FF_TA
#undef FF_P0

template <int DC>
FF_HD float estimate(int kind, const Item& it) { return kind == FF_TRACE_XTRACE ? xtrace<DC>(it) : hutchpp<DC>(it); }

} // namespace trace
} // namespace ff
