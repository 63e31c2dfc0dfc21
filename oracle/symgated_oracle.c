/*
 * Plain-C restatement of GNNome's SymGatedGCN edge-scoring forward (eval mode).  TEST INFRASTRUCTURE ONLY:
 * built by oracle/Makefile into oracle/_build/, loaded by oracle/c_oracle.py, used by tests/ as a second,
 * independent checker (compiled with REAL=double it is the high-precision arbiter of the 1e-4 bar; with
 * REAL=float it is a sequential-order fp32 evaluation).  Nothing under gnnome_amd/ links or calls it.
 *
 * Follows, line by line (paths in lbcb-sci/GNNome):
 *   models/full_graph.py:26-29        encoders, gnn, predictor
 *   layers/processor.py:17-18         loop over the layers
 *   layers/gated_gcn_full.py:85-140   SymGatedGCN.forward - BOTH directions' gates are evaluated, as there
 *   layers/score_predictor.py:13-16   cat(x_src, x_dst, e) -> W1 -> relu -> W2 -> relu -> W3
 * DGL 0.8.1 semantics restated: u_add_v, update_all(u_mul_e/copy_e, sum) and dgl.reverse (see the header of
 * oracle/symgated_oracle.py).  Reductions over a node's edges run in edge-id order.
 *
 * params: one packed array, in state_dict order with the integer num_batches_tracked entries left out:
 *   linear1_node W[M,Fn] b[M] | linear2_node W[H,M] b[H] | linear1_edge W[M,Fe] b[M] | linear2_edge W[H,M] b[H]
 *   per layer: A_1 A_2 A_3 B_1 B_2 B_3 (W[H,H] b[H] each) | bn_h w b rm rv | bn_e w b rm rv   (LayerNorm: w b only)
 *   predictor W1[hs,3H] b1[hs] W2[32,hs] b2[32] W3[32] b3[1]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

static void linear(const real* x, int rows, int in, const real* W, const real* b, int out, real* y) {
    for (int r = 0; r < rows; ++r)
        for (int o = 0; o < out; ++o) {
            real s = 0;
            for (int k = 0; k < in; ++k) s += x[(size_t)r * in + k] * W[(size_t)o * in + k];
            y[(size_t)r * out + o] = s + b[o];
        }
}

static void relu_(real* x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0 ? x[i] : 0;
}

/* eval BatchNorm1d (norm 0): (x-rm)/sqrt(rv+eps)*w+b ; LayerNorm (norm 1): per row over H */
static void norm_rows(real* x, int rows, int H, int kind, const real* p) {
    const real eps = (real)1e-5;
    for (int r = 0; r < rows; ++r) {
        real* v = x + (size_t)r * H;
        if (kind == 0) {
            for (int c = 0; c < H; ++c) v[c] = (v[c] - p[2 * H + c]) / (real)sqrt((double)(p[3 * H + c] + eps)) * p[c] + p[H + c];
        } else {
            real mu = 0, var = 0;
            for (int c = 0; c < H; ++c) mu += v[c];
            mu /= H;
            for (int c = 0; c < H; ++c) var += (v[c] - mu) * (v[c] - mu);
            var /= H;
            for (int c = 0; c < H; ++c) v[c] = (v[c] - mu) / (real)sqrt((double)(var + eps)) * p[c] + p[H + c];
        }
    }
}

int gnnome_oracle_forward(int N, int E, int Fn, int Fe, int M, int H, int L, int hs, int norm_kind, const int* src,
                          const int* dst, const real* x, const real* e_raw, const real* params, real* logits) {
    const size_t NH = (size_t)N * H, EH = (size_t)E * H;
    const real* p = params;
#define TAKE(n) (p += (n), p - (n))
    real* t_n = malloc(sizeof(real) * ((size_t)N * M + 1));
    real* t_e = malloc(sizeof(real) * ((size_t)E * M + 1));
    real* h = malloc(sizeof(real) * (NH + 1));
    real* e = malloc(sizeof(real) * (EH + 1));
    real* proj = malloc(sizeof(real) * (5 * NH + 1)); /* A1h A2h A3h B1h B2h */
    real* B3e = malloc(sizeof(real) * (EH + 1));
    real* e_ji = malloc(sizeof(real) * (EH + 1));
    real* e_ik = malloc(sizeof(real) * (EH + 1));
    real* acc = malloc(sizeof(real) * (4 * NH + 1)); /* sum_sigma_h_f, sum_sigma_f, sum_sigma_h_b, sum_sigma_b */
    real* hn = malloc(sizeof(real) * (NH + 1));
    if (!t_n || !t_e || !h || !e || !proj || !B3e || !e_ji || !e_ik || !acc || !hn) return -1;

    /* models/full_graph.py:26-27 */
    const real *W, *b;
    W = TAKE(M * Fn); b = TAKE(M); linear(x, N, Fn, W, b, M, t_n); relu_(t_n, (size_t)N * M);
    W = TAKE(H * M); b = TAKE(H); linear(t_n, N, M, W, b, H, h);
    W = TAKE(M * Fe); b = TAKE(M); linear(e_raw, E, Fe, W, b, M, t_e); relu_(t_e, (size_t)E * M);
    W = TAKE(H * M); b = TAKE(H); linear(t_e, E, M, W, b, H, e);

    const int nnorm = norm_kind == 0 ? 4 * H : 2 * H;
    for (int l = 0; l < L; ++l) { /* layers/processor.py:17-18 */
        for (int k = 0; k < 5; ++k) { /* gated_gcn_full.py:91-96 */
            W = TAKE(H * H); b = TAKE(H);
            linear(h, N, H, W, b, H, proj + k * NH);
        }
        W = TAKE(H * H); b = TAKE(H);
        linear(e, E, H, W, b, H, B3e); /* :97 */
        const real* bn_h = TAKE(nnorm);
        const real* bn_e = TAKE(nnorm);
        const real *A1h = proj, *A2h = proj + NH, *A3h = proj + 2 * NH, *B1h = proj + 3 * NH, *B2h = proj + 4 * NH;
        for (int k = 0; k < E; ++k) /* :104-105 and :117-118 (reversed graph: u = dst, v = src) */
            for (int c = 0; c < H; ++c) {
                e_ji[(size_t)k * H + c] = (B1h[(size_t)src[k] * H + c] + B2h[(size_t)dst[k] * H + c]) + B3e[(size_t)k * H + c];
                e_ik[(size_t)k * H + c] = (B2h[(size_t)dst[k] * H + c] + B1h[(size_t)src[k] * H + c]) + B3e[(size_t)k * H + c];
            }
        norm_rows(e_ji, E, H, norm_kind, bn_e); /* :106 */
        norm_rows(e_ik, E, H, norm_kind, bn_e); /* :119 */
        for (size_t i = 0; i < EH; ++i) { /* :107-110, :120-123 */
            e_ji[i] = (e_ji[i] > 0 ? e_ji[i] : 0) + e[i];
            e_ik[i] = (e_ik[i] > 0 ? e_ik[i] : 0) + e[i];
        }
        memset(acc, 0, sizeof(real) * 4 * NH);
        for (int k = 0; k < E; ++k) /* :111-113, :124-126 */
            for (int c = 0; c < H; ++c) {
                const real sf = 1 / (1 + (real)exp(-(double)e_ji[(size_t)k * H + c]));
                const real sb = 1 / (1 + (real)exp(-(double)e_ik[(size_t)k * H + c]));
                acc[(size_t)dst[k] * H + c] += A2h[(size_t)src[k] * H + c] * sf;
                acc[NH + (size_t)dst[k] * H + c] += sf;
                acc[2 * NH + (size_t)src[k] * H + c] += A3h[(size_t)dst[k] * H + c] * sb;
                acc[3 * NH + (size_t)src[k] * H + c] += sb;
            }
        for (size_t i = 0; i < NH; ++i) /* :114, :127, :129 */
            hn[i] = A1h[i] + acc[i] / (acc[NH + i] + (real)1e-6) + acc[2 * NH + i] / (acc[3 * NH + i] + (real)1e-6);
        norm_rows(hn, N, H, norm_kind, bn_h); /* :132 */
        for (size_t i = 0; i < NH; ++i) h[i] = (hn[i] > 0 ? hn[i] : 0) + h[i]; /* :134-137 */
        memcpy(e, e_ji, sizeof(real) * EH);                                      /* :140 */
    }

    /* layers/score_predictor.py:13-16 */
    const real* W1 = TAKE(hs * 3 * H); const real* b1 = TAKE(hs);
    const real* W2 = TAKE(32 * hs); const real* b2 = TAKE(32);
    const real* W3 = TAKE(32); const real* b3 = TAKE(1);
    real* z1 = malloc(sizeof(real) * hs);
    for (int k = 0; k < E; ++k) {
        for (int o = 0; o < hs; ++o) {
            real s = 0;
            const real* w = W1 + (size_t)o * 3 * H;
            for (int c = 0; c < H; ++c) s += h[(size_t)src[k] * H + c] * w[c];
            for (int c = 0; c < H; ++c) s += h[(size_t)dst[k] * H + c] * w[H + c];
            for (int c = 0; c < H; ++c) s += e[(size_t)k * H + c] * w[2 * H + c];
            s += b1[o];
            z1[o] = s > 0 ? s : 0;
        }
        real out = 0;
        for (int j = 0; j < 32; ++j) {
            real s = 0;
            for (int o = 0; o < hs; ++o) s += z1[o] * W2[(size_t)j * hs + o];
            s += b2[j];
            out += (s > 0 ? s : 0) * W3[j];
        }
        logits[k] = out + b3[0];
    }
    free(z1); free(t_n); free(t_e); free(h); free(e); free(proj); free(B3e); free(e_ji); free(e_ik); free(acc); free(hn);
    return 0;
}
