/*
 * rwkv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, dtype-faithful CPU restatement of the reference's per-timestep
 * RWKV-v4 uint8 forward pass (reference: include/rwkv/cuda/rwkv.cu:493-593,
 * plus the kernels it launches).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker
 * or as the reported CPU baseline -- never as a fallback for the HIP engine.
 *
 * Parity pin: the reference ships no golden vectors, so this
 * restatement is pinned against the reference's own kernel file compiled
 * unmodified with hipcc (oracle/_ref, see oracle/Makefile) -- live on the GPU
 * box (tests/test_ref_parity_gpu.py) and through the fixtures that run produced
 * (tests/golden/, generator: tools/make_golden.py).
 *
 * Each function cites the reference lines it restates.  Where the reference's
 * result depends on the order of float atomicAdd's (nondeterministic on the
 * device) this file fixes the order to ascending block index.
 *
 * Arithmetic notes (all verified against the reference source):
 *   - hipcc/nvcc contract a*b+c into an fma by default, so the GEMV inner
 *     statement  y += x * ((w*r) + o)  is two fmaf's; written explicitly here.
 *   - rwkv.cu:250 calls exp() on a *float* argument -> float expf();
 *     rwkv.cu:212,247,248,251,252 call exp() on doubles.
 *   - rwkv.cu:44 sqrt() of a float -> sqrtf(); rwkv.cu:43 divides in double,
 *     rwkv.cu:444 divides mean/emb in float.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define V_SIZE 50277ULL
#define JSPLIT 16ULL   /* MM8_ONE_JSPLIT, rwkv.cu:21 */
#define EMBBLOCK 16ULL /* rwkv.cu:24: each device thread owns 16 consecutive elements */
#ifndef KCHUNK
#define KCHUNK 64ULL
#endif
/* KCHUNK: output columns per OpenMP work item (host parallelisation only; the summation order per column does not depend on it) */

enum { MODE_PARRALEL = 0, MODE_GPT = 1 }; /* enums/enum.h:2-5 */

/* tensor slots, enums/enum.h:7-55 */
enum {
    X, EMBED, LAYERNORMS, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD,
    BUFFER1, BUFFER2, BUFFER3, BUFFER4, MIXK, MIXV, MIXR, KM, VM, RM,
    KR, VR, RR, O1, O2, O3, ATTOUT, ATTOUTR, ATTOUTO, FFNMIXK, FFNMIXV,
    FFNK, FFNV, FFNR, FFNKR, FFNVR, FFNRR, FFNKO, FFNVO, FFNRO,
    FFNKBUFFER, FFNVBUFFER, FFNRBUFFER, DECAY, BONUS, HEAD, HEADR, HEADO,
    N_TENSORS
};

/* element byte sizes, rwkv.h:84 */
static const uint64_t k_types[N_TENSORS] = {
    8, 4, 8, 8, 8, 8, 8, 8, 8, 4, 4, 4, 8, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4,
    4, 1, 4, 4, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4, 4, 8, 8, 4, 8, 8, 1, 4, 4};

/* element counts, rwkv.h:124-128 (a = n_layers, b = n_embed) */
uint64_t oracle_tensor_elems(uint64_t i, uint64_t a, uint64_t b)
{
    const uint64_t s[N_TENSORS] = {
        b, V_SIZE * b, 4 * (a + 1) * b, a * b, a * b, a * b, a * b, a * b, b, V_SIZE, b, b,
        a * b, a * b, a * b, a * b * b, a * b * b, a * b * b, a * b, a * b, a * b, a * b, a * b, a * b,
        a * b * b, a * b, a * b, a * b, a * b, a * b * b * 4, a * b * b * 4, a * b * b,
        a * b, a * b * 4, a * b, a * b, a * b * 4, a * b, b, b, b * 4, a * b, a * b, V_SIZE * b, b, b};
    return s[i];
}
uint64_t oracle_tensor_type(uint64_t i) { return k_types[i]; }

typedef struct {
    uint64_t L, D;
    void *t[N_TENSORS]; /* host pointers, file layout (Appendix A of SURVEY.md) */
    int owns;           /* 1 if t[] were malloc'ed by oracle_open_file */
} oracle_model;

/* ---- model.bin reader: rwkv.cu:638-717 (header 2 x u64, then 46 raw tensors) ---- */
oracle_model *oracle_open_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    oracle_model *m = (oracle_model *)calloc(1, sizeof(*m));
    if (fread(&m->L, 8, 1, f) != 1 || fread(&m->D, 8, 1, f) != 1) { fclose(f); free(m); return NULL; }
    m->owns = 1;
    for (int i = 0; i < N_TENSORS; i++) {
        uint64_t bytes = oracle_tensor_elems(i, m->L, m->D) * k_types[i];
        m->t[i] = malloc(bytes ? bytes : 1);
        if (!m->t[i] || fread(m->t[i], 1, bytes, f) != bytes) { fclose(f); return NULL; }
    }
    fclose(f);
    return m;
}

/* borrow 46 host pointers in file layout (scratch/state slots may be NULL) */
oracle_model *oracle_from_ptrs(uint64_t L, uint64_t D, void **ptrs)
{
    oracle_model *m = (oracle_model *)calloc(1, sizeof(*m));
    m->L = L; m->D = D; m->owns = 0;
    for (int i = 0; i < N_TENSORS; i++) m->t[i] = ptrs[i];
    return m;
}

void oracle_close(oracle_model *m)
{
    if (!m) return;
    if (m->owns) for (int i = 0; i < N_TENSORS; i++) free(m->t[i]);
    free(m);
}
uint64_t oracle_n_layers(const oracle_model *m) { return m->L; }
uint64_t oracle_n_embed(const oracle_model *m) { return m->D; }

/* ---- meanvar: rwkv.cu:412-465 (addall, variance).  Each device thread owns 16
 * consecutive elements and accumulates them into a *float* partial (the double
 * element is added in double and rounded back to float each step); partials are
 * combined by float atomicAdd -- here in ascending thread order. ---- */
void oracle_meanvar(uint64_t D, const double *a, uint64_t T, float *mean, float *var)
{
    for (uint64_t t = 0; t < T; t++) {
        float acc = 0.0f;
        for (uint64_t i0 = 0; i0 < D; i0 += EMBBLOCK) {
            float mini = 0.0f;
            for (uint64_t i = i0; i < i0 + EMBBLOCK && i < D; i++) mini = (float)((double)mini + a[i + t * D]);
            acc += mini;
        }
        mean[t] = acc;
    }
    for (uint64_t t = 0; t < T; t++) {
        float acc = 0.0f;
        const float m = mean[t] / (float)D; /* rwkv.cu:444: float / ull -> float */
        for (uint64_t i0 = 0; i0 < D; i0 += EMBBLOCK) {
            float mini = 0.0f;
            for (uint64_t i = i0; i < i0 + EMBBLOCK && i < D; i++) {
                double dlt = a[i + t * D] - (double)m;
                mini = (float)((double)mini + dlt * dlt);
            }
            acc += mini;
        }
        var[t] = acc;
    }
}

/* ---- cuda_layernorm: rwkv.cu:40-57.  Unbiased variance (D-1), no epsilon. ---- */
void oracle_layernorm(uint64_t D, const double *x, const double *ln, uint64_t row,
                      const float *mean, const float *var, double *out, uint64_t T)
{
    for (uint64_t t = 0; t < T; t++) {
        const double xmean = (double)mean[t] / (double)D;
        const double x2 = (double)sqrtf(var[t] / (float)(D - 1));
        for (uint64_t i = 0; i < D; i++)
            out[i + D * t] = ln[D * row + i] * ((x[i + D * t] - xmean) / x2) + ln[D * (row + 1) + i];
    }
}

/* ---- kernelc_mm8_one: rwkv.cu:267-295.  y[t][k] += sum over 16 j-slabs of a float
 * partial; x is double (cast to float per element) or float.  y must be
 * pre-initialised by the caller (zero or residual), exactly as in the reference.
 * Loop nest is (slab, j, k) so that k vectorises; the per-(slab,k) summation
 * order over j is the reference's. ---- */
static void mm8_core(uint64_t N, uint64_t M, const double *xd, const float *xf,
                     const uint8_t *w, float *y, const float *r, const float *o,
                     uint64_t layer, uint64_t T)
{
    const uint64_t slab = (N + JSPLIT - 1) / JSPLIT;
    const uint8_t *wl = w + layer * N * M;
    const float *rl = r + layer * N, *ol = o + layer * N;
#pragma omp parallel
    {
        float *part = (float *)malloc(sizeof(float) * KCHUNK);
#pragma omp for schedule(dynamic)
        for (uint64_t k0 = 0; k0 < M; k0 += KCHUNK) {
            const uint64_t kn = (k0 + KCHUNK <= M) ? KCHUNK : (M - k0);
            for (uint64_t t = 0; t < T; t++) {
                for (uint64_t s = 0; s < JSPLIT; s++) {
                    uint64_t j0 = s * slab, j1 = (s + 1) * slab;
                    if (j0 > N) j0 = N;
                    if (j1 > N) j1 = N;
                    for (uint64_t k = 0; k < kn; k++) part[k] = 0.0f;
                    for (uint64_t j = j0; j < j1; j++) {
                        const float xv = xd ? (float)xd[j + N * t] : xf[j + N * t];
                        const float rj = rl[j], oj = ol[j];
                        const uint8_t *wr = wl + j * M + k0;
                        for (uint64_t k = 0; k < kn; k++)
                            part[k] = fmaf(xv, fmaf((float)wr[k], rj, oj), part[k]);
                    }
                    float *yt = y + k0 + M * t;
                    for (uint64_t k = 0; k < kn; k++) yt[k] += part[k];
                }
            }
        }
        free(part);
    }
}
void oracle_mm8_one_f64(uint64_t N, uint64_t M, const double *x, const uint8_t *w, float *y,
                        const float *r, const float *o, uint64_t layer, uint64_t T)
{ mm8_core(N, M, x, NULL, w, y, r, o, layer, T); }
void oracle_mm8_one_f32(uint64_t N, uint64_t M, const float *x, const uint8_t *w, float *y,
                        const float *r, const float *o, uint64_t layer, uint64_t T)
{ mm8_core(N, M, NULL, x, w, y, r, o, layer, T); }

/* ---- kernel_mm8_threec: rwkv.cu:58-100.  xy is [T][3][D] float. ---- */
void oracle_mm8_three(uint64_t D, const float *xy, const uint8_t *w0, const uint8_t *w1, const uint8_t *w2,
                      const float *r0, const float *r1, const float *r2,
                      const float *o0, const float *o1, const float *o2,
                      float *y0, float *y1, float *y2, uint64_t layer, uint64_t T)
{
    const uint8_t *w[3] = {w0, w1, w2};
    const float *r[3] = {r0, r1, r2}, *o[3] = {o0, o1, o2};
    float *y[3] = {y0, y1, y2};
    float *xs = (float *)malloc(sizeof(float) * D * T);
    for (int m = 0; m < 3; m++) {
        for (uint64_t t = 0; t < T; t++) memcpy(xs + t * D, xy + m * D + t * D * 3, sizeof(float) * D);
        mm8_core(D, D, NULL, xs, w[m], y[m], r[m], o[m], layer, T);
    }
    free(xs);
}

/* ---- kernel_wkvc_forward: rwkv.cu:221-259 ---- */
void oracle_wkv(uint64_t C, const double *w, const double *u, const float *k, const float *v, const float *r,
                double *y, double *aa_, double *bb_, double *pp_, uint64_t layer, uint64_t layers,
                uint64_t T, int mode)
{
    for (uint64_t t = 0; t < T; t++)
        for (uint64_t ii = 0; ii < C; ii++) {
            uint64_t so = ii + layer * C;
            if (mode == MODE_PARRALEL) so += t * C * layers;
            double aa = aa_[so], bb = bb_[so], pp = pp_[so];
            const double vv = (double)v[ii + t * C];
            const double uw = u[ii + C * layer] + w[ii + C * layer];
            const double e1 = exp(uw + (double)k[ii + t * C]);
            const double wr1 = aa + e1 * vv;
            const double wr2 = bb + e1;
            double yy = wr1 / wr2;
            /* rwkv.cu:250: exp() of a float expression -> expf */
            const float er = expf(-r[ii + t * C]);
            yy = (1.0 / (1.0 + (double)er)) * yy;
            y[ii + t * C] = yy;
            const double ek = exp((double)k[ii + t * C]);
            const double ew = exp(w[ii + C * layer]);
            aa_[so] = (aa + ek * vv) * ew;
            bb_[so] = (bb + ek) * ew;
            pp_[so] = pp;
        }
}

/* ---- mixatt: rwkv.cu:351-392 (out is float [T][3][D]; state slot <- current ln1 output) ---- */
void oracle_mixatt(uint64_t D, const double *rc, double *ddd, const double *mixk, const double *mixv,
                   const double *mixr, float *outkvr, uint64_t layer, uint64_t layers, uint64_t T, int mode)
{
    for (uint64_t t = 0; t < T; t++)
        for (uint64_t i = 0; i < D; i++) {
            uint64_t so = i + layer * D;
            if (mode == MODE_PARRALEL) so += t * D * layers;
            const double c = rc[i + t * D], p = ddd[so];
            const double mk = mixk[i + layer * D], mv = mixv[i + layer * D], mr = mixr[i + layer * D];
            outkvr[i + t * D * 3] = (float)(mk * c + (1.0 - mk) * p);
            outkvr[i + D + t * D * 3] = (float)(mv * c + (1.0 - mv) * p);
            outkvr[i + 2 * D + t * D * 3] = (float)(mr * c + (1.0 - mr) * p);
            ddd[so] = c;
        }
}

/* ---- mixffn: rwkv.cu:313-349 (outputs stay double) ---- */
void oracle_mixffn(uint64_t D, const double *rc, double *ddd, const double *mixk, const double *mixr,
                   double *outk, double *outr, uint64_t layer, uint64_t layers, uint64_t T, int mode)
{
    for (uint64_t t = 0; t < T; t++)
        for (uint64_t i = 0; i < D; i++) {
            const uint64_t so = i + layer * D + (mode == MODE_PARRALEL ? t * D * layers : 0);
            const double c = rc[i + t * D], p = ddd[so];
            const double mk = mixk[i + layer * D], mr = mixr[i + layer * D];
            outk[i + t * D] = mk * c + (1.0 - mk) * p;
            outr[i + t * D] = mr * c + (1.0 - mr) * p;
            ddd[so] = c;
        }
}

/* ---- the host driver: cuda_rwkv_parralel, rwkv.cu:493-593.
 * state[5] = {xy, aa, bb, pp, dd}, each [S][L][D] doubles (S = T in PARRALEL mode, 1 in GPT).
 * logits: [T][50277] floats. ---- */
int oracle_forward(const oracle_model *m, const uint64_t *tokens, uint64_t T, int mode,
                   double **state, float *logits)
{
    const uint64_t L = m->L, D = m->D;
    const float *embed = (const float *)m->t[EMBED];
    const double *ln = (const double *)m->t[LAYERNORMS];
    double *sxy = state[0], *saa = state[1], *sbb = state[2], *spp = state[3], *sdd = state[4];

    double *x = (double *)malloc(8 * D * T), *buffer1 = (double *)malloc(8 * D * T);
    double *ffnk_in = (double *)malloc(8 * D * T), *ffnr_in = (double *)malloc(8 * D * T);
    float *buffer2 = (float *)malloc(4 * (D > V_SIZE ? D : V_SIZE) * T);
    float *buffer3 = (float *)malloc(4 * D * T), *buffer4 = (float *)malloc(4 * D * T);
    float *ffnrbuffer = (float *)malloc(4 * 4 * D * T);
    float *mean = (float *)malloc(4 * T), *var = (float *)malloc(4 * T);
    if (!x || !buffer1 || !buffer2 || !buffer3 || !buffer4 || !ffnrbuffer || !ffnk_in || !ffnr_in) return -1;

    /* rwkv.cu:513-524: embedding row (f32) -> f64, ln0 -> x */
    for (uint64_t t = 0; t < T; t++) {
        if (tokens[t] >= V_SIZE) return -2;
        for (uint64_t i = 0; i < D; i++) buffer1[i + t * D] = (double)embed[tokens[t] * D + i];
    }
    oracle_meanvar(D, buffer1, T, mean, var);
    oracle_layernorm(D, buffer1, ln, 0, mean, var, x, T);

    for (uint64_t l = 0; l < L; l++) {
        /* time-mix, rwkv.cu:535-553 */
        oracle_meanvar(D, x, T, mean, var);
        oracle_layernorm(D, x, ln, 4 * l + 2, mean, var, buffer1, T);
        oracle_mixatt(D, buffer1, sxy, (const double *)m->t[MIXK], (const double *)m->t[MIXV],
                      (const double *)m->t[MIXR], ffnrbuffer, l, L, T, mode);
        memset(buffer2, 0, 4 * D * T); memset(buffer3, 0, 4 * D * T); memset(buffer4, 0, 4 * D * T);
        oracle_mm8_three(D, ffnrbuffer, (const uint8_t *)m->t[KM], (const uint8_t *)m->t[VM], (const uint8_t *)m->t[RM],
                         (const float *)m->t[KR], (const float *)m->t[VR], (const float *)m->t[RR],
                         (const float *)m->t[O1], (const float *)m->t[O2], (const float *)m->t[O3],
                         buffer2, buffer3, buffer4, l, T);
        oracle_wkv(D, (const double *)m->t[DECAY], (const double *)m->t[BONUS], buffer2, buffer3, buffer4,
                   buffer1, saa, sbb, spp, l, L, T, mode);
        for (uint64_t i = 0; i < D * T; i++) buffer2[i] = (float)x[i];            /* :548 residual via accumulator */
        oracle_mm8_one_f64(D, D, buffer1, (const uint8_t *)m->t[ATTOUT], buffer2,
                           (const float *)m->t[ATTOUTR], (const float *)m->t[ATTOUTO], l, T);
        for (uint64_t i = 0; i < D * T; i++) x[i] = (double)buffer2[i];            /* :553 */

        /* channel-mix, rwkv.cu:557-577 */
        oracle_meanvar(D, x, T, mean, var);
        oracle_layernorm(D, x, ln, 4 * (l + 1), mean, var, buffer1, T);
        oracle_mixffn(D, buffer1, sdd, (const double *)m->t[FFNMIXK], (const double *)m->t[FFNMIXV],
                      ffnk_in, ffnr_in, l, L, T, mode);
        memset(buffer2, 0, 4 * D * T);
        oracle_mm8_one_f64(D, D, ffnr_in, (const uint8_t *)m->t[FFNR], buffer2,
                           (const float *)m->t[FFNRR], (const float *)m->t[FFNRO], l, T);
        for (uint64_t i = 0; i < D * T; i++)                                         /* sigmoid :212, exp in f64 */
            buffer4[i] = (float)(1.0 / (1.0 + exp(-(double)buffer2[i])));
        memset(ffnrbuffer, 0, 4 * 4 * D * T);
        oracle_mm8_one_f64(D, 4 * D, ffnk_in, (const uint8_t *)m->t[FFNK], ffnrbuffer,
                           (const float *)m->t[FFNKR], (const float *)m->t[FFNKO], l, T);
        for (uint64_t i = 0; i < 4 * D * T; i++) {                                   /* relu^2 :189-190 */
            float a = ffnrbuffer[i];
            a = a * (float)(a > 0);
            ffnrbuffer[i] = a * a;
        }
        memset(buffer3, 0, 4 * D * T);
        oracle_mm8_one_f32(4 * D, D, ffnrbuffer, (const uint8_t *)m->t[FFNV], buffer3,
                           (const float *)m->t[FFNVR], (const float *)m->t[FFNVO], l, T);
        for (uint64_t i = 0; i < D * T; i++) x[i] = x[i] + (double)(buffer3[i] * buffer4[i]); /* blockout :407 */
    }

    /* rwkv.cu:585-589 */
    oracle_meanvar(D, x, T, mean, var);
    oracle_layernorm(D, x, ln, 4 * L + 2, mean, var, buffer1, T);
    memset(buffer2, 0, 4 * V_SIZE * T);
    oracle_mm8_one_f64(D, V_SIZE, buffer1, (const uint8_t *)m->t[HEAD], buffer2,
                       (const float *)m->t[HEADR], (const float *)m->t[HEADO], 0, T);
    memcpy(logits, buffer2, 4 * V_SIZE * T);

    free(x); free(buffer1); free(buffer2); free(buffer3); free(buffer4); free(ffnrbuffer);
    free(ffnk_in); free(ffnr_in); free(mean); free(var);
    return 0;
}

/* ---- quantize_matrix: converter/convert_model.py:108-119.
 * xx is the torch weight [out][in] as float32 (row-major).  Produces u8 [in][out],
 * r[in] f32, o[in] f32.  Arithmetic follows torch: x-mini etc. in double because
 * mini was promoted by .double() in the reference. ---- */
void oracle_quantize_matrix(const float *xx, uint64_t n_out, uint64_t n_in, uint8_t *q, float *r, float *o)
{
    for (uint64_t j = 0; j < n_in; j++) {
        float mn = xx[j];
        for (uint64_t k = 1; k < n_out; k++) if (xx[k * n_in + j] < mn) mn = xx[k * n_in + j];
        const double mini = (double)mn;
        double mx = 0.0;
        for (uint64_t k = 0; k < n_out; k++) { double v = (double)xx[k * n_in + j] - mini; if (v > mx) mx = v; }
        const double ran = mx / 255.0;
        double fsum = 0.0;
        for (uint64_t k = 0; k < n_out; k++) {
            const double v = ((double)xx[k * n_in + j] - mini) / ran;
            const double tr = trunc(v);
            fsum += v - tr;
            q[j * n_out + k] = (uint8_t)tr;
        }
        r[j] = (float)ran;
        o[j] = (float)(mini + (fsum / (double)n_out) * ran);
    }
}

/* greedy pick used by the parity runs: storygen bans token 0 (out[0] = -99,
 * examples/storygen/storygen.cpp:66) and then samples; greedy = argmax. */
uint64_t oracle_argmax_ban0(const float *logits)
{
    uint64_t best = 1;
    for (uint64_t i = 2; i < V_SIZE; i++) if (logits[i] > logits[best]) best = i;
    return best;
}

/* ---- one pipeline stage of one token: layers [l0, l1) of rwkv.cu:528-581 on state slot `slot`
 * (state arrays are [slots][L][D]); embedding + ln0 (rwkv.cu:513-524) when l0 == 0, ln_out + head
 * (rwkv.cu:585-589) when l1 == L.  x is the residual vector handed between stages (f64[D], in/out).
 * Same arithmetic as oracle_forward with T = 1; used to check the layer-pipeline host logic. ---- */
int oracle_stage_forward(const oracle_model *m, uint64_t token, double *x, uint64_t l0, uint64_t l1,
                         double **state, uint64_t slot, float *logits)
{
    const uint64_t L = m->L, D = m->D;
    const double *ln = (const double *)m->t[LAYERNORMS];
    if (l0 >= l1 || l1 > L) return -3;
    const uint64_t so = slot * L * D;
    double *sxy = state[0] + so, *saa = state[1] + so, *sbb = state[2] + so, *spp = state[3] + so, *sdd = state[4] + so;
    double *buffer1 = (double *)malloc(8 * D), *ffnk_in = (double *)malloc(8 * D), *ffnr_in = (double *)malloc(8 * D);
    float *buffer2 = (float *)malloc(4 * (D > V_SIZE ? D : V_SIZE)), *buffer3 = (float *)malloc(4 * D), *buffer4 = (float *)malloc(4 * D);
    float *ffnrbuffer = (float *)malloc(4 * 4 * D), mean, var;
    if (l0 == 0) {
        if (token >= V_SIZE) return -2;
        const float *embed = (const float *)m->t[EMBED];
        for (uint64_t i = 0; i < D; i++) buffer1[i] = (double)embed[token * D + i];
        oracle_meanvar(D, buffer1, 1, &mean, &var);
        oracle_layernorm(D, buffer1, ln, 0, &mean, &var, x, 1);
    }
    for (uint64_t l = l0; l < l1; l++) {
        oracle_meanvar(D, x, 1, &mean, &var);
        oracle_layernorm(D, x, ln, 4 * l + 2, &mean, &var, buffer1, 1);
        oracle_mixatt(D, buffer1, sxy, (const double *)m->t[MIXK], (const double *)m->t[MIXV], (const double *)m->t[MIXR],
                      ffnrbuffer, l, L, 1, MODE_GPT);
        memset(buffer2, 0, 4 * D); memset(buffer3, 0, 4 * D); memset(buffer4, 0, 4 * D);
        oracle_mm8_three(D, ffnrbuffer, (const uint8_t *)m->t[KM], (const uint8_t *)m->t[VM], (const uint8_t *)m->t[RM],
                         (const float *)m->t[KR], (const float *)m->t[VR], (const float *)m->t[RR],
                         (const float *)m->t[O1], (const float *)m->t[O2], (const float *)m->t[O3], buffer2, buffer3, buffer4, l, 1);
        oracle_wkv(D, (const double *)m->t[DECAY], (const double *)m->t[BONUS], buffer2, buffer3, buffer4, buffer1, saa, sbb, spp, l, L, 1, MODE_GPT);
        for (uint64_t i = 0; i < D; i++) buffer2[i] = (float)x[i];
        oracle_mm8_one_f64(D, D, buffer1, (const uint8_t *)m->t[ATTOUT], buffer2, (const float *)m->t[ATTOUTR], (const float *)m->t[ATTOUTO], l, 1);
        for (uint64_t i = 0; i < D; i++) x[i] = (double)buffer2[i];
        oracle_meanvar(D, x, 1, &mean, &var);
        oracle_layernorm(D, x, ln, 4 * (l + 1), &mean, &var, buffer1, 1);
        oracle_mixffn(D, buffer1, sdd, (const double *)m->t[FFNMIXK], (const double *)m->t[FFNMIXV], ffnk_in, ffnr_in, l, L, 1, MODE_GPT);
        memset(buffer2, 0, 4 * D);
        oracle_mm8_one_f64(D, D, ffnr_in, (const uint8_t *)m->t[FFNR], buffer2, (const float *)m->t[FFNRR], (const float *)m->t[FFNRO], l, 1);
        for (uint64_t i = 0; i < D; i++) buffer4[i] = (float)(1.0 / (1.0 + exp(-(double)buffer2[i])));
        memset(ffnrbuffer, 0, 4 * 4 * D);
        oracle_mm8_one_f64(D, 4 * D, ffnk_in, (const uint8_t *)m->t[FFNK], ffnrbuffer, (const float *)m->t[FFNKR], (const float *)m->t[FFNKO], l, 1);
        for (uint64_t i = 0; i < 4 * D; i++) { float a = ffnrbuffer[i]; a = a * (float)(a > 0); ffnrbuffer[i] = a * a; }
        memset(buffer3, 0, 4 * D);
        oracle_mm8_one_f32(4 * D, D, ffnrbuffer, (const uint8_t *)m->t[FFNV], buffer3, (const float *)m->t[FFNVR], (const float *)m->t[FFNVO], l, 1);
        for (uint64_t i = 0; i < D; i++) x[i] = x[i] + (double)(buffer3[i] * buffer4[i]);
    }
    if (l1 == L && logits) {
        oracle_meanvar(D, x, 1, &mean, &var);
        oracle_layernorm(D, x, ln, 4 * L + 2, &mean, &var, buffer1, 1);
        memset(buffer2, 0, 4 * V_SIZE);
        oracle_mm8_one_f64(D, V_SIZE, buffer1, (const uint8_t *)m->t[HEAD], buffer2, (const float *)m->t[HEADR], (const float *)m->t[HEADO], 0, 1);
        memcpy(logits, buffer2, 4 * V_SIZE);
    }
    free(buffer1); free(ffnk_in); free(ffnr_in); free(buffer2); free(buffer3); free(buffer4); free(ffnrbuffer);
    return 0;
}

/* threads the OpenMP loops above run on (bench.py's cpu_baseline reports it next to the host's core count) */
int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* torch.distributed.run exports OMP_NUM_THREADS=1 to its workers: bench.py's N > 1 line sets the count for its CPU leg explicitly */
void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
