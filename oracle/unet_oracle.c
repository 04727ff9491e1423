/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the arithmetic of UnMicst's UNet forward pass, used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / CPU baseline.  Nothing in
 * unmicst_amd/ may call into this file.
 *
 * What it restates (reference = /root/reference, HMS-IDAC/UnMicst):
 *   legacy graph   UnMicst.py:51-187      (conv -> ReLU -> conv(extra) (+) 1x1 shortcut -> ReLU -> BN -> maxpool ...)
 *   v2 graph       UnMicst1-5.py:55-237   (== UnMicst2.py:52-235 at inference: conv (+) ks x ks shortcut -> BN ->
 *                                           LeakyReLU(0.2) -> maxpool ... 1x1 conv -> BN -> softmax)
 * The arithmetic itself lives in TensorFlow (third party, not vendored: Dockerfile:1 pins
 * tensorflow/tensorflow:2.7.1-gpu, conda.yml:4 pins tensorflow=1.15.0).  The op semantics restated here are
 * TensorFlow's published ones:
 *   tf.nn.conv2d(padding='SAME', stride 1)      cross-correlation, filter [kh,kw,Cin,Cout], zero pad (k-1)/2
 *   tf.nn.conv2d_transpose(stride 2, 'SAME')    gradient of the stride-2 SAME conv, filter [kh,kw,Cout,Cin]:
 *                                               out[2i + a - pad_before] += in[i] * W[a], pad_before = (k-2)/2
 *   tf.layers.batch_normalization(training=0)   gamma*(x-mean)/sqrt(var+1e-3)+beta   (epsilon from model.ckpt.meta)
 *   tf.nn.relu / tf.nn.leaky_relu(alpha=0.2) / tf.nn.max_pool 2x2 s2 / tf.concat axis 3 / tf.nn.softmax
 * Pinning: tests/test_oracle_golden.py checks this file + oracle/pi2d_oracle.py against the reference's own
 * known-answer artefacts ("UNet sample data/registration/105.tif" -> prob_maps/105_{Contours,Nuclei}PM_1.tif,
 * legacy graph + models/nucleiDAPI) to <= 1 uint8 LSB.  The v2 graph has no golden output in the reference
 * (weights not shipped): its op primitives are the ones pinned by the legacy golden; LeakyReLU, the ks x ks
 * shortcut and BN-before-activation follow TF documentation only ("parity unpinned" for v2 -- see DESIGN.md).
 *
 * Numerics: tensors are float32 NHWC between ops (as in TF); every dot product is accumulated in double and
 * rounded once to float32, so the oracle is at least as accurate as any fp32 summation order TF may use.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LAYERS 8
#define ORC_MAX_EXTRA 4

typedef struct {
    int graph;       /* 0 = legacy (UnMicst.py), 1 = v2 (UnMicst1-5.py / UnMicst2.py) */
    int imSize, nChannels, nClasses, nOut0, nLayers, ks, nExtraConvs, featMapsFact;
} orc_hparams;

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* tf.nn.conv2d, stride 1, padding SAME.  x [B,H,W,Cin], w [kh,kw,Cin,Cout], y [B,H,W,Cout]. */
void orc_conv2d_same(const float* x, int B, int H, int W, int Cin, const float* w, int kh, int kw, int Cout,
                     float* y) {
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2; /* odd kernels, stride 1: symmetric padding */
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    for (int co = 0; co < Cout; ++co) acc[co] = 0.0;
                    for (int a = 0; a < kh; ++a) {
                        const int iy = oy + a - ph;
                        if (iy < 0 || iy >= H) continue;
                        for (int c = 0; c < kw; ++c) {
                            const int ix = ox + c - pw;
                            if (ix < 0 || ix >= W) continue;
                            const float* xp = x + (((size_t)b * H + iy) * W + ix) * Cin;
                            const float* wp = w + ((size_t)a * kw + c) * Cin * Cout;
                            for (int ci = 0; ci < Cin; ++ci) {
                                const double xv = xp[ci];
                                const float* wr = wp + (size_t)ci * Cout;
                                for (int co = 0; co < Cout; ++co) acc[co] += xv * (double)wr[co];
                            }
                        }
                    }
                    float* yp = y + (((size_t)b * H + oy) * W + ox) * Cout;
                    for (int co = 0; co < Cout; ++co) yp[co] = (float)acc[co];
                }
        free(acc);
    }
}

/* tf.nn.conv2d_transpose, stride 2, padding SAME, output spatial size exactly 2h x 2w.
 * x [B,h,w,Cin], wt [kh,kw,Cout,Cin] (TF puts the OUTPUT channel before the input channel here), y [B,2h,2w,Cout].
 * Definition used: the transpose (gradient w.r.t. input) of the forward conv F: [2h,2w,Cout] -> [h,w,Cin],
 * F[i] = sum_a in[2i + a - pad_before] W[a], pad_total = max((h-1)*2 + k - 2h, 0) = k-2, pad_before = pad_total/2.
 * Written as a gather over output positions so that each output element is one double-accumulated dot product. */
void orc_conv2d_transpose_s2(const float* x, int B, int h, int w, int Cin, const float* wt, int kh, int kw,
                             int Cout, float* y) {
    const int H = 2 * h, W = 2 * w;
    const int pbh = (kh - 2) / 2, pbw = (kw - 2) / 2;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int u = 0; u < H; ++u)
                for (int v = 0; v < W; ++v) {
                    for (int co = 0; co < Cout; ++co) acc[co] = 0.0;
                    for (int a = 0; a < kh; ++a) {
                        const int t = u - a + pbh; /* = 2*i */
                        if (t < 0 || (t & 1)) continue;
                        const int i = t >> 1;
                        if (i >= h) continue;
                        for (int c = 0; c < kw; ++c) {
                            const int s = v - c + pbw;
                            if (s < 0 || (s & 1)) continue;
                            const int j = s >> 1;
                            if (j >= w) continue;
                            const float* xp = x + (((size_t)b * h + i) * w + j) * Cin;
                            const float* wp = wt + ((size_t)a * kw + c) * Cout * Cin;
                            for (int co = 0; co < Cout; ++co) {
                                const float* wr = wp + (size_t)co * Cin;
                                double s2 = 0.0;
                                for (int ci = 0; ci < Cin; ++ci) s2 += (double)xp[ci] * (double)wr[ci];
                                acc[co] += s2;
                            }
                        }
                    }
                    float* yp = y + (((size_t)b * H + u) * W + v) * Cout;
                    for (int co = 0; co < Cout; ++co) yp[co] = (float)acc[co];
                }
        free(acc);
    }
}

/* tf.layers.batch_normalization(training=False): y = gamma*(x-mean)/sqrt(var+eps)+beta, eps = 1e-3. In place. */
void orc_batchnorm(float* x, size_t npix, int C, const float* gamma, const float* beta, const float* mean,
                   const float* var) {
    const double eps = 0.001;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)npix; ++p)
        for (int c = 0; c < C; ++c) {
            const double s = (double)gamma[c] / sqrt((double)var[c] + eps);
            x[(size_t)p * C + c] = (float)(((double)x[(size_t)p * C + c] - (double)mean[c]) * s + (double)beta[c]);
        }
}

void orc_relu(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* tf.nn.leaky_relu default alpha = 0.2 (attr alpha=0.2000000030 in the .meta graphs). */
void orc_leaky_relu(float* x, size_t n) {
    const float alpha = 0.2f;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) x[i] = x[i] > 0.f ? x[i] : alpha * x[i];
}

void orc_add(float* a, const float* b, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) a[i] = a[i] + b[i];
}

/* tf.nn.max_pool 2x2 stride 2 SAME on even sizes (no padding). x [B,H,W,C] -> y [B,H/2,W/2,C]. */
void orc_maxpool2(const float* x, int B, int H, int W, int C, float* y) {
    const int h = H / 2, w = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j)
                for (int c = 0; c < C; ++c) {
                    const float* p = x + (((size_t)b * H + 2 * i) * W + 2 * j) * C + c;
                    float m = p[0];
                    if (p[C] > m) m = p[C];
                    if (p[(size_t)W * C] > m) m = p[(size_t)W * C];
                    if (p[(size_t)W * C + C] > m) m = p[(size_t)W * C + C];
                    y[(((size_t)b * h + i) * w + j) * C + c] = m;
                }
}

/* tf.concat([a, b], 3). */
void orc_concat(const float* a, int Ca, const float* b, int Cb, size_t npix, float* y) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)npix; ++p) {
        memcpy(y + (size_t)p * (Ca + Cb), a + (size_t)p * Ca, sizeof(float) * (size_t)Ca);
        memcpy(y + (size_t)p * (Ca + Cb) + Ca, b + (size_t)p * Cb, sizeof(float) * (size_t)Cb);
    }
}

/* tf.nn.softmax over the last axis (max-subtracted). In place. */
void orc_softmax(float* x, size_t npix, int C) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)npix; ++p) {
        float* v = x + (size_t)p * C;
        double m = v[0];
        for (int c = 1; c < C; ++c)
            if (v[c] > m) m = v[c];
        double e[16];
        double s = 0.0;
        for (int c = 0; c < C; ++c) {
            e[c] = exp((double)v[c] - m);
            s += e[c];
        }
        for (int c = 0; c < C; ++c) v[c] = (float)(e[c] / s);
    }
}

/* ---- canonical weight blob walker (layout documented in include/umx.h, "weight blob") ---- */
typedef struct {
    const float* p;
    size_t left;
    int ok;
} blob_t;

static const float* take(blob_t* b, size_t n) {
    if (b->left < n) {
        b->ok = 0;
        return b->p;
    }
    const float* r = b->p;
    b->p += n;
    b->left -= n;
    return r;
}

static float* fbuf(size_t n) { return (float*)malloc(sizeof(float) * (n ? n : 1)); }

/* Full forward pass.  x [B,P,P,C] (already normalised), out [B,P,P,K] softmax probabilities.
 * Returns 0 on success, 1 on a blob-size mismatch, 2 on unsupported hyper-parameters. */
int orc_forward(const orc_hparams* hp, const float* blob, size_t blob_floats, const float* x, int B, float* out) {
    const int L = hp->nLayers, ks = hp->ks, P = hp->imSize, nx = hp->nExtraConvs;
    if (L < 1 || L > ORC_MAX_LAYERS || nx > ORC_MAX_EXTRA || hp->nClasses > 16 || (P >> L) < 1 || (P & ((1 << L) - 1)))
        return 2;
    int nOut[ORC_MAX_LAYERS + 2];
    nOut[0] = hp->nChannels;
    nOut[1] = hp->nOut0;
    for (int i = 0; i < L; ++i) nOut[i + 2] = nOut[i + 1] * hp->featMapsFact;
    const int v2 = hp->graph == 1;
    const int kss = v2 ? ks : 1; /* shortcut kernel: ks x ks in v2 (UnMicst1-5.py:106-109), 1x1 in legacy (UnMicst.py:95-96) */
    blob_t bl = {blob, blob_floats, 1};

    const float* ds[ORC_MAX_LAYERS + 1]; /* dsX of the reference: dsX[0] = input, dsX[i+1] = pooled output of ld{i} */
    float* owned[ORC_MAX_LAYERS + 1];
    ds[0] = x;
    owned[0] = NULL;
    int S = P;
    for (int i = 0; i < L; ++i) {
        const int Ci = nOut[i], Co = nOut[i + 1];
        const size_t npix = (size_t)B * S * S;
        const float* w1 = take(&bl, (size_t)ks * ks * Ci * Co);
        const float* wx[ORC_MAX_EXTRA];
        for (int e = 0; e < nx; ++e) wx[e] = take(&bl, (size_t)ks * ks * Co * Co);
        const float* wsc = take(&bl, (size_t)kss * kss * Ci * Co);
        const float* g = take(&bl, Co);
        const float* be = take(&bl, Co);
        const float* mu = take(&bl, Co);
        const float* va = take(&bl, Co);
        if (!bl.ok) return 1;
        float* c00 = fbuf(npix * Co);
        float* tmp = fbuf(npix * Co);
        orc_conv2d_same(ds[i], B, S, S, Ci, w1, ks, ks, Co, c00);
        for (int e = 0; e < nx; ++e) { /* c00 = conv(act(c00), extra) */
            if (v2) orc_leaky_relu(c00, npix * Co); else orc_relu(c00, npix * Co);
            orc_conv2d_same(c00, B, S, S, Co, wx[e], ks, ks, Co, tmp);
            float* t = c00; c00 = tmp; tmp = t;
        }
        orc_conv2d_same(ds[i], B, S, S, Ci, wsc, kss, kss, Co, tmp); /* shortcut */
        orc_add(c00, tmp, npix * Co);
        if (v2) { /* leaky_relu(BN(c00+shortcut)) : UnMicst1-5.py:114 */
            orc_batchnorm(c00, npix, Co, g, be, mu, va);
            orc_leaky_relu(c00, npix * Co);
        } else { /* BN(relu(c00+shortcut)) : UnMicst.py:99 */
            orc_relu(c00, npix * Co);
            orc_batchnorm(c00, npix, Co, g, be, mu, va);
        }
        float* pooled = fbuf((size_t)B * (S / 2) * (S / 2) * Co);
        orc_maxpool2(c00, B, S, S, Co, pooled);
        free(c00);
        free(tmp);
        ds[i + 1] = pooled;
        owned[i + 1] = pooled;
        S /= 2;
    }
    /* bottom layer: lb */
    float* cur;
    {
        const int Ci = nOut[L], Co = nOut[L + 1];
        const size_t npix = (size_t)B * S * S;
        const float* w = take(&bl, (size_t)ks * ks * Ci * Co);
        cur = fbuf(npix * Co);
        orc_conv2d_same(ds[L], B, S, S, Ci, w, ks, ks, Co, cur);
        if (v2) { /* leaky_relu(BN(conv)) ; dropout is identity at inference : UnMicst1-5.py:136-139 */
            const float* g = take(&bl, Co);
            const float* be = take(&bl, Co);
            const float* mu = take(&bl, Co);
            const float* va = take(&bl, Co);
            if (!bl.ok) { free(cur); return 1; }
            orc_batchnorm(cur, npix, Co, g, be, mu, va);
            orc_leaky_relu(cur, npix * Co);
        } else {
            if (!bl.ok) { free(cur); return 1; }
            orc_relu(cur, npix * Co); /* UnMicst.py:114 */
        }
    }
    /* up-sampling layers, index = L-1 .. 0 (UnMicst1-5.py:231-232) */
    for (int idx = L - 1; idx >= 0; --idx) {
        const int Cskip = nOut[idx], Cup = nOut[idx + 1], Cin = nOut[idx + 2];
        const int S2 = S * 2;
        const size_t npix2 = (size_t)B * S2 * S2;
        const float* wt = take(&bl, (size_t)ks * ks * Cup * Cin);
        const float* w2 = take(&bl, (size_t)ks * ks * (Cskip + Cup) * Cup);
        const float *g = NULL, *be = NULL, *mu = NULL, *va = NULL;
        if (v2) {
            g = take(&bl, Cup);
            be = take(&bl, Cup);
            mu = take(&bl, Cup);
            va = take(&bl, Cup);
        }
        const float* wx[ORC_MAX_EXTRA];
        for (int e = 0; e < nx; ++e) wx[e] = take(&bl, (size_t)ks * ks * Cup * Cup);
        if (!bl.ok) { free(cur); return 1; }
        float* us = fbuf(npix2 * Cup);
        orc_conv2d_transpose_s2(cur, B, S, S, Cin, wt, ks, ks, Cup, us);
        free(cur);
        if (v2) orc_leaky_relu(us, npix2 * Cup); else orc_relu(us, npix2 * Cup);
        float* cc = fbuf(npix2 * (Cskip + Cup));
        orc_concat(ds[idx], Cskip, us, Cup, npix2, cc); /* concat3([dsX[index], us]) */
        free(us);
        float* cv = fbuf(npix2 * Cup);
        orc_conv2d_same(cc, B, S2, S2, Cskip + Cup, w2, ks, ks, Cup, cv);
        free(cc);
        if (v2) orc_batchnorm(cv, npix2, Cup, g, be, mu, va);
        if (v2) orc_leaky_relu(cv, npix2 * Cup); else orc_relu(cv, npix2 * Cup);
        float* tmp = fbuf(npix2 * Cup);
        for (int e = 0; e < nx; ++e) {
            orc_conv2d_same(cv, B, S2, S2, Cup, wx[e], ks, ks, Cup, tmp);
            if (v2) orc_leaky_relu(tmp, npix2 * Cup); else orc_relu(tmp, npix2 * Cup);
            float* t = cv; cv = tmp; tmp = t;
        }
        free(tmp);
        cur = cv;
        S = S2;
    }
    /* top layer + softmax */
    {
        const int Ci = nOut[1], K = hp->nClasses;
        const size_t npix = (size_t)B * S * S;
        const float* w = take(&bl, (size_t)Ci * K);
        if (!bl.ok) { free(cur); return 1; }
        orc_conv2d_same(cur, B, S, S, Ci, w, 1, 1, K, out);
        free(cur);
        if (v2) {
            const float* g = take(&bl, K);
            const float* be = take(&bl, K);
            const float* mu = take(&bl, K);
            const float* va = take(&bl, K);
            if (!bl.ok) return 1;
            orc_batchnorm(out, npix, K, g, be, mu, va);
        }
        orc_softmax(out, npix, K);
    }
    for (int i = 1; i <= L; ++i) free(owned[i]);
    return bl.left == 0 ? 0 : 1;
}
