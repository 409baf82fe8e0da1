/*
 * oracle/unet_exact.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement of the inference graph of build_UNetXception
 * (reference fl_tissue_model_tools/models.py:110-166) with a FIXED arithmetic order:
 * every contraction is a k-ordered chain of single-rounding f32 fused multiply-adds,
 *     acc = fmaf(a[k], w[k], acc),   k = input-channel blocks of 32, inside a block tap-major (ky, kx), then channel,
 * which is bit-for-bit what v_mfma_f32_32x32x2_f32 computes on gfx950 when the K dimension
 * is walked in the same order.  The HIP path follows the same order, so GPU and oracle
 * outputs are compared bit-exactly.
 *
 * The arithmetic of the real reference lives in tensorflow==2.14.1 (setup.py:73), which is
 * not in /root/reference: layer semantics (TF "SAME" padding, Conv2DTranspose flip, BN
 * inference form, nearest upsampling) follow SURVEY.md Appendix A1 and are cross-checked
 * against an independent PyTorch-CPU restatement (oracle/unet.py: forward_torch).
 * "parity unpinned" against TensorFlow itself.
 *
 * All tensors are NHWC float32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))

static inline float relu_f(float v) { return v > 0.0f ? v : 0.0f; }

/* ---- deterministic expf / sigmoid (same operation sequence as the HIP epilogue) ---- */
static inline float exp_det(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -88.0f) x = -88.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}

float orc_sigmoid(float z) { return 1.0f / (1.0f + exp_det(-z)); }

/*
 * Build a zero-padded (pad px each side) copy of the logical input
 *     X[n][y][x][c] = act( S[n][y >> up][x >> up][c] ),  y < (h << up), x < (w << up)
 * act = relu if relu_in.  Padding holds +0.0f and IS multiplied in the chain (the GPU stages
 * zeros for out-of-image taps and multiplies them too).
 */
static float *make_padded(const float *S, int N, int h, int w, int C, int up, int relu_in, int pad)
{
    int H = h << up, W = w << up;
    size_t Hp = H + 2 * pad, Wp = W + 2 * pad;
    float *P = (float *)calloc((size_t)N * Hp * Wp * C, sizeof(float));
    if (!P) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < H; y++) {
            const float *srow = S + (((size_t)n * h + (y >> up)) * w) * C;
            float *drow = P + (((size_t)n * Hp + y + pad) * Wp + pad) * C;
            for (int x = 0; x < W; x++) {
                const float *s = srow + (size_t)(x >> up) * C;
                float *d = drow + (size_t)x * C;
                if (relu_in)
                    for (int c = 0; c < C; c++) d[c] = relu_f(s[c]);
                else
                    memcpy(d, s, C * sizeof(float));
            }
        }
    return P;
}

/* micro kernel: XB pixels x OB output channels, K = taps*Cin chain in order */
#define XB 4
#define OB 64
#define CB 32
CLONES static void conv_block(const float *const *ip /*[taps][XB] row ptrs*/, int taps, int Cin,
                              const float *Wc, int Cout, int o0, int ob, float acc[XB][OB])
{
    for (int i = 0; i < XB; i++)
        for (int o = 0; o < OB; o++) acc[i][o] = 0.0f;
    /* K order of the chain (shared with csrc/unet_kernels.hip): input channels in blocks of CB = 32; inside a
       block tap-major (ky, kx); inside a (block, tap) the channels of every group of 8 in the order 0,4,1,5,2,6,3,7
       (a lane of the MFMA kernel reads 4 consecutive channels with one ds_read_b128, lanes 0-31 the lower and lanes
       32-63 the upper half of the group, and each v_mfma_f32_32x32x2_f32 consumes one channel of either half).
       Block-major order keeps the 3x3 window of a channel block hot in L2 across its taps. */
    for (int cb = 0; cb < Cin; cb += CB)
    for (int t = 0; t < taps; t++) {
        const float *w = Wc + (size_t)t * Cin * Cout + o0;
        for (int q = 0; q < CB; q++) {
            const int c = cb + (q & ~7) + ((q & 1) << 2) + ((q & 7) >> 1);
            if (c >= Cin) continue;
            const float *wr = w + (size_t)c * Cout;
            if (ob == OB) {
                for (int i = 0; i < XB; i++) {
                    float v = ip[t * XB + i][c];
#pragma omp simd
                    for (int o = 0; o < OB; o++) acc[i][o] = __builtin_fmaf(v, wr[o], acc[i][o]);
                }
            } else {
                for (int i = 0; i < XB; i++) {
                    float v = ip[t * XB + i][c];
                    for (int o = 0; o < ob; o++) acc[i][o] = __builtin_fmaf(v, wr[o], acc[i][o]);
                }
            }
        }
    }
}

/*
 * Generic conv (ksize 1 or 3, stride 1 or 2 for ksize 1) over logical input
 * Up^up(S) with optional relu on load.
 *   acc = chain over (ky,kx,c)
 *   v   = scale ? fmaf(acc, scale[o], shift[o]) : acc + shift[o]
 *   v  += resid ? resid[n][y >> rs][x >> rs][o] : nothing
 *   v   = relu_out ? relu(v) : v
 * Wc layout [ksize*ksize][Cin][Cout].  Output dims: (H/stride, W/stride) with H = h << up.
 */
int orc_conv(const float *S, int N, int h, int w, int Cin, int up, int relu_in, const float *Wc,
             int ksize, int stride, int Cout, const float *scale, const float *shift,
             const float *resid, int rs, int relu_out, float *out)
{
    int pad = ksize == 3 ? 1 : 0;
    int taps = ksize * ksize;
    int H = h << up, W = w << up;
    int Ho = H / stride, Wo = W / stride;
    if (ksize == 3 && stride != 1) return -1;
    float *P = make_padded(S, N, h, w, Cin, up, relu_in, pad);
    if (!P) return -2;
    size_t Hp = H + 2 * pad, Wp = W + 2 * pad;
    static const float zeros[2048] = {0};
    if (Cin > 2048) { free(P); return -3; }
    int rh = h >> 0; (void)rh;
    int rH = Ho >> rs, rW = Wo >> rs;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < Ho; y++) {
            float acc[XB][OB];
            const float *ip[9 * XB];
            for (int x0 = 0; x0 < Wo; x0 += XB) {
                int xb = Wo - x0 < XB ? Wo - x0 : XB;
                for (int t = 0; t < taps; t++) {
                    int a = t / ksize, b = t % ksize;
                    for (int i = 0; i < XB; i++) {
                        if (i < xb)
                            ip[t * XB + i] =
                                P + (((size_t)n * Hp + (size_t)y * stride + a) * Wp + (size_t)(x0 + i) * stride + b) * Cin;
                        else
                            ip[t * XB + i] = zeros;
                    }
                }
                for (int o0 = 0; o0 < Cout; o0 += OB) {
                    int ob = Cout - o0 < OB ? Cout - o0 : OB;
                    conv_block(ip, taps, Cin, Wc, Cout, o0, ob, acc);
                    for (int i = 0; i < xb; i++) {
                        int x = x0 + i;
                        float *op = out + (((size_t)n * Ho + y) * Wo + x) * Cout + o0;
                        const float *rp =
                            resid ? resid + (((size_t)n * rH + (y >> rs)) * rW + (x >> rs)) * Cout + o0 : NULL;
                        for (int o = 0; o < ob; o++) {
                            float v = scale ? __builtin_fmaf(acc[i][o], scale[o0 + o], shift[o0 + o])
                                            : acc[i][o] + shift[o0 + o];
                            if (rp) v = v + rp[o];
                            if (relu_out) v = relu_f(v);
                            op[o] = v;
                        }
                    }
                }
            }
        }
    free(P);
    return 0;
}

/*
 * 3x3 convolution over the 2x nearest-upsampled stored tensor, in sub-pixel form (first transposed convolution
 * of up blocks 2.. : models.py:150-152 applied to the UpSampling2D output of models.py:158-163).  Output pixel
 * (2i + py, 2j + px) only sees the stored pixels (i + py - 1 + a, j + px - 1 + b), a, b in {0, 1}; the 3x3 taps that
 * land on the same stored pixel are pre-summed (unet.py:subpixel_weights, f32 adds in (ky, kx) order), which is what
 * csrc/unet_kernels.hip (KS == 2) multiplies.  Wsub layout [4 classes = py * 2 + px][4 taps = a * 2 + b][Cin][Cout];
 * the chain per class follows conv_block's K order with 4 taps.  Same epilogue as orc_conv (no residual).
 */
int orc_conv_subpixel(const float *S, int N, int h, int w, int Cin, int relu_in, const float *Wsub, int Cout,
                      const float *scale, const float *shift, int relu_out, float *out)
{
    float *P = make_padded(S, N, h, w, Cin, 0, relu_in, 1);
    if (!P) return -2;
    size_t Hp = h + 2, Wp = w + 2;
    static const float zeros[2048] = {0};
    if (Cin > 2048) { free(P); return -3; }
    const int Ho = 2 * h, Wo = 2 * w;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < Ho; y++) {
            float acc[XB][OB];
            const float *ip[4 * XB];
            const int i0 = y >> 1, py = y & 1;
            for (int px = 0; px < 2; px++) {
                const float *Wc = Wsub + (size_t)(py * 2 + px) * 4 * Cin * Cout;
                for (int j0 = 0; j0 < w; j0 += XB) {
                    int xb = w - j0 < XB ? w - j0 : XB;
                    for (int t = 0; t < 4; t++)
                        for (int i = 0; i < XB; i++)
                            ip[t * XB + i] = i < xb ? P + (((size_t)n * Hp + i0 + py + (t >> 1)) * Wp + j0 + i + px + (t & 1)) * Cin
                                                    : zeros;
                    for (int o0 = 0; o0 < Cout; o0 += OB) {
                        int ob = Cout - o0 < OB ? Cout - o0 : OB;
                        conv_block(ip, 4, Cin, Wc, Cout, o0, ob, acc);
                        for (int i = 0; i < xb; i++) {
                            float *op = out + (((size_t)n * Ho + y) * Wo + 2 * (j0 + i) + px) * Cout + o0;
                            for (int o = 0; o < ob; o++) {
                                float v = scale ? __builtin_fmaf(acc[i][o], scale[o0 + o], shift[o0 + o]) : acc[i][o] + shift[o0 + o];
                                if (relu_out) v = relu_f(v);
                                op[o] = v;
                            }
                        }
                    }
                }
            }
        }
    free(P);
    return 0;
}

/* depthwise 3x3, zero pad 1, relu on load optional; Wd layout [9][C]; chain over taps in order */
CLONES int orc_dwconv(const float *S, int N, int H, int W, int C, int relu_in, const float *Wd, float *out)
{
    float *P = make_padded(S, N, H, W, C, 0, relu_in, 1);
    if (!P) return -2;
    size_t Hp = H + 2, Wp = W + 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                float *op = out + (((size_t)n * H + y) * W + x) * C;
                for (int c = 0; c < C; c++) op[c] = 0.0f;
                for (int t = 0; t < 9; t++) {
                    const float *ip = P + (((size_t)n * Hp + y + t / 3) * Wp + x + t % 3) * C;
                    const float *wr = Wd + (size_t)t * C;
#pragma omp simd
                    for (int c = 0; c < C; c++) op[c] = __builtin_fmaf(ip[c], wr[c], op[c]);
                }
            }
    free(P);
    return 0;
}

/*
 * stem: Conv2D(64, 3, strides=2, padding="same") on (N,H,W) single-channel input (models.py:119),
 * TF SAME for even H: pad 0 before / 1 after.  Ws layout [9][Cout].  BN + ReLU folded.
 */
int orc_stem(const float *X, int N, int H, int W, const float *Ws, int Cout, const float *scale,
             const float *shift, float *out)
{
    int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < Ho; y++)
            for (int x = 0; x < Wo; x++) {
                float v[9];
                for (int t = 0; t < 9; t++) {
                    int iy = 2 * y + t / 3, ix = 2 * x + t % 3;
                    v[t] = (iy < H && ix < W) ? X[((size_t)n * H + iy) * W + ix] : 0.0f;
                }
                float *op = out + (((size_t)n * Ho + y) * Wo + x) * Cout;
                for (int o = 0; o < Cout; o++) {
                    float acc = 0.0f;
                    for (int t = 0; t < 9; t++) acc = __builtin_fmaf(v[t], Ws[t * Cout + o], acc);
                    op[o] = relu_f(__builtin_fmaf(acc, scale[o], shift[o]));
                }
            }
    return 0;
}

/* MaxPooling2D(3, strides=2, padding="same") (pad 0 before / 1 after for even dims, -inf) + residual add
 * (models.py:138-144): out = maxpool(P2) + R */
int orc_maxpool_add(const float *P2, int N, int H, int W, int C, const float *R, float *out)
{
    int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < Ho; y++)
            for (int x = 0; x < Wo; x++) {
                float *op = out + (((size_t)n * Ho + y) * Wo + x) * C;
                const float *rp = R + (((size_t)n * Ho + y) * Wo + x) * C;
                for (int c = 0; c < C; c++) {
                    float m = -INFINITY;
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) {
                            int iy = 2 * y + a, ix = 2 * x + b;
                            if (iy < H && ix < W) {
                                float v = P2[(((size_t)n * H + iy) * W + ix) * C + c];
                                m = v > m ? v : m;
                            }
                        }
                    op[c] = m + rp[c];
                }
            }
    return 0;
}

/* final Conv2D(1, 3, activation="sigmoid", padding="same") on Up2(S) (models.py:158,166), in the sub-pixel form of
 * orc_conv_subpixel: output (2i + py, 2j + px) reads the stored pixels (i + py - 1 + a, j + px - 1 + b) with the 3x3 taps
 * pre-summed per parity class.  Wsub layout [4 classes][4 slots][C] (unet.py:subpixel_weights of the (3,3,C,1) kernel).
 * Chain order (shared with csrc/unet_kernels.hip:final_kernel): 4-channel group ascending, inside a group slot-major,
 * then channel ascending; out-of-image stored pixels contribute fmaf(0, w, acc).  out (N, 2h, 2w) */
int orc_final(const float *S, int N, int h, int w, int C, const float *Wsub, float bias, float *out)
{
    int H = 2 * h, W = 2 * w;
    static const float zeros[2048] = {0};
    if (C > 2048 || C % 4) return -3;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int i = y >> 1, py = y & 1, j = x >> 1, px = x & 1;
                const float *Wc = Wsub + (size_t)(py * 2 + px) * 4 * C;
                const float *ip[4];
                for (int t = 0; t < 4; t++) {
                    int iy = i + py - 1 + (t >> 1), ix = j + px - 1 + (t & 1);
                    ip[t] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? S + (((size_t)n * h + iy) * w + ix) * C : zeros;
                }
                float acc = 0.0f;
                for (int cg = 0; cg < C; cg += 4)
                    for (int t = 0; t < 4; t++)
                        for (int c = cg; c < cg + 4; c++) acc = __builtin_fmaf(ip[t][c], Wc[(size_t)t * C + c], acc);
                out[((size_t)n * H + y) * W + x] = orc_sigmoid(acc + bias);
            }
    return 0;
}
