/* cv_resize_port.c -- second, independent restatement of OpenCV's 8-bit INTER_LINEAR resize, written in the shape of
 * OpenCV's own C++ (modules/imgproc/src/resize.cpp, 4.x): cv::resize -> resizeGeneric_ with
 *   HResizeLinear<uchar, int, short, INTER_RESIZE_COEF_SCALE>     (one row -> int32 buffer, weights alpha[dx*2+k])
 *   VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS*2>>
 * and the coefficient tables of cv::resize's INTER_LINEAR branch (xofs / ialpha / yofs / ibeta, xmin / xmax).
 * It keeps that code's structure -- per destination row two buffered source rows run through the horizontal pass,
 * then the vertical pass with its own special form ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 --
 * instead of oracle/cv_ref.py's whole-image array expressions, so that agreement of the two (tests/
 * test_oracle_cv_crosscheck.py, bit for bit) is agreement of two separately written programs.  TEST INFRASTRUCTURE:
 * compiled by the test with gcc, never shipped.  Reference call sites: smartVidCrop.py:333-335, :1080, :1158. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define INTER_RESIZE_COEF_BITS 11
#define INTER_RESIZE_COEF_SCALE (1 << INTER_RESIZE_COEF_BITS)

static int cv_floor(float v) { int i = (int)v; return i - (i > v); }
static int cv_round_f(float v) { return (int)lrintf(v); }                 /* round half to even (default FP mode) */
static short saturate_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static int clip_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* HResizeLinear: src rows (count of them) -> dst int rows */
static void hresize_linear(const uint8_t **src, int **dst, int count, const int *xofs, const short *alpha,
                           int swidth, int dwidth, int cn, int xmin, int xmax) {
    (void)swidth; (void)xmin;
    for (int k = 0; k < count; ++k) {
        const uint8_t *S = src[k];
        int *D = dst[k];
        int dx = 0;
        for (; dx < xmax; ++dx) {
            const int sx = xofs[dx];
            D[dx] = S[sx] * alpha[dx * 2] + S[sx + cn] * alpha[dx * 2 + 1];
        }
        for (; dx < dwidth; ++dx) D[dx] = (int)S[xofs[dx]] * INTER_RESIZE_COEF_SCALE;
    }
}

/* VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>> (the uchar specialisation) */
static void vresize_linear(const int **src, uint8_t *dst, const short *beta, int width) {
    const int b0 = beta[0], b1 = beta[1];
    const int *S0 = src[0], *S1 = src[1];
    for (int x = 0; x < width; ++x) {
        const int v = (((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2;
        dst[x] = (uint8_t)clip_i(v, 0, 255);
    }
}

/* cv::resize(src, dst, dsize=(dw, dh), fx, fy, INTER_LINEAR) for CV_8UC(cn).  scale_x / scale_y <= 0: from the sizes
 * (inv_scale = dsize / ssize, as when dsize is given); else the given source-per-destination scale (1 / fx). */
int cv_resize_linear_u8(const uint8_t *src, int sh, int sw, int cn, uint8_t *dst, int dh, int dw, double scale_x, double scale_y) {
    if (scale_x <= 0) scale_x = 1.0 / ((double)dw / sw);
    if (scale_y <= 0) scale_y = 1.0 / ((double)dh / sh);
    const int ksize = 2, ksize2 = ksize / 2;
    const int width = dw * cn;
    int *xofs = (int *)malloc(sizeof(int) * width), *yofs = (int *)malloc(sizeof(int) * dh);
    short *ialpha = (short *)malloc(sizeof(short) * width * ksize), *ibeta = (short *)malloc(sizeof(short) * dh * ksize);
    int *rows[2] = {(int *)malloc(sizeof(int) * width), (int *)malloc(sizeof(int) * width)};
    if (!xofs || !yofs || !ialpha || !ibeta || !rows[0] || !rows[1]) return -1;
    int xmin = 0, xmax = dw;
    float cbuf[2];
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < ksize2 - 1) { xmin = dx + 1; if (sx < 0) { fx = 0; sx = 0; } }
        if (sx + ksize2 >= sw) { xmax = xmax < dx ? xmax : dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        for (int k = 0; k < cn; ++k) xofs[dx * cn + k] = sx * cn + k;
        cbuf[0] = 1.f - fx; cbuf[1] = fx;
        for (int k = 0; k < ksize; ++k) ialpha[dx * cn * ksize + k] = saturate_short(cv_round_f(cbuf[k] * INTER_RESIZE_COEF_SCALE));
        for (int k = ksize; k < cn * ksize; ++k) ialpha[dx * cn * ksize + k] = ialpha[dx * cn * ksize + k - ksize];
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        cbuf[0] = 1.f - fy; cbuf[1] = fy;
        for (int k = 0; k < ksize; ++k) ibeta[dy * ksize + k] = saturate_short(cv_round_f(cbuf[k] * INTER_RESIZE_COEF_SCALE));
    }
    xmin *= cn; xmax *= cn;
    /* resizeGeneric_Invoker: per destination row, the ksize source rows (clipped) through the horizontal pass.
     * (OpenCV re-uses rows already in the buffer; recomputing them gives the same values.) */
    for (int dy = 0; dy < dh; ++dy) {
        const uint8_t *srows[2];
        const int sy0 = yofs[dy];
        for (int k = 0; k < ksize; ++k) srows[k] = src + (size_t)clip_i(sy0 - ksize2 + 1 + k, 0, sh - 1) * sw * cn;
        hresize_linear(srows, rows, ksize, xofs, ialpha, sw * cn, width, cn, xmin, xmax);
        vresize_linear((const int **)rows, dst + (size_t)dy * width, ibeta + dy * ksize, width);
    }
    free(xofs); free(yofs); free(ialpha); free(ibeta); free(rows[0]); free(rows[1]);
    return 0;
}
