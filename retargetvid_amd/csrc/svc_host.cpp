// svc_host.cpp -- the HOST stages between the per-frame centres and the crop windows, as native code.
//
// BASELINE.json's north_star keeps these stages on the host ("LOESS smoothing stays on host"); they are native so that a
// multi-video job (retargetvid_amd/scheduler.py) can run them on a thread pool off the interpreter lock: in Python they
// were ~5 ms per video (a pseudo-inverse per new shot length, SciPy call overhead), i.e. ~1 s of interpreter time for the
// 200-video set against ~0.85 s of GPU time.  No GPU, no handle; plain pointers to HOST memory.
//
// Reference (smartVidCrop.py unless noted), function by function:
//   svc_host_fill_empty_centres   sc_handle_empty_centers                    :1221-1300
//   svc_host_interp_segment       interp_handler                             :1528-1548  (scipy.interpolate.interp1d, kinds
//                                 'linear' and 'quadratic' with fill_value='extrapolate')
//   svc_host_lowpass              sc_butter_lowpass_filter                   :1599-1627  (scipy.signal.filtfilt, method 'pad',
//                                 odd extension of 3 * taps samples, and the moving-average fall-back for short series)
//   svc_host_loess                loess_handler -> pyloess.Loess.estimate    :1629-1646, 3rd_party_libs/loess/pyloess.py:13-95
//   svc_host_savgol               scipy.signal.savgol_filter(mode='interp')  :1643
//   svc_host_temporal             sc_interpolate + sc_smoothing for one video :1550-1597, :1648-1734
//   svc_host_boxes                sc_compute_bb                              :979-1048
//   svc_host_focus_stability      get_points_on_line, sc_check_for_extra_cuts, the focus hold  :1337-1455, :2425-2473
//
// What is bit-for-bit SciPy's arithmetic (same operations, same order; -ffp-contract=off): linear interpolation, the
// filtfilt chain (odd extension, direct-form-II-transposed recurrence, initial conditions zi * x0) given SciPy's b, a, zi,
// the moving-average fall-back, the box arithmetic.  What is the same mathematical result through a better conditioned
// route (differences ~1e-9 .. 1e-7 px, tests/test_host_native.py): the quadratic spline (banded LU of the same collocation
// system; LAPACK's gbsv in SciPy), LOESS (local regression in coordinates centred on the estimated point; the reference
// forms pinv(X^T W X) in coordinates normalised over the whole shot, whose condition number is ~1e7) and Savitzky-Golay.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/svc.h"

void svc_set_error(const char *fmt, ...);

namespace {

// ---- scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate')._call_linear --------------------------------
void interp_linear(const double *x, const double *y, int n, int n_out, double *out) {
    for (int t = 0; t < n_out; ++t) {
        const double xn = (double)t;
        int idx = (int)(std::lower_bound(x, x + n, xn) - x);         // searchsorted, side='left'
        idx = std::min(std::max(idx, 1), n - 1);
        const int lo = idx - 1, hi = idx;
        const double slope = (y[hi] - y[lo]) / (x[hi] - x[lo]);
        out[t] = slope * (xn - x[lo]) + y[lo];
    }
}

// B-spline basis functions of degree k that are non-zero on knot interval ell, at x (scipy's _deBoor_D, derivative 0)
inline void deboor(const double *t, double x, int k, int ell, double *h) {
    double hh[4];
    h[0] = 1.0;
    for (int j = 1; j <= k; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
        for (int n = 1; n <= j; ++n) {
            const int ind = ell + n;
            const double xb = t[ind], xa = t[ind - j];
            if (xb == xa) { h[n] = 0.0; continue; }
            const double w = hh[n - 1] / (xb - xa);
            h[n - 1] += w * (xb - x);
            h[n] = w * (x - xa);
        }
    }
}

// knot interval of x: t[l] <= x < t[l+1], clamped to [k, nt-1] (scipy's find_interval with extrapolate=True)
inline int find_interval(const double *t, int k, int nt, double x, int guess) {
    int l = (guess > k && guess < nt) ? guess : k;
    while (x < t[l] && l != k) --l;
    ++l;
    while (x >= t[l] && l != nt) ++l;
    return l - 1;
}

// scipy.interpolate.interp1d(kind='quadratic') = make_interp_spline(x, y, k=2): not-a-knot knots for an even degree (the
// midpoints of the data sites without the first and the last), collocation system solved with partial pivoting, de Boor
// evaluation with polynomial extrapolation.  Two series over the same sites share the factorisation.
int interp_quadratic(const double *x, const double *y1, const double *y2, int n, int n_out, double *o1, double *o2) {
    const int k = 2, nt = n;
    std::vector<double> t(n + 3);
    t[0] = t[1] = t[2] = x[0];
    for (int m = 1; m <= n - 3; ++m) t[2 + m] = (x[m + 1] + x[m]) / 2;
    t[n] = t[n + 1] = t[n + 2] = x[n - 1];
    // band storage: row r holds columns r-2 .. r+4 (two sub-diagonals, two super-diagonals + the fill-in of row swaps)
    const int W = 7;
    std::vector<double> ab((size_t)n * W, 0.0), c1(y1, y1 + n), c2(y2, y2 + n);
    auto at = [&](int r, int c) -> double & { return ab[(size_t)r * W + (c - r + 2)]; };
    int left = k;
    for (int i = 0; i < n; ++i) {
        left = find_interval(t.data(), k, nt, x[i], left);
        double h[3];
        deboor(t.data(), x[i], k, left, h);
        for (int a = 0; a <= k; ++a) {
            const int col = left - k + a;
            if (col < i - 2 || col > i + 2) { if (h[a] != 0.0) return -1; continue; }
            at(i, col) = h[a];
        }
    }
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(at(c, c));
        for (int r = c + 1; r <= std::min(n - 1, c + 2); ++r)
            if (fabs(at(r, c)) > best) { best = fabs(at(r, c)); p = r; }
        if (best == 0.0) return -1;
        const int cend = std::min(n - 1, c + 4);
        if (p != c) {
            for (int col = c; col <= cend; ++col) std::swap(at(c, col), at(p, col));
            std::swap(c1[c], c1[p]);
            std::swap(c2[c], c2[p]);
        }
        const double piv = at(c, c);
        for (int r = c + 1; r <= std::min(n - 1, c + 2); ++r) {
            const double m = at(r, c) / piv;
            if (m == 0.0) continue;
            for (int col = c + 1; col <= cend; ++col) at(r, col) -= m * at(c, col);
            c1[r] -= m * c1[c];
            c2[r] -= m * c2[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s1 = c1[r], s2 = c2[r];
        for (int col = r + 1; col <= std::min(n - 1, r + 4); ++col) { s1 -= at(r, col) * c1[col]; s2 -= at(r, col) * c2[col]; }
        c1[r] = s1 / at(r, r);
        c2[r] = s2 / at(r, r);
    }
    int l = k;
    for (int i = 0; i < n_out; ++i) {
        const double xn = (double)i;
        l = find_interval(t.data(), k, nt, xn, l);
        double h[3];
        deboor(t.data(), xn, k, l, h);
        double s1 = 0.0, s2 = 0.0;
        for (int a = 0; a <= k; ++a) { s1 += c1[l + a - k] * h[a]; s2 += c2[l + a - k] * h[a]; }
        o1[i] = s1;
        o2[i] = s2;
    }
    return 0;
}

// interp_handler (:1528-1548) for the two series of one shot
int interp_segment(const double *sx, const double *d1, const double *d2, int n, int n_out, double *o1, double *o2) {
    if (n < 1) return -1;
    if (n < 3) {
        for (int i = 0; i < n_out; ++i) { o1[i] = d1[0]; o2[i] = d2[0]; }
        return 0;
    }
    for (int i = 1; i < n; ++i)
        if (!(sx[i] > sx[i - 1])) return -1;                       // interp1d sorts; the selected frame numbers are increasing anyway
    if (n <= 6) {
        interp_linear(sx, d1, n, n_out, o1);
        interp_linear(sx, d2, n, n_out, o2);
        return 0;
    }
    return interp_quadratic(sx, d1, d2, n, n_out, o1, o2);
}

// scipy.signal.lfilter for float64 (the direct form II transposed loop of _lfilter.c), z = initial and final state
void lfilter(const double *b, const double *a, int taps, const double *x, int n, double *y, double *z) {
    if (taps == 1) {
        for (int i = 0; i < n; ++i) y[i] = x[i] * b[0];
        return;
    }
    for (int i = 0; i < n; ++i) {
        const double xn = x[i];
        const double yn = z[0] + b[0] * xn;
        for (int m = 0; m < taps - 2; ++m) z[m] = z[m + 1] + xn * b[m + 1] - yn * a[m + 1];
        z[taps - 2] = xn * b[taps - 1] - yn * a[taps - 1];
        y[i] = yn;
    }
}

// scipy.signal.filtfilt(b, a, x) (method='pad', padtype='odd', padlen=3*taps); false when x is too short (it raises there)
bool filtfilt(const double *b_in, const double *a_in, const double *zi, int taps, const double *x, int n, double *out) {
    const int edge = 3 * taps;
    if (n <= edge) return false;
    std::vector<double> b(b_in, b_in + taps), a(a_in, a_in + taps), z(std::max(taps - 1, 1));
    const double a0 = a[0];
    for (int i = 0; i < taps; ++i) { b[i] /= a0; a[i] /= a0; }
    const int m = n + 2 * edge;
    std::vector<double> ext(m), y(m), r(m);
    const double l2 = 2 * x[0], r2 = 2 * x[n - 1];
    for (int i = 0; i < edge; ++i) ext[i] = l2 - x[edge - i];
    memcpy(&ext[edge], x, sizeof(double) * n);
    for (int i = 0; i < edge; ++i) ext[edge + n + i] = r2 - x[n - 2 - i];
    for (int i = 0; i < taps - 1; ++i) z[i] = zi[i] * ext[0];
    lfilter(b.data(), a.data(), taps, ext.data(), m, y.data(), z.data());
    const double y0 = y[m - 1];
    for (int i = 0; i < m; ++i) r[i] = y[m - 1 - i];
    for (int i = 0; i < taps - 1; ++i) z[i] = zi[i] * y0;
    lfilter(b.data(), a.data(), taps, r.data(), m, y.data(), z.data());
    for (int i = 0; i < n; ++i) out[i] = y[m - 1 - (edge + i)];
    return true;
}

// sc_butter_lowpass_filter (:1599-1627): filtfilt, or -- series too short for its padding -- the 5-point moving average
// over the interior (np.convolve(x, ones(5), 'same') / 5 written back to x[2:n-2])
void lowpass(const double *b, const double *a, const double *zi, int taps, const double *x, int n, double *out) {
    if (filtfilt(b, a, zi, taps, x, n, out)) return;
    memcpy(out, x, sizeof(double) * n);
    for (int i = 2; i < n - 2; ++i) out[i] = ((((x[i - 2] + x[i - 1]) + x[i]) + x[i + 1]) + x[i + 2]) / 5;
}

// Solves the (d+1) x (d+1) system M beta = v in place (partial pivoting); false when singular.
bool solve_small(double *M, double *v, int q) {
    for (int c = 0; c < q; ++c) {
        int p = c;
        for (int r = c + 1; r < q; ++r)
            if (fabs(M[r * q + c]) > fabs(M[p * q + c])) p = r;
        if (!(fabs(M[p * q + c]) > 1e-300)) return false;
        if (p != c) {
            for (int j = 0; j < q; ++j) std::swap(M[c * q + j], M[p * q + j]);
            std::swap(v[c], v[p]);
        }
        for (int r = c + 1; r < q; ++r) {
            const double m = M[r * q + c] / M[c * q + c];
            for (int j = c; j < q; ++j) M[r * q + j] -= m * M[c * q + j];
            v[r] -= m * v[c];
        }
    }
    for (int r = q - 1; r >= 0; --r) {
        double s = v[r];
        for (int j = r + 1; j < q; ++j) s -= M[r * q + j] * v[j];
        v[r] = s / M[r * q + r];
    }
    return true;
}

const int MAX_DEG = 5;

// pyloess.Loess(arange(n), y).estimate(j, window, degree=degree) for every j (pyloess.py:13-95): x and y normalised to
// [0, 1] over the series, the `window` nearest points (get_min_range: the nearer outer neighbour joins next, the right one
// on a tie, clamped at the ends), tricube weights of distance / largest distance, weighted polynomial regression,
// evaluated at x_j, de-normalised.  The regression runs in u = (x_i - x_j) / largest distance (a polynomial fit does not
// depend on an affine change of abscissa; the estimate is the constant coefficient).  A constant series (or a degenerate
// window) gives NaN like the reference's 0/0, which loess_handler turns into "keep the input".
void loess(const double *y, int n, int window, int degree, double *out) {
    const double nan = NAN;
    double ymin = y[0], ymax = y[0];
    bool bad = false;
    for (int i = 0; i < n; ++i) {
        if (!(y[i] == y[i])) bad = true;
        ymin = std::min(ymin, y[i]);
        ymax = std::max(ymax, y[i]);
    }
    if (bad || !(ymax > ymin) || n < 2 || window < 2 || window > n || degree < 0 || degree > MAX_DEG) {
        for (int i = 0; i < n; ++i) out[i] = nan;
        return;
    }
    std::vector<double> nx(n), ny(n);
    for (int i = 0; i < n; ++i) {
        nx[i] = (double)i / (double)(n - 1);
        ny[i] = (y[i] - ymin) / (ymax - ymin);
    }
    const int h = (window - 1) / 2, q = degree + 1;
    for (int j = 0; j < n; ++j) {
        int lo = std::min(std::max(j - h, 0), n - (2 * h + 1));
        if (window % 2 == 0) {                                    // one more point, on the nearer side (pyloess.py:27-48)
            const int hi = lo + 2 * h;
            const double dl = fabs(nx[std::max(lo - 1, 0)] - nx[j]), dr = fabs(nx[std::min(hi + 1, n - 1)] - nx[j]);
            bool left = (hi == n - 1) || (lo > 0 && dl < dr);
            left = left && (j != 0);
            left = left || (j == n - 1);
            if (left) --lo;
        }
        double dmax = 0.0;
        for (int i = 0; i < window; ++i) dmax = std::max(dmax, fabs(nx[lo + i] - nx[j]));
        double M[(MAX_DEG + 1) * (MAX_DEG + 1)] = {0}, v[MAX_DEG + 1] = {0}, mom[2 * MAX_DEG + 1] = {0};
        for (int i = 0; i < window; ++i) {
            const double d = nx[lo + i] - nx[j];
            const double r = fabs(d) / dmax;
            const double c = 1.0 - r * r * r;
            const double w = (r >= -1 && r <= 1) ? c * c * c : 0.0;
            const double u = d / dmax;
            double pw = w;
            for (int a = 0; a <= 2 * degree; ++a) {
                mom[a] += pw;
                if (a <= degree) v[a] += pw * ny[lo + i];
                pw *= u;
            }
        }
        for (int a = 0; a < q; ++a)
            for (int b = 0; b < q; ++b) M[a * q + b] = mom[a + b];
        double est;
        if (solve_small(M, v, q)) est = v[0];
        else est = nan;
        out[j] = est * (ymax - ymin) + ymin;
    }
}

// scipy.signal.savgol_filter(y, window, degree) with mode='interp': interior = least-squares polynomial of the centred
// window evaluated at its middle; the first and last window // 2 samples = the polynomial of the first / last window
// evaluated at their positions.  Rows of the hat matrix come from the normal equations in u = offset / halflen.
int savgol(const double *y, int n, int window, int degree, double *out) {
    if (window > n || window < 1 || window % 2 == 0 || degree >= window || degree < 0 || degree > MAX_DEG) return -1;
    const int h = window / 2, q = degree + 1;
    const double sc = h > 0 ? (double)h : 1.0;
    double mom[2 * MAX_DEG + 1] = {0};
    for (int i = -h; i <= h; ++i) {
        double pw = 1.0;
        const double u = i / sc;
        for (int a = 0; a <= 2 * degree; ++a) { mom[a] += pw; pw *= u; }
    }
    // hat row for evaluation offset e (in samples from the window's middle): c_i = sum_a g_a u_i^a, M g = (e/sc)^a
    auto hat_row = [&](int e, double *c) -> bool {
        double M[(MAX_DEG + 1) * (MAX_DEG + 1)], g[MAX_DEG + 1];
        for (int a = 0; a < q; ++a)
            for (int b = 0; b < q; ++b) M[a * q + b] = mom[a + b];
        double pw = 1.0;
        for (int a = 0; a < q; ++a) { g[a] = pw; pw *= e / sc; }
        if (!solve_small(M, g, q)) return false;
        for (int i = -h; i <= h; ++i) {
            double s = 0.0, p2 = 1.0;
            for (int a = 0; a < q; ++a) { s += g[a] * p2; p2 *= i / sc; }
            c[i + h] = s;
        }
        return true;
    };
    std::vector<double> c(window);
    if (!hat_row(0, c.data())) return -1;
    for (int j = h; j < n - h; ++j) {
        double s = 0.0;
        for (int i = 0; i < window; ++i) s += c[i] * y[j - h + i];
        out[j] = s;
    }
    for (int e = -h; e < 0; ++e) {                               // the first h samples and, mirrored, the last h
        if (!hat_row(e, c.data())) return -1;
        double s = 0.0, s2 = 0.0;
        for (int i = 0; i < window; ++i) {
            s += c[i] * y[i];
            s2 += c[window - 1 - i] * y[n - window + i];          // offset -e from the middle of the last window
        }
        out[h + e] = s;
        out[n - 1 - h - e] = s2;
    }
    return 0;
}

// loess_handler (:1629-1646)
int smooth_handler(const double *d, int n, int loess_filt, int window, int degree, double *out) {
    if (n < 10) { memcpy(out, d, sizeof(double) * n); return 0; }
    if (loess_filt) {
        loess(d, n, window, degree, out);
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += out[i];
        if (!(s == s)) memcpy(out, d, sizeof(double) * n);
        return 0;
    }
    return savgol(d, n, window, degree, out);
}

// numpy.arange(start, stop, step) for float arguments: ceil((stop - start) / step) elements, element i = start + i * step
inline long arange_len(double start, double stop, double step) {
    const double v = ceil((stop - start) / step);
    return v > 0 ? (long)v : 0;
}

// get_points_on_line + the mean of sc_check_for_extra_cuts (:1337-1455): the integer-stepped samples strictly after p1 up to p2
// along the dominant axis, kept where they fall inside the image, and the mean map value over them.  The reference builds
// them in a float32 buffer with a float32 slope; the same conversions in the same order here.  false = "no jump statistic"
// (a move below min_d, a sample count that does not match the buffer -- NumPy raises there --, no sample inside the image).
bool jump_mean(const uint8_t *map, int h, int w, double p1x, double p1y, double p2x, double p2y, double min_d, double *mean) {
    const double dX = p2x - p1x, dY = p2y - p1y, dXa = fabs(dX), dYa = fabs(dY);
    if (dXa < min_d && dYa < min_d) return false;
    const long m = (long)ceil(std::max(dYa, dXa));
    if (m < 1) return false;
    const bool negY = p1y > p2y, negX = p1x > p2x;
    const double sy0 = negY ? p1y - 1 : p1y + 1, sys = negY ? -1.0 : 1.0;
    const double sx0 = negX ? p1x - 1 : p1x + 1, sxs = negX ? -1.0 : 1.0;
    const long ny = arange_len(sy0, negY ? (p1y - dYa) - 1 : (p1y + dYa) + 1, sys);
    const long nx = arange_len(sx0, negX ? (p1x - dXa) - 1 : (p1x + dXa) + 1, sxs);
    // an array assigned to a column of the buffer must have its length (or length 1: broadcast)
    auto fits = [&](long len) { return len == m || len == 1; };
    int mode;                                                // 0: x fixed, 1: y fixed, 2: y leads, 3: x leads
    if (p1x == p2x) { if (!fits(ny)) return false; mode = 0; }
    else if (p1y == p2y) { if (!fits(nx)) return false; mode = 1; }
    else if (dYa > dXa) { if (!fits(ny)) return false; mode = 2; }
    else { if (!fits(nx)) return false; mode = 3; }
    const float slope = mode == 2 ? (float)dX / (float)dY : (mode == 3 ? (float)dY / (float)dX : 0.f);
    const float fp1x = (float)p1x, fp1y = (float)p1y;
    double total = 0.0;
    long cnt = 0;
    for (long i = 0; i < m; ++i) {
        float bx, by;
        if (mode == 0 || mode == 2) {
            by = (float)(sy0 + (double)(ny == 1 ? 0 : i) * sys);
            if (mode == 0) bx = fp1x;
            else bx = (float)((double)(long)(slope * (by - fp1y)) + p1x);
        } else {
            bx = (float)(sx0 + (double)(nx == 1 ? 0 : i) * sxs);
            if (mode == 1) by = fp1y;
            else by = (float)((double)(long)(slope * (bx - fp1x)) + p1y);
        }
        if (bx >= 0.f && by >= 0.f && bx < (float)w && by < (float)h) {
            total += (double)map[(long)floorf(by) * w + (long)floorf(bx)];
            ++cnt;
        }
    }
    if (cnt == 0) return false;
    *mean = total / (double)cnt;
    return true;
}

}   // namespace

extern "C" int svc_host_focus_stability(double *cx, double *cy, int n, const uint8_t *maps_nhw, int h, int w, double fr, int skip,
                                        double min_d_jump, double stab_t, double stab_s, double *jumps, int32_t *inds) {
    if (n < 0 || (n && (!cx || !cy || !maps_nhw || !jumps || !inds)) || h < 1 || w < 1 || !(fr > 0)) {
        svc_set_error("svc_host_focus_stability: invalid argument");
        return SVC_E_INVALID;
    }
    int ni = 0;
    for (int i = 0; i < n; ++i) jumps[i] = 255.0;
    for (int i = 1; i < n; ++i) {
        double mean;
        if (jump_mean(maps_nhw + (size_t)i * h * w, h, w, cx[i - 1], cy[i - 1], cx[i], cy[i], min_d_jump, &mean)) jumps[i] = mean;
        if (jumps[i] < stab_t) inds[ni++] = i;
    }
    // the focus hold (:2448-2473): between two low-saliency jumps close in time the first centre is kept
    for (int k = 0; k + 1 < ni; ++k) {
        const int start = std::max(inds[k] - 1, 0), end = std::min(inds[k + 1] + 1, n - 1);
        if (((double)((end - start) * skip)) / fr <= stab_s)
            for (int j = 0; j < end - start; ++j) { cx[start + j] = cx[start]; cy[start + j] = cy[start]; }
    }
    return ni;
}

extern "C" int svc_host_fill_empty_centres(double *cx, double *cy, int n_sel, const int32_t *seg_sel, int n_seg) {
    if (n_sel < 0 || (n_sel && (!cx || !cy)) || n_seg < 1 || !seg_sel) { svc_set_error("svc_host_fill_empty_centres: invalid argument"); return SVC_E_INVALID; }
    int i = 0;
    while (i < n_sel) {
        if (cx[i] == cx[i]) { ++i; continue; }
        int lo = i, hi = i;
        while (hi + 1 < n_sel && !(cx[hi + 1] == cx[hi + 1])) ++hi;
        int ds = INT32_MAX, de = INT32_MAX;
        for (int s = 0; s < n_seg; ++s) {
            ds = std::min(ds, abs(seg_sel[2 * s] - lo));
            de = std::min(de, abs(seg_sel[2 * s + 1] - hi));
        }
        int src = ds < de ? hi + 1 : lo - 1;
        if (src < 0) src += n_sel;                                // a negative index wraps, as in the reference's list
        if (src >= n_sel) { svc_set_error("svc_host_fill_empty_centres: the run of empty centres %d..%d has no successor to copy from (IndexError in the reference)", lo, hi); return SVC_E_INVALID; }
        for (int j = lo; j <= hi; ++j) { cx[j] = cx[src]; cy[j] = cy[src]; }
        i = hi + 1;
    }
    int left = 0;
    for (int j = 0; j < n_sel; ++j) left += !(cx[j] == cx[j]);
    return left;
}

extern "C" int svc_host_interp_segment(const double *sampled_t, const double *d1, const double *d2, int n, int n_out, double *out1, double *out2) {
    if (n < 1 || n_out < 0 || !sampled_t || !d1 || !d2 || (n_out && (!out1 || !out2))) { svc_set_error("svc_host_interp_segment: invalid argument"); return SVC_E_INVALID; }
    if (interp_segment(sampled_t, d1, d2, n, n_out, out1, out2)) { svc_set_error("svc_host_interp_segment: sample times must be strictly increasing (singular collocation system)"); return SVC_E_INVALID; }
    return SVC_OK;
}

extern "C" int svc_host_lowpass(const double *b, const double *a, const double *zi, int taps, const double *x, int n, double *out) {
    if (taps < 1 || taps > 64 || !b || !a || (taps > 1 && !zi) || n < 0 || (n && (!x || !out)) || a[0] == 0.0) { svc_set_error("svc_host_lowpass: invalid argument"); return SVC_E_INVALID; }
    if (n) lowpass(b, a, zi, taps, x, n, out);
    return SVC_OK;
}

extern "C" int svc_host_loess(const double *y, int n, int window, int degree, double *out) {
    if (n < 0 || (n && (!y || !out))) { svc_set_error("svc_host_loess: invalid argument"); return SVC_E_INVALID; }
    if (n) loess(y, n, window, degree, out);
    return SVC_OK;
}

extern "C" int svc_host_savgol(const double *y, int n, int window, int degree, double *out) {
    if (n < 1 || !y || !out || savgol(y, n, window, degree, out)) { svc_set_error("svc_host_savgol: window must be odd, <= n and > degree"); return SVC_E_INVALID; }
    return SVC_OK;
}

extern "C" int svc_host_temporal(const SvcTemporalParams *p, const double *lp_b, const double *lp_a, const double *lp_zi,
                                 const double *cx, const double *cy, int n_sel, const int32_t *true_inds,
                                 const int32_t *seg, const int32_t *seg_sel, int n_seg, int fc,
                                 double *xi, double *yi, double *xs, double *ys) {
    if (!p || p->struct_size != sizeof(SvcTemporalParams)) { svc_set_error("svc_host_temporal: SvcTemporalParams.struct_size does not match this library"); return SVC_E_INVALID; }
    if (n_sel < 1 || fc < 1 || n_seg < 1 || !cx || !cy || !true_inds || !seg || !seg_sel || !xi || !yi || !xs || !ys ||
        (p->lp_filt && (p->lp_taps < 1 || p->lp_taps > 64 || !lp_b || !lp_a || (p->lp_taps > 1 && !lp_zi)))) {
        svc_set_error("svc_host_temporal: invalid argument");
        return SVC_E_INVALID;
    }
    // sc_interpolate (:1550-1597): per shot, the selected frames' centres -> one centre per decoded frame, appended
    int pos = 0;
    std::vector<double> st;
    for (int s = 0; s < n_seg; ++s) {
        const int si = seg[2 * s], ei = seg[2 * s + 1] + 1, sis = seg_sel[2 * s], eis = seg_sel[2 * s + 1] + 1;
        const int m = eis - sis, cl = ei - si;
        if (m < 1 || sis < 0 || eis > n_sel || cl < 0 || pos + cl > fc) { svc_set_error("svc_host_temporal: shot %d: inconsistent segmentation", s); return SVC_E_INVALID; }
        st.resize(m);
        int mn = true_inds[sis];
        for (int i = 0; i < m; ++i) mn = std::min(mn, (int)true_inds[sis + i]);
        for (int i = 0; i < m; ++i) st[i] = (double)(true_inds[sis + i] - mn);
        if (interp_segment(st.data(), cx + sis, cy + sis, m, cl, xi + pos, yi + pos)) { svc_set_error("svc_host_temporal: shot %d: frame numbers not increasing", s); return SVC_E_INVALID; }
        pos += cl;
    }
    const int produced = pos;
    // sc_smoothing (:1648-1734): per shot, low-pass then LOESS / Savitzky-Golay
    std::vector<double> lp;
    for (int s = 0; s < n_seg; ++s) {
        const int si = seg[2 * s], ei = seg[2 * s + 1] + 1, cl = ei - si;
        if (si < 0 || ei > produced || cl < 1) { svc_set_error("svc_host_temporal: shot %d outside the interpolated series", s); return SVC_E_INVALID; }
        int win = std::min((int)(p->fr * p->loess_w_secs), cl - 2);
        if (((win % 2) + 2) % 2 == 0) win -= 1;
        lp.resize(cl);
        for (int k = 0; k < 2; ++k) {
            const double *src = (k ? yi : xi) + si;
            double *dst = (k ? ys : xs) + si;
            const double *d = src;
            if (p->lp_filt) { lowpass(lp_b, lp_a, lp_zi, p->lp_taps, src, cl, lp.data()); d = lp.data(); }
            if (smooth_handler(d, cl, p->loess_filt, win, p->loess_degree, dst)) {
                svc_set_error("svc_host_temporal: shot %d: Savitzky-Golay window %d / degree %d not usable on %d frames", s, win, p->loess_degree, cl);
                return SVC_E_INVALID;
            }
        }
    }
    return produced;
}

extern "C" int svc_host_boxes(const double *xs, const double *ys, int fc, int w_orig, int h_orig, int w_process, int h_process,
                              int w_final, int h_final, const int32_t *borders_tblr, int64_t *boxes, int64_t *centres, int32_t *fbb_wh) {
    if (fc < 0 || (fc && (!xs || !ys || !boxes)) || w_orig < 1 || h_orig < 1 || w_process < 1 || h_process < 1) { svc_set_error("svc_host_boxes: invalid argument"); return SVC_E_INVALID; }
    const int bt = borders_tblr ? borders_tblr[0] : 0, bb = borders_tblr ? borders_tblr[1] : 0;
    const int bl = borders_tblr ? borders_tblr[2] : 0, br = borders_tblr ? borders_tblr[3] : 0;
    const double scale_h = (double)h_process / (double)h_orig, scale_w = (double)w_process / (double)w_orig;
    int fbb_w = w_final, fbb_h = h_final;
    if (h_final == h_orig) {
        fbb_h = h_final - bt - bb;
        fbb_w = (int)(((double)fbb_h / (double)h_final) * w_final);
    }
    if (w_final == w_orig) {
        fbb_w = w_final - bl - br;
        fbb_h = (int)(((double)fbb_w / (double)w_final) * h_final);
    }
    if (fbb_wh) { fbb_wh[0] = fbb_w; fbb_wh[1] = fbb_h; }
    const int64_t hw1 = (int64_t)(fbb_w / 2.0), hw2 = fbb_w - hw1, hh1 = (int64_t)(fbb_h / 2.0), hh2 = fbb_h - hh1;
    for (int i = 0; i < fc; ++i) {
        const int64_t x = (int64_t)trunc(xs[i] / scale_w), y = (int64_t)trunc(ys[i] / scale_h);
        if (centres) { centres[2 * i] = x; centres[2 * i + 1] = y; }
        int64_t x1 = x - hw1, x2 = x + hw2, y1 = y - hh1, y2 = y + hh2;
        if (x1 < bl) { x1 = bl; x2 = bl + fbb_w; }                 // the four clamps in the reference's order (:1027-1044)
        if (x2 > w_orig - br) { x2 = w_orig - br; x1 = x2 - fbb_w; }
        if (y1 < bt) { y1 = bt; y2 = bt + fbb_h; }
        if (y2 > h_orig - bb) { y2 = h_orig - bb; y1 = y2 - fbb_h; }
        boxes[4 * i] = x1; boxes[4 * i + 1] = y1; boxes[4 * i + 2] = x2; boxes[4 * i + 3] = y2;
    }
    return SVC_OK;
}
