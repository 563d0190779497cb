// Reference update of one iteration on the device: what the
// reference's main node does on the CPU between two searches (test_mref_gpu_align.py:517-564,
// test_reffree_gpu_align.py:374-429):
//   fsc / fsc_mask between the even and odd class sums (EMData::calc_fourier_shell_correlation),
//   the default user function ref_ali2d: filt_tanl (tangent low-pass; formula as in the
//   reference's own cu_apply_tanl_filter_to_tex, cuda/gpu_aln_noref.cu:786-816) and
//   center_2D method 1 (EMData::phase_cog + fshift), then normalize.mask.
// Images are small (90^2 ... 256^2) and few (R ... 2R per iteration), so the 2-D transforms are
// separable direct DFTs with double accumulation -- any image size, no FFT plan, exact to f32.
#pragma once
#include <vector>

#include <hip/hip_runtime.h>

namespace ralign {

// twiddle table: tw[t] = (cos, sin)(2 pi t / nx), t < nx, in double
// pass 1 of the forward transform: T[y][kx] = sum_x v(y,x) e^{-2 pi i kx x / nx}, kx <= nx/2;
// v = img, or (img - mean) * mask for fsc_mask (mean = per-image mean under the mask)
__global__ void dft_rows_kernel(int nx, const float *__restrict__ img, const float *__restrict__ mask,
                                const float *__restrict__ mean, const double2 *__restrict__ tw, double2 *__restrict__ T)
{
    const int nxh = nx / 2 + 1, m = blockIdx.x, y = blockIdx.y;
    const float *row = img + ((size_t)m * nx + y) * nx;
    const float mu = mean ? mean[m] : 0.f;
    for (int kx = threadIdx.x; kx < nxh; kx += blockDim.x) {
        double re = 0, im = 0;
        int t = 0;
        for (int x = 0; x < nx; x++) {
            float v = row[x];
            if (mask) v = (v - mu) * mask[y * nx + x];
            const double2 w = tw[t];
            re += (double)v * w.x; im -= (double)v * w.y;
            t += kx; if (t >= nx) t -= nx;
        }
        T[((size_t)m * nx + y) * nxh + kx] = make_double2(re, im);
    }
}

// pass 2: F[ky][kx] = sum_y T[y][kx] e^{-SIGN... 2 pi i ky y / nx}; SIGN = -1 forward, +1 inverse
template <int SIGN>
__global__ void dft_cols_kernel(int nx, const double2 *__restrict__ T, const double2 *__restrict__ tw, double2 *__restrict__ F)
{
    const int nxh = nx / 2 + 1, m = blockIdx.x, ky = blockIdx.y;
    for (int kx = threadIdx.x; kx < nxh; kx += blockDim.x) {
        double re = 0, im = 0;
        int t = 0;
        for (int y = 0; y < nx; y++) {
            const double2 v = T[((size_t)m * nx + y) * nxh + kx];
            const double c = tw[t].x, s = SIGN * tw[t].y;
            re += v.x * c - v.y * s; im += v.x * s + v.y * c;
            t += ky; if (t >= nx) t -= nx;
        }
        F[((size_t)m * nx + ky) * nxh + kx] = make_double2(re, im);
    }
}

// last pass of the inverse transform (complex half spectrum -> real): out[y][x] =
// 1/nx^2 * sum_kx c_kx Re(U[y][kx] e^{+2 pi i kx x / nx}), c = 1 for kx = 0 and the even Nyquist, else 2
__global__ void idft_rows_kernel(int nx, const double2 *__restrict__ U, const double2 *__restrict__ tw, float *__restrict__ out)
{
    const int nxh = nx / 2 + 1, m = blockIdx.x, y = blockIdx.y;
    const double2 *row = U + ((size_t)m * nx + y) * nxh;
    const double sc = 1.0 / ((double)nx * nx);
    for (int x = threadIdx.x; x < nx; x += blockDim.x) {
        double acc = 0;
        int t = 0;
        for (int kx = 0; kx < nxh; kx++) {
            const double2 v = row[kx], w = tw[t];
            const double term = v.x * w.x - v.y * w.y;
            acc += (kx == 0 || 2 * kx == nx) ? term : 2.0 * term;
            t += x; if (t >= nx) t -= nx;
        }
        out[((size_t)m * nx + y) * nx + x] = (float)(acc * sc);
    }
}

// per-image mean under the mask (Util.infomask(img, m, True)[0]) for fsc_mask
__global__ __launch_bounds__(256) void masked_mean_kernel(int npix, const float *__restrict__ img, const float *__restrict__ mask,
                                                          float *__restrict__ mean)
{
    __shared__ double sh[4];
    __shared__ int shn[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    double s = 0; int n = 0;
    for (int i = tid; i < npix; i += blockDim.x)
        if (mask[i] > 0.5f) { s += img[(size_t)m * npix + i]; n++; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); n += __shfl_xor(n, o); }
    if ((tid & 63) == 0) { sh[tid >> 6] = s; shn[tid >> 6] = n; }
    __syncthreads();
    if (tid == 0) {
        s = sh[0] + sh[1] + sh[2] + sh[3]; n = shn[0] + shn[1] + shn[2] + shn[3];
        mean[m] = n ? (float)(s / n) : 0.f;
    }
}

// Fourier shell correlation of image pairs (2c, 2c+1).
// EMData::calc_fourier_shell_correlation, w = 1: shell = round(inc * sqrt((kx/nx2)^2 + (ky/ny2)^2)),
// inc = nx/2; the kx = 0 column counts ky >= 0 only.  out[c][0][r] = fsc, out[c][1][r] = points.
// shell_off [len + 1], shell_idx: the coefficients (ky * nxh + kx) of every shell in scan order (ky outer, kx inner), built once
// on the host (fsc_shell_table).  Thread (q, r) sums quarter q of shell r's list in list order, the quarters are added in
// order: a fixed summation order (bitwise reproducible run to run).
#define RA_FSC_PARTS 4
inline void fsc_shell_table(int nx, std::vector<int> &off, std::vector<int> &idx)
{
    const int nxh = nx / 2 + 1, inc = nx / 2, len = inc + 1;
    const float d2 = 1.0f / (float)inc / (float)inc;
    std::vector<std::vector<int>> lists(len);
    for (int ky = 0; ky < nx; ky++) {
        const int kys = ky > inc ? ky - nx : ky;
        for (int kx = 0; kx < nxh; kx++) {
            if (kx == 0 && kys < 0) continue;
            const float argx = 0.5f * sqrtf((float)(kys * kys) * d2 + (float)(kx * kx) * d2);
            const int rr = (int)floorf((float)inc * 2.0f * argx + 0.5f);
            if (rr < len) lists[rr].push_back(ky * nxh + kx);
        }
    }
    off.assign(1, 0); idx.clear();
    for (int r = 0; r < len; r++) { idx.insert(idx.end(), lists[r].begin(), lists[r].end()); off.push_back((int)idx.size()); }
}

__global__ void fsc_kernel(int nx, const double2 *__restrict__ F, const int *__restrict__ shell_off, const int *__restrict__ shell_idx,
                           float *__restrict__ out)
{
    extern __shared__ double fsc_part[];        // [RA_FSC_PARTS][len][4]
    const int nxh = nx / 2 + 1, c = blockIdx.x, len = nx / 2 + 1;
    const double2 *f = F + (size_t)(2 * c) * nx * nxh, *g = f + (size_t)nx * nxh;
    for (int t = threadIdx.x; t < RA_FSC_PARTS * len; t += blockDim.x) {
        const int q = t / len, r = t - q * len;
        const int o0 = shell_off[r], cnt = shell_off[r + 1] - o0;
        const int j0 = o0 + cnt * q / RA_FSC_PARTS, j1 = o0 + cnt * (q + 1) / RA_FSC_PARTS;
        double ret = 0, n1 = 0, n2 = 0, lr = 0;
        for (int j = j0; j < j1; j++) {
            const int i = shell_idx[j];
            const double2 a = f[i], b = g[i];
            ret += a.x * b.x + a.y * b.y; n1 += a.x * a.x + a.y * a.y; n2 += b.x * b.x + b.y * b.y; lr += 2;
        }
        double *dst = fsc_part + ((size_t)q * len + r) * 4;
        dst[0] = ret; dst[1] = n1; dst[2] = n2; dst[3] = lr;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < len; r += blockDim.x) {
        double ret = 0, n1 = 0, n2 = 0, lr = 0;
        for (int q = 0; q < RA_FSC_PARTS; q++) {
            const double *src = fsc_part + ((size_t)q * len + r) * 4;
            ret += src[0]; n1 += src[1]; n2 += src[2]; lr += src[3];
        }
        out[((size_t)c * 2) * len + r] = (lr > 0 && n1 > 0 && n2 > 0) ? (float)(ret / sqrt(n1 * n2)) : 0.f;
        out[((size_t)c * 2 + 1) * len + r] = (float)lr;
    }
}

// Average FSC curve of the live classes and sp_filter.fit_tanh(dres, low = 0.1) with sp_utilities.amoeba on the DEVICE, so that
// the reference update of an iteration needs no host round trip (the host fit is ~210 us of libm tanh between two
// synchronisations).  One wave; lane i evaluates frequency point i (i + 64, ...) of the objective, the points are summed by
// lane and then over the wave, the simplex logic is the host routine (ra_fit_tanh) statement for statement, in double.
// in:  fsc_all [nref][2][len] (fsc_kernel), counts [nref]
// out: fit = {fl, aa after the caller's clamps fl_lo <= fl <= fl_hi, aa <= aa_hi; fl, aa as fitted; status (0 ok, 1: every class
//      below min_count)}, curve [3][len] = {frequency, averaged fsc as fit_tanh leaves it (zeroed behind its first drop below
//      `low`), points per shell of the last live class}: what sp_statistics.fsc returns and test_mref_gpu_align.py:537-548 averages
__global__ __launch_bounds__(64) void fsc_fit_kernel(int nx, int nref, const float *__restrict__ fsc_all, const int *__restrict__ counts,
                                                     int min_count, float fl_lo, float fl_hi, float aa_hi, float *__restrict__ fit,
                                                     float *__restrict__ curve)
{
    extern __shared__ double fit_lds[];          // [len] terms of the objective | [len] freq | [len] fsc (floats behind)
    const int len = nx / 2 + 1, lane = threadIdx.x;
    double *term = fit_lds;
    float *freq = reinterpret_cast<float *>(fit_lds + len), *fsc = freq + len;
    int live = 0, last = -1;
    for (int j = 0; j < nref; j++)
        if (counts[j] >= min_count) { live++; last = j; }
    if (live == 0) {
        if (lane == 0) { fit[0] = fl_hi; fit[1] = aa_hi; fit[2] = fit[3] = 0.f; fit[4] = 1.f; }
        return;
    }
    // ave_fsc / c_fsc of the reference's loop; kept only if its sum is not 0 (ra_class_fsc)
    for (int i = lane; i < len; i += 64) {
        double a = 0.0;
        for (int j = 0; j < nref; j++)
            if (counts[j] >= min_count) a += fsc_all[((size_t)j * 2) * len + i];
        term[i] = a;
    }
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < len; i++) tot += term[i];
    for (int i = lane; i < len; i += 64) {
        freq[i] = (float)((double)i / (2.0 * (nx / 2)));
        fsc[i] = (tot != 0.0) ? (float)(term[i] / live) : fsc_all[((size_t)last * 2) * len + i];
        curve[i] = freq[i];
        curve[2 * len + i] = fsc_all[((size_t)last * 2 + 1) * len + i];
    }
    __syncthreads();
    const int n = len;
    const double low = 0.1;
    // "setzero": everything behind the first i >= 1 with 2 fsc / (1 + fsc) < low
    int first = n;
    for (int i = 1; i < n; i++)
        if (2 * (double)fsc[i] / (1.0 + fsc[i]) < low) { first = i; break; }
    __syncthreads();
    for (int i = first + lane; i < n; i += 64) fsc[i] = 0.0f;
    __syncthreads();
    double f0 = -1.0;
    for (int i = 1; i < n - 1; i++)
        if (2 * (double)fsc[i] / (1.0 + fsc[i]) < 0.5) { f0 = freq[i - 1]; break; }
    float fl, aa;
    if (f0 < 0.0) {
        if (fsc[n - 1] < 0.5f) { fl = 0.5f; aa = 0.2f; } else { fl = 0.49f; aa = 0.1f; }
    } else {
        // the objective: f_i = 2 r_i / (1 + r_i) is the same in every evaluation (fsc[0] < 0 is flipped once, as the host
        // routine does in its first call), and pi / (2 a1 a0) is formed once per evaluation instead of by three divisions per
        // point (another rounding of the same quantity: the fit agrees with the host's to ~1e-7, its tolerances are 1e-4)
        __syncthreads();
        if (lane == 0 && fsc[0] < 0.0f) fsc[0] *= -1.0f;
        __syncthreads();
        double fi[2] = {0.0, 0.0}, fr[2] = {0.0, 0.0};          // up to 128 frequency points: two per lane
        for (int i = lane, u = 0; i < n && u < 2; i += 64, u++) { const double r = fsc[i]; fi[u] = 2 * r / (1.0 + r); fr[u] = freq[i]; }
        const int npt = (n > lane ? 1 : 0) + (n > lane + 64 ? 1 : 0);
        auto func = [&](const double *a) -> double {
            double v = 0.0;
            const bool on = a[0] != 0 && a[1] != 0;
            const double c = on ? M_PI / (2.0 * a[1] * a[0]) : 0.0;
            for (int u = 0; u < npt; u++) {
                double qt = 0;
                if (on) qt = fi[u] - 0.5 * (tanh(c * (fr[u] + a[0])) - tanh(c * (fr[u] - a[0])));
                v -= qt * qt;
            }
            if (n > 128)          // longer curves (boxes beyond 254 pixels): the remaining points the plain way
                for (int i = lane + 128; i < n; i += 64) {
                    const double r = fsc[i], f = 2 * r / (1.0 + r);
                    const double qt = on ? f - 0.5 * (tanh(c * (freq[i] + a[0])) - tanh(c * (freq[i] - a[0]))) : 0.0;
                    v -= qt * qt;
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            return v;
        };
        const double scale[2] = {0.05, 0.05}, ftol = 1.e-4, xtol = 1.e-4;
        double sx[3][2] = {{f0, 0.1}, {f0 + scale[0], 0.1}, {f0, 0.1 + scale[1]}}, fv[3];
        for (int i = 0; i < 3; i++) fv[i] = func(sx[i]);
        int iteration = 0, best = 0;
        while (true) {
            int worst = 0; best = 0;
            for (int i = 0; i < 3; i++) { if (fv[i] > fv[best]) best = i; if (fv[i] < fv[worst]) worst = i; }
            double pavg[2] = {0, 0};
            for (int i = 0; i < 3; i++) if (i != worst) { pavg[0] += sx[i][0]; pavg[1] += sx[i][1]; }
            pavg[0] /= 2; pavg[1] /= 2;
            const double simscale = (fabs(pavg[0] - sx[worst][0]) / scale[0] + fabs(pavg[1] - sx[worst][1]) / scale[1]) / 2;
            const double fscale = (fabs(fv[best]) + fabs(fv[worst])) / 2.0;
            const double frange = fscale != 0.0 ? fabs(fv[best] - fv[worst]) / fscale : 0.0;
            if ((frange < ftol && simscale < xtol) || iteration >= 500) break;
            double pnew[2] = {2.0 * pavg[0] - sx[worst][0], 2.0 * pavg[1] - sx[worst][1]};
            double fnew = func(pnew);
            if (fnew <= fv[worst]) {           // worse than the worst: shrink towards the best
                for (int i = 0; i < 3; i++)
                    if (i != best && i != worst) {
                        sx[i][0] = 0.5 * sx[best][0] + 0.5 * sx[i][0]; sx[i][1] = 0.5 * sx[best][1] + 0.5 * sx[i][1];
                        fv[i] = func(sx[i]);
                    }
                pnew[0] = 0.5 * sx[best][0] + 0.5 * sx[worst][0]; pnew[1] = 0.5 * sx[best][1] + 0.5 * sx[worst][1];
                fnew = func(pnew);
            } else if (fnew >= fv[best]) {     // better than the best: try to expand
                double p2[2] = {3.0 * pavg[0] - 2.0 * sx[worst][0], 3.0 * pavg[1] - 2.0 * sx[worst][1]};
                const double f2 = func(p2);
                if (f2 > fnew) { pnew[0] = p2[0]; pnew[1] = p2[1]; fnew = f2; }
            }
            sx[worst][0] = pnew[0]; sx[worst][1] = pnew[1]; fv[worst] = fnew;
            iteration++;
        }
        fl = (float)sx[best][0]; aa = (float)sx[best][1];
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) curve[len + i] = fsc[i];
    if (lane == 0) {
        fit[2] = fl; fit[3] = aa; fit[4] = 0.f;
        // the clamps of sp_user_functions.ref_ali2d (aa = min(aa, 0.2); fl = max(min(0.4, fl), 0.12); notebook/00 log)
        fit[1] = fminf(aa, aa_hi);
        fit[0] = fmaxf(fminf(fl_hi, fl), fl_lo);
    }
}

// filt_tanl on the half spectrum, then center_2D: center = 1 -> phase_cog of the FILTERED image read off
// its (0,1) and (1,0) Fourier coefficients, center = -1 -> the shift cs_in (average-centre rule of the
// reference-free driver), center = 0 -> none; the shift by -cs is applied as a phase ramp (fshift).
// flaa != null: fl, aa = flaa[0], flaa[1] (device memory: fsc_fit_kernel's result)
__global__ void filter_center_kernel(int nx, double2 *__restrict__ F, float fl, float aa, int center,
                                     const float *__restrict__ cs_in, float *__restrict__ cs_out, const float *__restrict__ flaa)
{
    __shared__ double cs[2];
    const int nxh = nx / 2 + 1, m = blockIdx.x, tid = threadIdx.x;
    if (flaa) { fl = flaa[0]; aa = flaa[1]; }
    double2 *f = F + (size_t)m * nx * nxh;
    const double c = M_PI / (2.0 * (double)aa * (double)fl);
    if (fl > 0.f)
        for (int i = tid; i < nx * nxh; i += blockDim.x) {
            const int ky = i / nxh, kx = i - ky * nxh, kys = ky > nx / 2 ? ky - nx : ky;
            const double fx = (double)kx / nx, fy = (double)kys / nx, d = sqrt(fx * fx + fy * fy);
            const double h = 0.5 * (tanh(c * (d + fl)) - tanh(c * (d - fl)));
            f[i].x *= h; f[i].y *= h;
        }
    __syncthreads();
    if (tid == 0) {
        cs[0] = cs[1] = 0.0;
        if (center == 1) {
            // C + iS = sum T_j e^{+i P j} = conj(F[1][0]) for the row sums (y), conj(F[0][1]) for the column sums (x)
            const double P = 2.0 * M_PI / nx;
            double F1 = atan2(-f[nxh].y, f[nxh].x);      // F[ky = 1][kx = 0]
            if (F1 < 0.0) F1 += 2.0 * M_PI;
            cs[1] = F1 / P - nx / 2;
            F1 = atan2(-f[1].y, f[1].x);                  // F[ky = 0][kx = 1]
            if (F1 < 0.0) F1 += 2.0 * M_PI;
            cs[0] = F1 / P - nx / 2;
        } else if (center == -1 && cs_in) {
            cs[0] = cs_in[2 * m]; cs[1] = cs_in[2 * m + 1];
        }
        if (cs_out) { cs_out[2 * m] = (float)cs[0]; cs_out[2 * m + 1] = (float)cs[1]; }
    }
    __syncthreads();
    if (center != 0 && (cs[0] != 0.0 || cs[1] != 0.0))
        for (int i = tid; i < nx * nxh; i += blockDim.x) {
            const int ky = i / nxh, kx = i - ky * nxh, kys = ky > nx / 2 ? ky - nx : ky;
            // fshift(img, -cs): F'(k) = F(k) e^{+2 pi i (kx csx + ky csy) / nx}
            const double ph = 2.0 * M_PI * ((double)kx * cs[0] + (double)kys * cs[1]) / nx;
            const double cc = cos(ph), ss = sin(ph);
            const double2 v = f[i];
            f[i] = make_double2(v.x * cc - v.y * ss, v.x * ss + v.y * cc);
        }
}

// class averages (even + odd) * (1 / count) without the normalisation (Util.add_img + Util.mul_scalar,
// test_mref_gpu_align.py:534-535); classes below min_count are left untouched
__global__ __launch_bounds__(256) void class_average_kernel(int npix, const float *__restrict__ sums, const int *__restrict__ counts,
                                                            int min_count, float *__restrict__ refs)
{
    const int r = blockIdx.x, cnt = counts[r];
    if (cnt < min_count) return;
    const float sc = (float)(1.0 / (double)(float)cnt);
    const float *ev = sums + (size_t)r * 2 * npix, *od = ev + npix;
    for (int i = threadIdx.x; i < npix; i += blockDim.x) refs[(size_t)r * npix + i] = (ev[i] + od[i]) * sc;
}

// normalize.mask(no_sigma=1) in place (test_mref_gpu_align.py:563)
__global__ __launch_bounds__(256) void normalize_mask_kernel(int npix, const float *__restrict__ mask, float *__restrict__ imgs)
{
    __shared__ double sh[2][4];
    __shared__ int shn[4];
    const int tid = threadIdx.x;
    float *dst = imgs + (size_t)blockIdx.x * npix;
    double s = 0, q = 0; int nm = 0;
    for (int i = tid; i < npix; i += blockDim.x)
        if (mask[i] > 0.5f) { const float v = dst[i]; s += v; q += v * (double)v; nm++; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); nm += __shfl_xor(nm, o); }
    if ((tid & 63) == 0) { sh[0][tid >> 6] = s; sh[1][tid >> 6] = q; shn[tid >> 6] = nm; }
    __syncthreads();
    s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    nm = shn[0] + shn[1] + shn[2] + shn[3];
    const float mean = (float)s / nm;
    const float sigma = sqrtf((float)((q - s * s / nm) / (nm - 1)));
    for (int i = tid; i < npix; i += blockDim.x) dst[i] = (dst[i] - mean) / sigma;
}

}  // namespace ralign
