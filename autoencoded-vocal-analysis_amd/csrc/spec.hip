// Shotgun spectrograms on the device: get_spec of the reference
// (ava/preprocessing/utils.py:18-110) for a whole BATCH of windows whose audio already lives in HBM, as called by
// FixedWindowDataset.__getitem__ (ava/models/window_vae_dataset.py:189-256):
//
//   slice audio[max(0,s1):min(len,s2)], subtract its mean   utils.py:62-73      spec_prep_kernel   (1 workgroup / window)
//   scipy.signal.stft (hann, zero boundary, zero padded to whole hops, 'spectrum' scaling), log(|.| + 1e-12)
//                                                           utils.py:74-76      spec_stft_kernel   (1 workgroup / frame)
//   interp2d (bilinear on the (t, f) grid, fill outside), (x - min) / (max - min), clip to [0, 1]
//                                                           utils.py:77-103     spec_interp_kernel (1 thread / pixel)
//
// Everything is fp64, as in the reference (int16 audio - float64 mean -> complex128 STFT): the log turns relative
// errors of small bins into absolute ones, and bins 100 dB under the frame's peak are above the clip floor.  The output
// is the fp32 [n, F, T] batch the VAE consumes (the reference converts with numpy_to_tensor, models/utils.py:444-446).
// Cost is irrelevant next to the train step: a batch of 256 windows is 7 k (finch: 512-point) to 38 k (mouse: 1024-point)
// FFTs, 0.3 to 2 GFLOP.  The frame times and bin frequencies are computed with the reference's own operation order
// (scipy's arange / fs - (nperseg/2) / fs + max(0, t1); rfftfreq's k * (1 / (n d))) and without contraction, so that
// the interval a target point falls into is decided by the same numbers.
#include "common.h"

#define AVA_SPEC_EPS 1e-12

enum { AVA_AUDIO_I16 = 0, AVA_AUDIO_I32 = 1, AVA_AUDIO_F32 = 2, AVA_AUDIO_F64 = 3 };

struct SpecMeta {        // one per window, written by spec_prep_kernel
  long long lo;          // first sample of the slice inside the concatenated audio buffer
  int n;                 // samples in the slice
  int nframes;           // STFT frames (0: the reference returns zeros for this window; -1: scratch too small)
  int j0, j1;            // frames the interpolation can touch (inclusive): the others are never computed
  double mean;           // subtracted DC offset (0 when remove_dc_offset is off)
  double t_shift;        // max(0, t1)
};

struct SpecArgs {
  const void* audio;
  const long long* file_off;
  const long long* file_len;
  const int* file_idx;
  const double* t1;
  const double* t2;
  const double* target_times;    // [n][T]
  const double* target_freqs;    // [F]
  const double* window;          // [nperseg]
  SpecMeta* meta;
  double* twiddle;               // [nperseg/2][2]: exp(-2 pi i k / nperseg), written by spec_prep_kernel's workgroup 0
  double* ftimes;                // [n][maxframes]: frame times of each window, written by spec_prep_kernel
  int* krange;                   // [2]: first / last frequency bin the target frequencies can touch (workgroup 0)
  double* logmag;                // [n][maxframes][nperseg/2 + 1]
  float* out;                    // [n][F][T]
  float* out_max;                // [n] or null
  double* vals;                  // [n][F*T]: clipped fp64 spectrograms handed to spec_normalize_kernel (normalize only)
  double q_gamma;                // within_syll_normalize: np.quantile's interpolation weight ...
  int q_lo, normalize;           // ... between the order statistics q_lo and q_lo + 1 (0-based)
  double fs, scale, spec_min, range, fill_value, fbin;    // fbin: rfftfreq's 1 / (nperseg * (1 / fs))
  int n, maxframes, nperseg, nstep, F, T, dtype, remove_dc;
};

__device__ __forceinline__ double audio_at(const void* base, int dtype, long long i) {
  switch (dtype) {
    case AVA_AUDIO_I16: return (double)reinterpret_cast<const short*>(base)[i];
    case AVA_AUDIO_I32: return (double)reinterpret_cast<const int*>(base)[i];
    case AVA_AUDIO_F32: return (double)reinterpret_cast<const float*>(base)[i];
    default: return reinterpret_cast<const double*>(base)[i];
  }
}

// frame time j of a window: scipy's  arange(nperseg/2, ..., hop) / fs - (nperseg/2) / fs, then utils.py:75's + max(0, t1)
__device__ __forceinline__ double frame_time(int j, const SpecArgs& a, double t_shift) {
  const double half = 0.5 * (double)a.nperseg;
  const double c = __ddiv_rn(half + (double)j * (double)a.nstep, a.fs);
  return __dadd_rn(__dsub_rn(c, __ddiv_rn(half, a.fs)), t_shift);
}

#define AVA_SPEC_PREP_T 1024
// utils.py:58-73: sample range of the window, the "too short" rule, the mean of the slice.  Fixed-order sum (thread
// strides, then a tree): deterministic; exact for integer audio (|sum| < 2^53), where it equals numpy's pairwise sum.
// int16 recordings (the usual wav) are read 8 samples per load.  Also: the range of frames the target times of this
// window can fall between, and (workgroup 0) the twiddle table of the transform.
__global__ __launch_bounds__(AVA_SPEC_PREP_T) void spec_prep_kernel(const SpecArgs a) {
  __shared__ double red[AVA_SPEC_PREP_T];
  __shared__ double tlo[64], thi[64];
  const int w = blockIdx.x, t = threadIdx.x;
  if (w == 0) {
    // power of two: exp(-2 pi i k / N) for k < N / 2 (k / N is exact); any other length: the whole circle, k < N, for the direct
    // transform (spec_dft_kernel), whose index (k n) mod N is exact -- only the quotient k / N is rounded, 1e-16 of the angle
    const int ntw = (a.nperseg & (a.nperseg - 1)) == 0 ? a.nperseg / 2 : a.nperseg;
    for (int k = t; k < ntw; k += AVA_SPEC_PREP_T) {
      double sn, cs;
      sincospi(-2.0 * (double)k / (double)a.nperseg, &sn, &cs);
      a.twiddle[2 * k] = cs;
      a.twiddle[2 * k + 1] = sn;
    }
  }
  if (w == 0 && t < 64) {                                       // bins the target frequencies lie between (+- 2 of slack)
    double mnf = 1e300, mxf = -1e300;
    for (int i = t; i < a.F; i += 64) {
      const double y = a.target_freqs[i];
      if (y < mnf) mnf = y;
      if (y > mxf) mxf = y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double m2 = __shfl_xor(mnf, o, 64), x2 = __shfl_xor(mxf, o, 64);
      mnf = m2 < mnf ? m2 : mnf;
      mxf = x2 > mxf ? x2 : mxf;
    }
    if (t == 0) {
      const int K1 = a.nperseg / 2;
      const double b0 = floor(mnf / a.fbin) - 2.0, b1 = floor(mxf / a.fbin) + 3.0;
      a.krange[0] = b0 > 0.0 ? (b0 < (double)K1 ? (int)b0 : K1) : 0;
      a.krange[1] = b1 < (double)K1 ? (b1 > 0.0 ? (int)b1 : 0) : K1;
    }
  }
  const long long len = a.file_len[a.file_idx[w]];
  const long long base = a.file_off[a.file_idx[w]];
  const double t1 = a.t1[w], t2 = a.t2[w];
  const long long s1 = (long long)rint(__dmul_rn(t1, a.fs));      // int(round(t1*fs)): round half to even
  const long long s2 = (long long)rint(__dmul_rn(t2, a.fs));
  const long long lo = s1 > 0 ? s1 : 0, hi = s2 < len ? s2 : len;
  const long long cnt = hi - lo;
  const bool valid = !(cnt < a.nperseg || s2 <= 0 || s1 >= len);
  double s = 0.0;
  if (valid && a.remove_dc) {
    const long long e0 = base + lo, e1 = e0 + cnt;
    if (a.dtype == AVA_AUDIO_I16) {
      const short* p = reinterpret_cast<const short*>(a.audio);
      long long a0 = (e0 + 7) & ~7ll, a1 = e1 & ~7ll;                 // 16-byte aligned body [a0, a1)
      if (a0 > a1) { a0 = e1; a1 = e1; }
      for (long long i = e0 + t; i < a0; i += AVA_SPEC_PREP_T) s += (double)p[i];
      const int4* p8 = reinterpret_cast<const int4*>(p + a0);
      const long long nv = (a1 - a0) >> 3;
      for (long long v = t; v < nv; v += AVA_SPEC_PREP_T) {
        const int4 q = p8[v];
        const int ws[4] = {q.x, q.y, q.z, q.w};
        int acc = 0;                                                   // 8 int16: exact in int32
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (int)(short)(ws[u] & 0xffff) + (ws[u] >> 16);
        s += (double)acc;
      }
      for (long long i = a1 + t; i < e1; i += AVA_SPEC_PREP_T) s += (double)p[i];
    } else {
      for (long long i = e0 + t; i < e1; i += AVA_SPEC_PREP_T) s += audio_at(a.audio, a.dtype, i);
    }
  }
  red[t] = s;
  // smallest / largest target time of the window (NaNs ignored)
  double mn = 1e300, mx = -1e300;
  for (int i = t; i < a.T; i += AVA_SPEC_PREP_T) {
    const double x = a.target_times[(size_t)w * a.T + i];
    if (x < mn) mn = x;
    if (x > mx) mx = x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double m2 = __shfl_xor(mn, o, 64), x2 = __shfl_xor(mx, o, 64);
    mn = m2 < mn ? m2 : mn;
    mx = x2 > mx ? x2 : mx;
  }
  if ((t & 63) == 0) { tlo[t >> 6] = mn; thi[t >> 6] = mx; }
  __syncthreads();
  for (int o = AVA_SPEC_PREP_T / 2; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  {
    const int nfa = valid ? (int)((cnt + a.nstep - 1) / a.nstep) + 1 : 0;
    if (nfa <= a.maxframes)
      for (int j = t; j < nfa; j += AVA_SPEC_PREP_T) a.ftimes[(size_t)w * a.maxframes + j] = frame_time(j, a, t1 > 0.0 ? t1 : 0.0);
  }
  if (t == 0) {
    SpecMeta m;
    m.lo = base + lo;
    m.n = valid ? (int)cnt : 0;
    // boundary='zeros' adds nperseg/2 on both sides, padded=True fills up to a whole number of hops:
    // frames = ceil(n / hop) + 1
    int nf = valid ? (int)((cnt + a.nstep - 1) / a.nstep) + 1 : 0;
    if (nf > a.maxframes) nf = -1;                                 // caller's max_samples was too small: poisoned below
    m.nframes = nf;
    m.mean = (valid && a.remove_dc) ? red[0] / (double)cnt : 0.0;
    m.t_shift = t1 > 0.0 ? t1 : 0.0;
    for (int i = 1; i < AVA_SPEC_PREP_T / 64; ++i) {
      mn = tlo[i] < mn ? tlo[i] : mn;
      mx = thi[i] > mx ? thi[i] : mx;
    }
    // frames l, l+1 bracket a target time x when l = floor((x - t_0) fs / hop); two frames of slack either side
    int j0 = 0, j1 = nf - 1;
    if (nf > 0) {
      const double x0 = frame_time(0, a, m.t_shift), per = a.fs / (double)a.nstep;
      const double f0 = floor((mn - x0) * per) - 2.0, f1 = floor((mx - x0) * per) + 3.0;
      if (f0 > 0.0) j0 = f0 < (double)(nf - 1) ? (int)f0 : nf - 1;
      if (f1 < (double)(nf - 1)) j1 = f1 > 0.0 ? (int)f1 : 0;
    }
    m.j0 = j0; m.j1 = j1;
    a.meta[w] = m;
    if (a.out_max != nullptr) a.out_max[w] = 0.f;
  }
}

// Frames of one window: Hann window, N-point transform of the real frame as an N/2-point complex FFT of the
// even/odd-interleaved samples (radix-2 decimation in time in LDS) + the split step, log-magnitude of the one-sided
// spectrum.  A workgroup walks the needed frames of its window with stride gridDim.x; the twiddle table is loaded once.
#define PD(i) ((i) + ((i) >> 3))
template <int LOGN>
__global__ __launch_bounds__(256) void spec_stft_kernel(const SpecArgs a) {
  constexpr int N = 1 << LOGN, H = N / 2, LOGH = LOGN - 1;
  // every LDS array is indexed through PD(i) = i + i / 8: one pad double per eight spreads the power-of-two strides of
  // the bit-reversed store, the butterflies and the twiddle look-ups over the banks (73 % of the LDS cycles of the
  // unpadded kernel were bank conflicts)
  __shared__ double re[H + H / 8 + 1], im[H + H / 8 + 1];
  __shared__ double twr[H + H / 8 + 1], twi[H + H / 8 + 1];       // exp(-2 pi i k / N), k < N/2
  const int w = blockIdx.y, t = threadIdx.x;
  const SpecMeta m = a.meta[w];
  if (m.nframes <= 0 || m.j0 + (int)blockIdx.x > m.j1) return;
  for (int k = t; k < H; k += 256) {
    twr[PD(k)] = a.twiddle[2 * k];
    twi[PD(k)] = a.twiddle[2 * k + 1];
  }
  const int k0 = a.krange[0], k1 = a.krange[1];             // bins outside are never read by the interpolation
  // samples of a frame: thread t owns the pairs (2 i, 2 i + 1), i = t + 256 u; raw values are fetched one frame ahead
  constexpr int U = H / 256 > 0 ? H / 256 : 1;
  double raw[U][2];
  auto fetch = [&](int j) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = t + 256 * u;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const long long idx = (long long)j * a.nstep + 2 * i + e - N / 2;      // position in the slice (zeros outside)
        const bool in = i < H && idx >= 0 && idx < m.n;
        const double x = audio_at(a.audio, a.dtype, m.lo + (in ? idx : 0));      // unconditional load, clamped address
        raw[u][e] = in ? x - m.mean : 0.0;                                       // zero boundary / padding
      }
    }
  };
  fetch(m.j0 + blockIdx.x);
  for (int j = m.j0 + blockIdx.x; j <= m.j1; j += gridDim.x) {
    __syncthreads();                                       // twiddles ready / previous frame's reads retired
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = t + 256 * u;
      if (i < H) {
        const double v0 = __dmul_rn(raw[u][0], a.window[2 * i]);
        const double v1 = __dmul_rn(raw[u][1], a.window[2 * i + 1]);
        const int r = (int)(__brev((unsigned)i) >> (32 - LOGH));
        re[PD(r)] = v0;
        im[PD(r)] = v1;
      }
    }
    if (j + (int)gridDim.x <= m.j1) fetch(j + gridDim.x);  // in flight under this frame's butterflies
    __syncthreads();
    // Two radix-2 stages (half = h, then 2 h) per pass over LDS: the four points i0 + {0, h, 2h, 3h} of a group are
    // combined in registers (same operations, same order as two separate stages; half the LDS traffic, which bounds
    // this kernel).  An odd stage count ends with one plain radix-2 pass.
    int st = 0;
#pragma unroll 1
    for (; st + 1 < LOGH; st += 2) {
      const int h = 1 << st;
      for (int b = t; b < H / 4; b += 256) {
        const int pos = b & (h - 1);
        const int i0 = ((b >> st) << (st + 2)) + pos, i1 = i0 + h, i2 = i1 + h, i3 = i2 + h;
        const int k1 = pos << (LOGN - 1 - st), k2 = pos << (LOGN - 2 - st);
        const double w1r = twr[PD(k1)], w1i = twi[PD(k1)], w2r = twr[PD(k2)], w2i = twi[PD(k2)];
        const double x1r = re[PD(i1)], x1i = im[PD(i1)], x3r = re[PD(i3)], x3i = im[PD(i3)];
        const double p1r = w1r * x1r - w1i * x1i, p1i = w1r * x1i + w1i * x1r;
        const double p3r = w1r * x3r - w1i * x3i, p3i = w1r * x3i + w1i * x3r;
        const double u0r = re[PD(i0)], u0i = im[PD(i0)], u2r = re[PD(i2)], u2i = im[PD(i2)];
        const double b0r = u0r + p1r, b0i = u0i + p1i, b1r = u0r - p1r, b1i = u0i - p1i;     // stage st
        const double b2r = u2r + p3r, b2i = u2i + p3i, b3r = u2r - p3r, b3i = u2i - p3i;
        const double q2r = w2r * b2r - w2i * b2i, q2i = w2r * b2i + w2i * b2r;               // stage st + 1: W^pos
        // twiddle of the pair (i1, i3) is W_{4h}^{pos + h} = exp(-2 pi i (pos + h) / (4 h)): entry k2 + N/4 of the table
        const double w3r = twr[PD(k2 + H / 2)], w3i = twi[PD(k2 + H / 2)];
        const double q3r = w3r * b3r - w3i * b3i, q3i = w3r * b3i + w3i * b3r;
        re[PD(i0)] = b0r + q2r; im[PD(i0)] = b0i + q2i;
        re[PD(i2)] = b0r - q2r; im[PD(i2)] = b0i - q2i;
        re[PD(i1)] = b1r + q3r; im[PD(i1)] = b1i + q3i;
        re[PD(i3)] = b1r - q3r; im[PD(i3)] = b1i - q3i;
      }
      __syncthreads();
    }
    if (st < LOGH) {
      const int half = 1 << st;
      for (int b = t; b < H / 2; b += 256) {
        const int pos = b & (half - 1);
        const int i0 = ((b >> st) << (st + 1)) + pos, i1 = i0 + half;
        const int tk = pos << (LOGN - 1 - st);               // exp(-2 pi i pos / (2 half)) in units of the N table
        const double wr = twr[PD(tk)], wi = twi[PD(tk)];
        const double xr = re[PD(i1)], xi = im[PD(i1)];
        const double pr = wr * xr - wi * xi, pi = wr * xi + wi * xr;
        const double ur = re[PD(i0)], ui = im[PD(i0)];
        re[PD(i0)] = ur + pr; im[PD(i0)] = ui + pi;
        re[PD(i1)] = ur - pr; im[PD(i1)] = ui - pi;
      }
      __syncthreads();
    }
    // split: X_k = E_k + W_N^k O_k,  E_k = (Z_k + conj Z_{H-k}) / 2,  O_k = -i (Z_k - conj Z_{H-k}) / 2,  k = 0 .. H
    double* dst = a.logmag + ((size_t)w * a.maxframes + j) * (H + 1);
    for (int k = k0 + t; k <= k1; k += 256) {
      const int ka = k & (H - 1), kb = (H - k) & (H - 1);
      const double zr = re[PD(ka)], zi = im[PD(ka)], cr = re[PD(kb)], ci = -im[PD(kb)];
      const double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
      const double dr = 0.5 * (zr - cr), di = 0.5 * (zi - ci);
      const double orr = di, oi = -dr;                                  // -i (dr + i di)
      const double wr = k == H ? -1.0 : twr[PD(k)], wi = k == H ? 0.0 : twi[PD(k)];
      const double xr = er + (wr * orr - wi * oi), xi = ei + (wr * oi + wi * orr);
      // |X|: no overflow / underflow guard needed at audio magnitudes (numpy's abs is hypot: same value to an ulp)
      dst[k] = log(__dadd_rn(__dmul_rn(sqrt(xr * xr + xi * xi), a.scale), AVA_SPEC_EPS));
    }
  }
}

#undef PD

// The same for a segment length that is NOT a power of two (the reference hands any nperseg to scipy.signal.stft,
// ava/preprocessing/utils.py:66-68; scipy's pocketfft takes any length): the one-sided spectrum by direct summation in fp64,
//   X_k = sum_n v_n exp(-2 pi i (k n mod N) / N),   v = (x - mean) * window,
// for the bins k0 .. k1 the interpolation can touch only.  The frame and the N twiddles live in LDS; a thread owns a bin and walks
// n with the table index advanced by k modulo N (exact integer arithmetic: no argument reduction error).  N^2 work instead of
// N log N -- 0.3 ms for a batch of 256 windows at N = 400 -- on a path whose power-of-two lengths (every example script of the
// reference uses 512 or 1024) keep the radix-2 kernel above.  Error of a direct sum: <= N eps sum |v|, as pocketfft's to a factor.
#define AVA_SPEC_DFT_MAXN 2048
__global__ __launch_bounds__(256) void spec_dft_kernel(const SpecArgs a) {
  __shared__ double v[AVA_SPEC_DFT_MAXN];
  __shared__ double twr[AVA_SPEC_DFT_MAXN], twi[AVA_SPEC_DFT_MAXN];
  const int w = blockIdx.y, t = threadIdx.x, N = a.nperseg, K1 = N / 2;
  const SpecMeta m = a.meta[w];
  if (m.nframes <= 0 || m.j0 + (int)blockIdx.x > m.j1) return;
  for (int k = t; k < N; k += 256) {
    twr[k] = a.twiddle[2 * k];
    twi[k] = a.twiddle[2 * k + 1];
  }
  const int k0 = a.krange[0], k1 = a.krange[1];
  for (int j = m.j0 + blockIdx.x; j <= m.j1; j += gridDim.x) {
    __syncthreads();                                       // twiddles ready / previous frame's reads retired
    for (int i = t; i < N; i += 256) {
      const long long idx = (long long)j * a.nstep + i - N / 2;              // position in the slice (zeros outside)
      const bool in = idx >= 0 && idx < m.n;
      const double x = audio_at(a.audio, a.dtype, m.lo + (in ? idx : 0));
      v[i] = in ? __dmul_rn(x - m.mean, a.window[i]) : 0.0;
    }
    __syncthreads();
    double* dst = a.logmag + ((size_t)w * a.maxframes + j) * (K1 + 1);
    for (int k = k0 + t; k <= k1; k += 256) {
      double xr = 0.0, xi = 0.0;
      int q = 0;                                           // (k n) mod N
      for (int n = 0; n < N; ++n) {
        const double vn = v[n];
        xr = fma(vn, twr[q], xr);
        xi = fma(vn, twi[q], xi);
        q += k;
        if (q >= N) q -= N;
      }
      dst[k] = log(__dadd_rn(__dmul_rn(sqrt(xr * xr + xi * xi), a.scale), AVA_SPEC_EPS));
    }
  }
}

// utils.py:77-103.  Linear B-spline evaluation in FITPACK's order (fpbspl: h0 = f (t[l+1] - x), h1 = f (x - t[l]) with
// f = 1 / (t[l+1] - t[l]); fpbisp: sum over x then y of (c * hx) * hy), then interp2d's out-of-bounds rule, then
// normalisation and clip.  A workgroup owns AVA_SPEC_ROWS frequency rows of one window: the knot interval and the two
// basis values of every target TIME are computed once per workgroup (LDS), those of a target FREQUENCY once per row
// visit, so a pixel costs four loads, the 4-term sum and the normalising division.
#define AVA_SPEC_ROWS 16
#define AVA_SPEC_TMAX 512        // target times per window the column table holds (num_time_bins; larger: AVA_EINVAL)
__global__ __launch_bounds__(256) void spec_interp_kernel(const SpecArgs a) {
  __shared__ double chx0[AVA_SPEC_TMAX], chx1[AVA_SPEC_TMAX];
  __shared__ int cl[AVA_SPEC_TMAX];                                  // knot interval of column ti, -1: outside -> fill value
  __shared__ double rhy0[AVA_SPEC_ROWS], rhy1[AVA_SPEC_ROWS];
  __shared__ int rq[AVA_SPEC_ROWS];
  const int w = blockIdx.y, f0 = blockIdx.x * AVA_SPEC_ROWS, t = threadIdx.x;
  const SpecMeta m = a.meta[w];
  const int rows = a.F - f0 < AVA_SPEC_ROWS ? a.F - f0 : AVA_SPEC_ROWS;
  float* obase = a.out + ((size_t)w * a.F + f0) * a.T;
  if (m.nframes <= 0) {                                              // utils.py:68-69: np.zeros / the "scratch too small" marker
    const float z = m.nframes == 0 ? 0.f : __builtin_nanf("");
    for (int i = t; i < rows * a.T; i += 256) obase[i] = z;
    return;
  }
  const int K = a.nperseg / 2 + 1;
  const double* ft = a.ftimes + (size_t)w * a.maxframes;
  const double val = a.fbin;                       // bin frequencies: rfftfreq(n, d) = arange(n/2 + 1) * (1 / (n d)), d = 1 / fs
  const double xmin = ft[0], xmax = ft[m.nframes - 1];
  for (int ti = t; ti < a.T; ti += 256) {
    const double x = a.target_times[(size_t)w * a.T + ti];
    int l = -1;
    double hx0 = 0.0, hx1 = 0.0;
    if (!(x < xmin || x > xmax || !(x == x))) {
      l = (int)floor((x - xmin) * a.fs / (double)a.nstep);
      l = l < 0 ? 0 : (l > m.nframes - 2 ? m.nframes - 2 : l);
      while (l > 0 && x < ft[l]) --l;
      while (l < m.nframes - 2 && x >= ft[l + 1]) ++l;
      const double tl = ft[l], tr = ft[l + 1];
      const double fx = __ddiv_rn(1.0, __dsub_rn(tr, tl));
      hx0 = __dmul_rn(fx, __dsub_rn(tr, x));
      hx1 = __dmul_rn(fx, __dsub_rn(x, tl));
    }
    cl[ti] = l; chx0[ti] = hx0; chx1[ti] = hx1;
  }
  if (t < rows) {
    const double y = a.target_freqs[f0 + t];
    const double ymax = __dmul_rn((double)(K - 1), val);
    int q = -1;
    double hy0 = 0.0, hy1 = 0.0;
    if (!(y < 0.0 || y > ymax || !(y == y))) {
      q = (int)floor(y / val);
      q = q < 0 ? 0 : (q > K - 2 ? K - 2 : q);
      while (q > 0 && y < __dmul_rn((double)q, val)) --q;
      while (q < K - 2 && y >= __dmul_rn((double)(q + 1), val)) ++q;
      const double yl = __dmul_rn((double)q, val), yr = __dmul_rn((double)(q + 1), val);
      const double fy = __ddiv_rn(1.0, __dsub_rn(yr, yl));
      hy0 = __dmul_rn(fy, __dsub_rn(yr, y));
      hy1 = __dmul_rn(fy, __dsub_rn(y, yl));
    }
    rq[t] = q; rhy0[t] = hy0; rhy1[t] = hy1;
  }
  __syncthreads();
  float fmax = 0.f;
  for (int i = t; i < rows * a.T; i += 256) {
    const int r = i / a.T, ti = i - r * a.T;
    const int l = cl[ti], q = rq[r];
    double v;
    if (l < 0 || q < 0) {
      v = a.fill_value;
    } else {
      const double hx0 = chx0[ti], hx1 = chx1[ti], hy0 = rhy0[r], hy1 = rhy1[r];
      const double* c0 = a.logmag + ((size_t)w * a.maxframes + l) * K + q;       // coefficient c[time l][freq q]
      const double* c1 = c0 + K;
      double sp = 0.0;
      sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(c0[0], hx0), hy0));
      sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(c0[1], hx0), hy1));
      sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(c1[0], hx1), hy0));
      sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(c1[1], hx1), hy1));
      v = sp;
    }
    v = __dsub_rn(v, a.spec_min);
    v = __ddiv_rn(v, a.range);                                         // utils.py:101-102
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    if (a.normalize) {                                                 // utils.py:104-108 follow in spec_normalize_kernel
      a.vals[((size_t)w * a.F + f0) * a.T + i] = v;
    } else {
      const float vf = (float)v;
      obase[i] = vf;
      fmax = vf > fmax ? vf : fmax;
    }
  }
  if (a.out_max != nullptr && !a.normalize) {                          // one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float y = __shfl_xor(fmax, o, 64); fmax = y > fmax ? y : fmax; }
    if ((t & 63) == 0 && fmax > 0.f) atomicMax(reinterpret_cast<int*>(a.out_max + w), __float_as_int(fmax));
  }
}

// within_syll_normalize (utils.py:104-108): spec -= np.quantile(spec, q); spec[spec < 0] = 0; spec /= max(spec) + EPSILON.
// One workgroup per window.  The quantile is numpy's default ('linear'): a[lo] + (a[lo+1] - a[lo]) * gamma with lo and
// gamma from the host (numpy's own expression for the virtual index), evaluated with numpy's two-sided lerp.  a[lo] is
// found by an MSB-first radix select over the bit patterns (the values are in [0, 1], so the unsigned order of the
// patterns is the numeric order; counts are integers: deterministic), a[lo+1] from one more counting pass.
#define AVA_SPEC_NORM_T 1024
__global__ __launch_bounds__(AVA_SPEC_NORM_T) void spec_normalize_kernel(const SpecArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_prefix, sh_next;
  __shared__ unsigned sh_k, sh_le;
  __shared__ double sh_max[AVA_SPEC_NORM_T / 64];
  const int w = blockIdx.x, t = threadIdx.x;
  const SpecMeta m = a.meta[w];
  if (m.nframes <= 0) return;                                        // zeros (or the NaN marker) were written already
  const int n = a.F * a.T;
  const double* v = a.vals + (size_t)w * n;
  unsigned long long prefix = 0;
  unsigned k = (unsigned)a.q_lo;
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
    for (int i = t; i < n; i += AVA_SPEC_NORM_T) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
      if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255ull], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned c = 0, b = 0;
      for (; b < 256; ++b) {
        if (c + hist[b] > k) break;
        c += hist[b];
      }
      sh_prefix = prefix | ((unsigned long long)b << shift);
      sh_k = k - c;
    }
    __syncthreads();
    prefix = sh_prefix;
    k = sh_k;
    __syncthreads();
  }
  const double alo = __longlong_as_double((long long)prefix);
  // a[lo + 1]: alo again when more than lo + 1 values are <= alo, else the smallest value above it; and the maximum
  if (t == 0) { sh_le = 0; sh_next = ~0ull; }
  __syncthreads();
  unsigned le = 0;
  unsigned long long nxt = ~0ull;
  double mx = 0.0;
  for (int i = t; i < n; i += AVA_SPEC_NORM_T) {
    const double x = v[i];
    const unsigned long long key = (unsigned long long)__double_as_longlong(x);
    if (key <= prefix) ++le; else if (key < nxt) nxt = key;
    mx = x > mx ? x : mx;
  }
  atomicAdd(&sh_le, le);
  atomicMin(&sh_next, nxt);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
  if ((t & 63) == 0) sh_max[t >> 6] = mx;
  __syncthreads();
  for (int i = 0; i < AVA_SPEC_NORM_T / 64; ++i) mx = sh_max[i] > mx ? sh_max[i] : mx;
  const double ahi = (sh_le >= (unsigned)a.q_lo + 2u || a.q_lo + 1 >= n) ? alo : __longlong_as_double((long long)sh_next);
  // numpy's _lerp: a + (b - a) * t, replaced by b - (b - a) * (1 - t) where t >= 0.5
  const double diff = __dsub_rn(ahi, alo);
  double qv = __dadd_rn(alo, __dmul_rn(diff, a.q_gamma));
  if (a.q_gamma >= 0.5) qv = __dsub_rn(ahi, __dmul_rn(diff, __dsub_rn(1.0, a.q_gamma)));
  double top = __dsub_rn(mx, qv);                                    // max of the shifted, floored spectrogram
  top = top < 0.0 ? 0.0 : top;
  const double den = __dadd_rn(top, AVA_SPEC_EPS);
  float fmax = 0.f;
  float* o = a.out + (size_t)w * n;
  for (int i = t; i < n; i += AVA_SPEC_NORM_T) {
    double x = __dsub_rn(v[i], qv);
    x = x < 0.0 ? 0.0 : x;
    const float f = (float)__ddiv_rn(x, den);
    o[i] = f;
    fmax = f > fmax ? f : fmax;
  }
  if (a.out_max != nullptr && fmax > 0.f) atomicMax(reinterpret_cast<int*>(a.out_max + w), __float_as_int(fmax));
}

static int frames_for(int max_samples, int nstep) { return (max_samples + nstep - 1) / nstep + 1; }

static bool spec_shape_ok(int nperseg, int noverlap) {
  if (nperseg < 64 || nperseg > 2048) return false;       // a power of two: radix-2 kernel; any other length: direct transform
  return noverlap >= 0 && noverlap < nperseg;
}

extern "C" size_t ava_spec_workspace_bytes(int n, int max_samples, int nperseg, int noverlap, int F, int T, int normalize) {
  if (n <= 0 || max_samples <= 0 || F <= 0 || T <= 0 || !spec_shape_ok(nperseg, noverlap)) return 0;
  const size_t frames = (size_t)frames_for(max_samples, nperseg - noverlap);
  return 256 + 16 + (((size_t)n * sizeof(SpecMeta) + 15) & ~(size_t)15) + (size_t)2 * nperseg * sizeof(double) +
         (size_t)n * frames * sizeof(double) + (size_t)n * frames * (size_t)(nperseg / 2 + 1) * sizeof(double) +
         (normalize ? (size_t)n * F * T * sizeof(double) : 0);
}

extern "C" int ava_get_spec_batch(const void* audio, int audio_dtype, const int64_t* file_off, const int64_t* file_len,
                                  const int32_t* file_idx, const double* t1, const double* t2, const double* target_times,
                                  int n, int max_samples, double fs, int nperseg, int noverlap, const double* window,
                                  double scale, const double* target_freqs, int F, int T, double spec_min, double spec_max,
                                  double fill_value, int remove_dc, int normalize, int q_lo, double q_gamma, float* out,
                                  float* out_max, void* ws, size_t ws_bytes, ava_stream_t s) {
  if (audio == nullptr || file_off == nullptr || file_len == nullptr || file_idx == nullptr || t1 == nullptr ||
      t2 == nullptr || target_times == nullptr || window == nullptr || target_freqs == nullptr || out == nullptr)
    return AVA_EINVAL;
  if (n <= 0 || F <= 0 || T <= 0 || T > AVA_SPEC_TMAX || max_samples <= 0 || !(fs > 0.0) || !spec_shape_ok(nperseg, noverlap)) return AVA_EINVAL;
  if (audio_dtype < AVA_AUDIO_I16 || audio_dtype > AVA_AUDIO_F64) return AVA_EINVAL;
  if (!(spec_max != spec_min)) return AVA_EINVAL;
  if (normalize && (q_lo < 0 || q_lo >= F * T || !(q_gamma >= 0.0 && q_gamma <= 1.0))) return AVA_EINVAL;
  if (ws == nullptr || ws_bytes < ava_spec_workspace_bytes(n, max_samples, nperseg, noverlap, F, T, normalize)) return AVA_EWORKSPACE;
  SpecArgs a;
  a.audio = audio; a.file_off = reinterpret_cast<const long long*>(file_off);
  a.file_len = reinterpret_cast<const long long*>(file_len); a.file_idx = file_idx;
  a.t1 = t1; a.t2 = t2; a.target_times = target_times; a.target_freqs = target_freqs; a.window = window;
  char* base = reinterpret_cast<char*>(ws);
  base += (256 - (reinterpret_cast<uintptr_t>(base) & 255)) & 255;
  a.krange = reinterpret_cast<int*>(base);
  base += 16;
  a.meta = reinterpret_cast<SpecMeta*>(base);
  a.twiddle = reinterpret_cast<double*>(base + (((size_t)n * sizeof(SpecMeta) + 15) & ~(size_t)15));
  a.ftimes = a.twiddle + 2 * nperseg;                 // room for the whole circle (lengths that are not a power of two)
  a.logmag = a.ftimes + (size_t)n * frames_for(max_samples, nperseg - noverlap);
  a.vals = a.logmag + (size_t)n * frames_for(max_samples, nperseg - noverlap) * (size_t)(nperseg / 2 + 1);
  a.normalize = normalize ? 1 : 0; a.q_lo = q_lo; a.q_gamma = q_gamma;
  a.out = out; a.out_max = out_max;
  a.fs = fs; a.scale = scale; a.spec_min = spec_min; a.range = spec_max - spec_min; a.fill_value = fill_value;
  a.fbin = 1.0 / ((double)nperseg * (1.0 / fs));      // host IEEE arithmetic: the very operations of scipy.fft.rfftfreq
  a.n = n; a.nperseg = nperseg; a.nstep = nperseg - noverlap; a.maxframes = frames_for(max_samples, a.nstep);
  a.F = F; a.T = T; a.dtype = audio_dtype; a.remove_dc = remove_dc;
  hipStream_t st = to_stream(s);
  hipLaunchKernelGGL(spec_prep_kernel, dim3(n), dim3(AVA_SPEC_PREP_T), 0, st, a);
  AVA_CHECK_LAUNCH();
  const dim3 fgrid(a.maxframes < 24 ? a.maxframes : 24, n);      // a workgroup strides over its window's needed frames
  if ((nperseg & (nperseg - 1)) != 0) hipLaunchKernelGGL(spec_dft_kernel, fgrid, dim3(256), 0, st, a);
  else switch (nperseg) {
    case 64: hipLaunchKernelGGL(spec_stft_kernel<6>, fgrid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL(spec_stft_kernel<7>, fgrid, dim3(256), 0, st, a); break;
    case 256: hipLaunchKernelGGL(spec_stft_kernel<8>, fgrid, dim3(256), 0, st, a); break;
    case 512: hipLaunchKernelGGL(spec_stft_kernel<9>, fgrid, dim3(256), 0, st, a); break;
    case 1024: hipLaunchKernelGGL(spec_stft_kernel<10>, fgrid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(spec_stft_kernel<11>, fgrid, dim3(256), 0, st, a); break;
  }
  AVA_CHECK_LAUNCH();
  hipLaunchKernelGGL(spec_interp_kernel, dim3(ceil_div(F, AVA_SPEC_ROWS), n), dim3(256), 0, st, a);
  AVA_CHECK_LAUNCH();
  if (normalize) {
    hipLaunchKernelGGL(spec_normalize_kernel, dim3(n), dim3(AVA_SPEC_NORM_T), 0, st, a);
    AVA_CHECK_LAUNCH();
  }
  return AVA_OK;
}
