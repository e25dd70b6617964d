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
  int nframes;           // STFT frames (0: the reference returns zeros for this window)
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
  double* logmag;                // [n][maxframes][nperseg/2 + 1]
  float* out;                    // [n][F][T]
  float* out_max;                // [n] or null
  double fs, scale, spec_min, range, fill_value;
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

// utils.py:58-73: sample range of the window, the "too short" rule, the mean of the slice.  Fixed-order sum (thread
// strides, then a tree): deterministic; exact for integer audio (|sum| < 2^53), where it equals numpy's pairwise sum.
__global__ __launch_bounds__(256) void spec_prep_kernel(const SpecArgs a) {
  __shared__ double red[256];
  const int w = blockIdx.x, t = threadIdx.x;
  const long long len = a.file_len[a.file_idx[w]];
  const long long base = a.file_off[a.file_idx[w]];
  const double t1 = a.t1[w], t2 = a.t2[w];
  const long long s1 = (long long)rint(__dmul_rn(t1, a.fs));      // int(round(t1*fs)): round half to even
  const long long s2 = (long long)rint(__dmul_rn(t2, a.fs));
  const long long lo = s1 > 0 ? s1 : 0, hi = s2 < len ? s2 : len;
  const long long cnt = hi - lo;
  const bool valid = !(cnt < a.nperseg || s2 <= 0 || s1 >= len);
  double s = 0.0;
  if (valid && a.remove_dc)
    for (long long i = t; i < cnt; i += 256) s += audio_at(a.audio, a.dtype, base + lo + i);
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
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
    a.meta[w] = m;
    if (a.out_max != nullptr) a.out_max[w] = 0.f;
  }
}

// One frame: window, N-point FFT (radix-2 decimation in time in LDS, twiddle table in LDS), log-magnitude of the
// one-sided spectrum.
template <int LOGN>
__global__ __launch_bounds__(256) void spec_stft_kernel(const SpecArgs a) {
  constexpr int N = 1 << LOGN;
  __shared__ double re[N], im[N];
  __shared__ double twr[N / 2], twi[N / 2];
  const int j = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
  const SpecMeta m = a.meta[w];
  if (j >= m.nframes) return;
  for (int k = t; k < N / 2; k += 256) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)N, &sn, &cs);      // exp(-2 pi i k / N); k / N is exact
    twr[k] = cs;
    twi[k] = sn;
  }
  for (int i = t; i < N; i += 256) {
    const long long idx = (long long)j * a.nstep + i - N / 2;      // position in the slice (zeros outside)
    double v = 0.0;
    if (idx >= 0 && idx < m.n) v = __dmul_rn(audio_at(a.audio, a.dtype, m.lo + idx) - m.mean, a.window[i]);
    const int r = (int)(__brev((unsigned)i) >> (32 - LOGN));
    re[r] = v;
    im[r] = 0.0;
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < LOGN; ++s) {
    const int half = 1 << s;
    for (int b = t; b < N / 2; b += 256) {
      const int pos = b & (half - 1);
      const int i0 = ((b >> s) << (s + 1)) + pos, i1 = i0 + half;
      const int tk = pos << (LOGN - 1 - s);
      const double wr = twr[tk], wi = twi[tk];
      const double xr = re[i1], xi = im[i1];
      const double pr = wr * xr - wi * xi, pi = wr * xi + wi * xr;
      const double ur = re[i0], ui = im[i0];
      re[i0] = ur + pr; im[i0] = ui + pi;
      re[i1] = ur - pr; im[i1] = ui - pi;
    }
    __syncthreads();
  }
  double* dst = a.logmag + ((size_t)w * a.maxframes + j) * (N / 2 + 1);
  for (int k = t; k <= N / 2; k += 256) dst[k] = log(__dadd_rn(__dmul_rn(hypot(re[k], im[k]), a.scale), AVA_SPEC_EPS));
}

// frame time j of a window: scipy's  arange(nperseg/2, ..., hop) / fs - (nperseg/2) / fs, then utils.py:75's + max(0, t1)
__device__ __forceinline__ double frame_time(int j, const SpecArgs& a, double t_shift) {
  const double half = 0.5 * (double)a.nperseg;
  const double c = __ddiv_rn(half + (double)j * (double)a.nstep, a.fs);
  return __dadd_rn(__dsub_rn(c, __ddiv_rn(half, a.fs)), t_shift);
}

// utils.py:77-103 for one output pixel.  Linear B-spline evaluation in FITPACK's order (fpbspl: h0 = f (t[l+1] - x),
// h1 = f (x - t[l]) with f = 1 / (t[l+1] - t[l]); fpbisp: sum over x then y of (c * hx) * hy), then interp2d's
// out-of-bounds rule, then normalisation and clip.
__global__ __launch_bounds__(256) void spec_interp_kernel(const SpecArgs a) {
  const int w = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.F * a.T) return;
  const int fi = p / a.T, ti = p - fi * a.T;
  const SpecMeta m = a.meta[w];
  float* o = a.out + ((size_t)w * a.F + fi) * a.T + ti;
  if (m.nframes == 0) { *o = 0.f; return; }                          // utils.py:68-69: np.zeros
  if (m.nframes < 0) { *o = __builtin_nanf(""); return; }            // workspace too small for this window: loud
  const int K = a.nperseg / 2 + 1;
  const double x = a.target_times[(size_t)w * a.T + ti], y = a.target_freqs[fi];
  // bin frequencies: rfftfreq(n, d) = arange(n/2 + 1) * (1 / (n d)), d = 1 / fs
  const double val = __ddiv_rn(1.0, __dmul_rn((double)a.nperseg, __ddiv_rn(1.0, a.fs)));
  const double xmin = frame_time(0, a, m.t_shift), xmax = frame_time(m.nframes - 1, a, m.t_shift);
  const double ymin = 0.0, ymax = __dmul_rn((double)(K - 1), val);
  double v;
  if (x < xmin || x > xmax || y < ymin || y > ymax || !(x == x) || !(y == y)) {
    v = a.fill_value;
  } else {
    int l = (int)floor((x - xmin) * a.fs / (double)a.nstep);
    l = l < 0 ? 0 : (l > m.nframes - 2 ? m.nframes - 2 : l);
    while (l > 0 && x < frame_time(l, a, m.t_shift)) --l;
    while (l < m.nframes - 2 && x >= frame_time(l + 1, a, m.t_shift)) ++l;
    int q = (int)floor(y / val);
    q = q < 0 ? 0 : (q > K - 2 ? K - 2 : q);
    while (q > 0 && y < __dmul_rn((double)q, val)) --q;
    while (q < K - 2 && y >= __dmul_rn((double)(q + 1), val)) ++q;
    const double tl = frame_time(l, a, m.t_shift), tr = frame_time(l + 1, a, m.t_shift);
    const double fx = __ddiv_rn(1.0, __dsub_rn(tr, tl));
    const double hx0 = __dmul_rn(fx, __dsub_rn(tr, x)), hx1 = __dmul_rn(fx, __dsub_rn(x, tl));
    const double yl = __dmul_rn((double)q, val), yr = __dmul_rn((double)(q + 1), val);
    const double fy = __ddiv_rn(1.0, __dsub_rn(yr, yl));
    const double hy0 = __dmul_rn(fy, __dsub_rn(yr, y)), hy1 = __dmul_rn(fy, __dsub_rn(y, yl));
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
  const float vf = (float)v;
  *o = vf;
  if (a.out_max != nullptr && vf > 0.f) atomicMax(reinterpret_cast<int*>(a.out_max + w), __float_as_int(vf));
}

static int frames_for(int max_samples, int nstep) { return (max_samples + nstep - 1) / nstep + 1; }

static bool spec_shape_ok(int nperseg, int noverlap) {
  if (nperseg < 64 || nperseg > 2048 || (nperseg & (nperseg - 1)) != 0) return false;
  return noverlap >= 0 && noverlap < nperseg;
}

extern "C" size_t ava_spec_workspace_bytes(int n, int max_samples, int nperseg, int noverlap) {
  if (n <= 0 || max_samples <= 0 || !spec_shape_ok(nperseg, noverlap)) return 0;
  const size_t frames = (size_t)frames_for(max_samples, nperseg - noverlap);
  return 256 + (size_t)n * sizeof(SpecMeta) + (size_t)n * frames * (size_t)(nperseg / 2 + 1) * sizeof(double);
}

extern "C" int ava_get_spec_batch(const void* audio, int audio_dtype, const int64_t* file_off, const int64_t* file_len,
                                  const int32_t* file_idx, const double* t1, const double* t2, const double* target_times,
                                  int n, int max_samples, double fs, int nperseg, int noverlap, const double* window,
                                  double scale, const double* target_freqs, int F, int T, double spec_min, double spec_max,
                                  double fill_value, int remove_dc, float* out, float* out_max, void* ws, size_t ws_bytes,
                                  ava_stream_t s) {
  if (audio == nullptr || file_off == nullptr || file_len == nullptr || file_idx == nullptr || t1 == nullptr ||
      t2 == nullptr || target_times == nullptr || window == nullptr || target_freqs == nullptr || out == nullptr)
    return AVA_EINVAL;
  if (n <= 0 || F <= 0 || T <= 0 || max_samples <= 0 || !(fs > 0.0) || !spec_shape_ok(nperseg, noverlap)) return AVA_EINVAL;
  if (audio_dtype < AVA_AUDIO_I16 || audio_dtype > AVA_AUDIO_F64) return AVA_EINVAL;
  if (!(spec_max != spec_min)) return AVA_EINVAL;
  if (ws == nullptr || ws_bytes < ava_spec_workspace_bytes(n, max_samples, nperseg, noverlap)) return AVA_EWORKSPACE;
  SpecArgs a;
  a.audio = audio; a.file_off = reinterpret_cast<const long long*>(file_off);
  a.file_len = reinterpret_cast<const long long*>(file_len); a.file_idx = file_idx;
  a.t1 = t1; a.t2 = t2; a.target_times = target_times; a.target_freqs = target_freqs; a.window = window;
  char* base = reinterpret_cast<char*>(ws);
  base += (256 - (reinterpret_cast<uintptr_t>(base) & 255)) & 255;
  a.meta = reinterpret_cast<SpecMeta*>(base);
  a.logmag = reinterpret_cast<double*>(base + (((size_t)n * sizeof(SpecMeta) + 15) & ~(size_t)15));
  a.out = out; a.out_max = out_max;
  a.fs = fs; a.scale = scale; a.spec_min = spec_min; a.range = spec_max - spec_min; a.fill_value = fill_value;
  a.n = n; a.nperseg = nperseg; a.nstep = nperseg - noverlap; a.maxframes = frames_for(max_samples, a.nstep);
  a.F = F; a.T = T; a.dtype = audio_dtype; a.remove_dc = remove_dc;
  hipStream_t st = to_stream(s);
  hipLaunchKernelGGL(spec_prep_kernel, dim3(n), dim3(256), 0, st, a);
  AVA_CHECK_LAUNCH();
  const dim3 fgrid(a.maxframes, n);
  switch (nperseg) {
    case 64: hipLaunchKernelGGL(spec_stft_kernel<6>, fgrid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL(spec_stft_kernel<7>, fgrid, dim3(256), 0, st, a); break;
    case 256: hipLaunchKernelGGL(spec_stft_kernel<8>, fgrid, dim3(256), 0, st, a); break;
    case 512: hipLaunchKernelGGL(spec_stft_kernel<9>, fgrid, dim3(256), 0, st, a); break;
    case 1024: hipLaunchKernelGGL(spec_stft_kernel<10>, fgrid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(spec_stft_kernel<11>, fgrid, dim3(256), 0, st, a); break;
  }
  AVA_CHECK_LAUNCH();
  hipLaunchKernelGGL(spec_interp_kernel, dim3(ceil_div(F * T, 256), n), dim3(256), 0, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
