// Audio front end of LlamaAudio (modelling/audio.py:26-36,53-60):
//   MelSpectrogram(sample_rate 16k, n_fft 512, win 400 hann periodic, hop 160, 128 slaney mels, power 2, centre/reflect)
//   -> [..., :-1].clip(1e-12).log10() -> minus per-bin mean over time -> bf16, written TIME-MAJOR with one zero row of
//   padding on each side, which makes the im2col matrix of the k=3 convolutions a strided view for the GEMM kernel.
// The STFT arithmetic follows torchaudio's documented semantics (the package is absent: parity unpinned).
// fp32 throughout: a direct 512-point DFT per frame (2 GFLOP per 40 s clip - noise next to the decoder's 139 TFLOP step).
#include "common.h"

#define NFFT 512
#define NBIN 257

// One block (256 threads) per frame.  tw: [512][2] (cos, sin) of 2*pi*j/512; win: [512] window already zero-padded/centred;
// fb: [257][n_mels] filterbank.  mel out: [B][n_mels][n_frames] fp32 (the layout MelSpectrogram returns).
__global__ __launch_bounds__(256) void mel_power_kernel(const float* __restrict__ audio, int64_t L, const float* __restrict__ tw,
                                                        const float* __restrict__ win, const float* __restrict__ fb, float* __restrict__ mel,
                                                        int n_frames, int hop, int n_mels) {
  __shared__ float xs[NFFT];
  __shared__ float cs[NFFT], sn[NFFT];
  __shared__ float pw[NBIN + 3];
  const int t = blockIdx.x, b = blockIdx.y;
  const float* a = audio + (int64_t)b * L;
  for (int n = threadIdx.x; n < NFFT; n += 256) {
    int64_t i = (int64_t)t * hop + n - NFFT / 2;  // centre=True: frame t is centred on sample t*hop
    if (i < 0) i = -i;                             // reflect padding (no edge repeat)
    if (i >= L) i = 2 * (L - 1) - i;
    i = i < 0 ? 0 : (i >= L ? L - 1 : i);
    xs[n] = a[i] * win[n];
    cs[n] = tw[2 * n];
    sn[n] = tw[2 * n + 1];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NBIN; k += 256) {
    float re = 0.f, im = 0.f;
    int idx = 0;
#pragma unroll 8
    for (int n = 0; n < NFFT; ++n) {
      re += xs[n] * cs[idx];
      im -= xs[n] * sn[idx];
      idx = (idx + k) & (NFFT - 1);
    }
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < n_mels; m += 256) {
    float acc = 0.f;
    for (int f = 0; f < NBIN; ++f) acc += pw[f] * fb[f * n_mels + m];
    mel[((int64_t)b * n_mels + m) * n_frames + t] = acc;
  }
}

// One block per (mel bin, batch): log10(clip(mel[..., :-1], 1e-12)) - mean_t, bf16, feat[b][1 + t][m] (row stride n_mels).
__global__ __launch_bounds__(256) void logmel_cmn_kernel(const float* __restrict__ mel, bf16_t* __restrict__ feat, int n_frames, int n_mels) {
  __shared__ float red[16];
  const int m = blockIdx.x, b = blockIdx.y;
  const int T = n_frames - 1;
  const float* src = mel + ((int64_t)b * n_mels + m) * n_frames;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s += log10f(fmaxf(src[t], 1e-12f));
  s = block_sum(s, red);
  const float mean = s / (float)T;
  bf16_t* dst = feat + (int64_t)b * (T + 2) * n_mels + m;
  for (int t = threadIdx.x; t < T; t += 256) dst[(int64_t)(t + 1) * n_mels] = f2bf(log10f(fmaxf(src[t], 1e-12f)) - mean);
  if (threadIdx.x == 0) { dst[0] = 0; dst[(int64_t)(T + 1) * n_mels] = 0; }
}

extern "C" int llx_mel_spectrogram(const float* audio, int64_t B, int64_t L, const float* twiddle, const float* window, const float* fbank,
                                   float* mel, int64_t n_frames, int64_t hop, int64_t n_mels, hipStream_t stream) {
  LLX_REQUIRE(audio && twiddle && window && fbank && mel, "llx_mel_spectrogram: null pointer");
  LLX_REQUIRE(B > 0 && L > NFFT / 2 && n_frames > 0 && n_mels > 0 && n_mels <= 256, "llx_mel_spectrogram: bad sizes (L must exceed n_fft/2 for reflect padding)");
  hipLaunchKernelGGL(mel_power_kernel, dim3((unsigned)n_frames, (unsigned)B), dim3(256), 0, stream, audio, L, twiddle, window, fbank, mel,
                     (int)n_frames, (int)hop, (int)n_mels);
  LLX_LAUNCH_CHECK("llx_mel_spectrogram");
  return LLX_OK;
}

// feat: bf16 [B, n_frames + 1, n_mels] = one zero row, n_frames-1 feature rows, one zero row.
extern "C" int llx_logmel_cmn(const float* mel, void* feat, int64_t B, int64_t n_frames, int64_t n_mels, hipStream_t stream) {
  LLX_REQUIRE(mel && feat && B > 0 && n_frames > 1 && n_mels > 0, "llx_logmel_cmn: bad arguments");
  hipLaunchKernelGGL(logmel_cmn_kernel, dim3((unsigned)n_mels, (unsigned)B), dim3(256), 0, stream, mel, (bf16_t*)feat, (int)n_frames, (int)n_mels);
  LLX_LAUNCH_CHECK("llx_logmel_cmn");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ GELU (exact erf, nn.GELU default)
__global__ void gelu_fwd_kernel(const bf16_t* __restrict__ z, int64_t z_ld, bf16_t* __restrict__ y, int64_t y_ld, int64_t rows, int cols) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(z + r * z_ld + c);
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = bflo(v[e]), b = bfhi(v[e]);
    o[e] = pack_bf2(0.5f * a * (1.f + erff(a * 0.70710678118654752440f)), 0.5f * b * (1.f + erff(b * 0.70710678118654752440f)));
  }
  *reinterpret_cast<u32x4_t*>(y + r * y_ld + c) = o;
}

// dz = dy * (Phi(z) + z * phi(z))
__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dy, int64_t dy_ld, const bf16_t* __restrict__ z, int64_t z_ld, bf16_t* __restrict__ dz,
                                int64_t dz_ld, int64_t rows, int cols) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const u32x4_t g = *reinterpret_cast<const u32x4_t*>(dy + r * dy_ld + c);
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(z + r * z_ld + c);
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float res[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float a = p ? bfhi(v[e]) : bflo(v[e]);
      const float d = p ? bfhi(g[e]) : bflo(g[e]);
      const float cdf = 0.5f * (1.f + erff(a * 0.70710678118654752440f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * a * a);
      res[p] = d * (cdf + a * pdf);
    }
    o[e] = pack_bf2(res[0], res[1]);
  }
  *reinterpret_cast<u32x4_t*>(dz + r * dz_ld + c) = o;
}

extern "C" int llx_gelu_fwd(const void* z, int64_t z_ld, void* y, int64_t y_ld, int64_t rows, int64_t cols, hipStream_t stream) {
  LLX_REQUIRE(z && y && cols % 8 == 0 && z_ld % 8 == 0 && y_ld % 8 == 0, "llx_gelu_fwd: bad arguments");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)z, z_ld, (bf16_t*)y, y_ld, rows, (int)cols);
  LLX_LAUNCH_CHECK("llx_gelu_fwd");
  return LLX_OK;
}

extern "C" int llx_gelu_bwd(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, void* dz, int64_t dz_ld, int64_t rows, int64_t cols,
                            hipStream_t stream) {
  LLX_REQUIRE(dy && z && dz && cols % 8 == 0 && dy_ld % 8 == 0 && z_ld % 8 == 0 && dz_ld % 8 == 0, "llx_gelu_bwd: bad arguments");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)dy, dy_ld, (const bf16_t*)z, z_ld,
                     (bf16_t*)dz, dz_ld, rows, (int)cols);
  LLX_LAUNCH_CHECK("llx_gelu_bwd");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ conv1d(k=3) helpers
// Gradient of the padded time-major input from the im2col gradient: dpad[p][c] = sum_{kk} dA[(p-kk)/stride][kk*C + c]
// over kk in {0,1,2} with (p-kk) divisible by stride and the row in range.  dA: [M, 3C]; dpad: [P, C].
__global__ void col2im3_kernel(const bf16_t* __restrict__ dA, bf16_t* __restrict__ dpad, int M, int C, int P, int stride) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = C >> 3;
  if (idx >= (int64_t)P * cpr) return;
  const int p = (int)(idx / cpr), c = (int)(idx % cpr) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int q = p - kk;
    if (q < 0 || q % stride != 0) continue;
    const int l = q / stride;
    if (l >= M) continue;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dA + (int64_t)l * 3 * C + kk * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += bflo(v[e]); acc[2 * e + 1] += bfhi(v[e]); }
  }
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf2(acc[2 * e], acc[2 * e + 1]);
  *reinterpret_cast<u32x4_t*>(dpad + (int64_t)p * C + c) = o;
}

extern "C" int llx_col2im3(const void* dA, void* dpad, int64_t M, int64_t C, int64_t P, int64_t stride, hipStream_t stream) {
  LLX_REQUIRE(dA && dpad && C % 8 == 0 && M > 0 && P > 0 && stride > 0, "llx_col2im3: bad arguments");
  hipLaunchKernelGGL(col2im3_kernel, dim3((unsigned)cdiv64(P * (C / 8), 256)), dim3(256), 0, stream, (const bf16_t*)dA, (bf16_t*)dpad, (int)M,
                     (int)C, (int)P, (int)stride);
  LLX_LAUNCH_CHECK("llx_col2im3");
  return LLX_OK;
}

// Conv1d weight [D][C][3] <-> GEMM weight [D][3][C] (to_gemm = 1: w[d][c][k] -> out[d][k][c]; 0: the inverse).
__global__ void conv_w_reorder_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t D, int C, int to_gemm) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * C * 3) return;
  const int64_t d = idx / (3 * C);
  const int rem = (int)(idx % (3 * C));
  if (to_gemm) { const int k = rem / C, c = rem % C; out[idx] = in[(d * C + c) * 3 + k]; }
  else { const int c = rem / 3, k = rem % 3; out[idx] = in[(d * 3 + k) * C + c]; }
}

extern "C" int llx_conv_w_reorder(const void* in, void* out, int64_t D, int64_t C, int to_gemm, hipStream_t stream) {
  LLX_REQUIRE(in && out && D > 0 && C > 0, "llx_conv_w_reorder: bad arguments");
  hipLaunchKernelGGL(conv_w_reorder_kernel, dim3((unsigned)cdiv64(D * C * 3, 256)), dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, D, (int)C,
                     to_gemm);
  LLX_LAUNCH_CHECK("llx_conv_w_reorder");
  return LLX_OK;
}
