// Mel front-end (reference preprocessing/utils.py:68-141): the HBM-bound pieces around the two
// contractions (windowed frames x DFT basis, magnitudes x mel basis), which run on the contraction kernel of gemm.hip.
#include "common.h"

namespace {

// frames[m][k] = win[k] * x[m*hop + k - left], zero outside the signal (lws pads fsize-hop samples on both sides and
// the tail up to a whole frame; utils.py:82-103).  One thread writes 4 consecutive k (16-byte stores).
__global__ void stft_frames_kernel(const float* __restrict__ wav, int64_t n, const float* __restrict__ win,
                                   float* __restrict__ frames, int M, int fsize, int hop, int left) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = fsize >> 2;
  if (idx >= (int64_t)M * per_row) return;
  const int m = (int)(idx / per_row), k = ((int)(idx - (int64_t)m * per_row)) << 2;
  const int64_t s = (int64_t)m * hop + k - left;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = s + e;
    v[e] = (i >= 0 && i < n) ? wav[i] * win[k + e] : 0.f;
  }
  *reinterpret_cast<f32x4*>(frames + (int64_t)m * fsize + k) = v;
}

// reim[row][0..nbp) = Re, [nbp..2nbp) = Im  ->  mag[row][j] = sqrt(Re^2 + Im^2)
__global__ void stft_magnitude_kernel(const float* __restrict__ reim, float* __restrict__ mag, int64_t rows, int nbp) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * nbp) return;
  const int64_t r = idx / nbp;
  const int j = (int)(idx - r * nbp);
  const float re = reim[r * 2 * nbp + j], im = reim[r * 2 * nbp + nbp + j];
  mag[idx] = sqrtf(re * re + im * im);
}

// out[c][col0 + m] = clip((20*log10(max(min_level, mel[m][c])) - ref_db - min_db) / -min_db, 0, 1)   (utils.py:127-137)
__global__ void mel_db_normalize_kernel(const float* __restrict__ mel, float* __restrict__ out, int M, int C,
                                        int64_t ld_out, int64_t col0, float min_level, float ref_db, float min_db) {
  __shared__ float tile[32][33];
  const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, c = c0 + tx;
    tile[r][tx] = (m < M && c < C) ? mel[(int64_t)m * C + c] : 1.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, m = m0 + tx;
    if (c < C && m < M) {
      const float db = 20.f * log10f(fmaxf(min_level, tile[tx][r])) - ref_db;
      out[(int64_t)c * ld_out + col0 + m] = fminf(fmaxf((db - min_db) / -min_db, 0.f), 1.f);
    }
  }
}

}  // namespace

DVAE_API int dvae_stft_frames(const float* wav, int64_t n, const float* window, float* frames, int M, int fsize,
                              int hop, int left, void* stream) {
  if (!wav || !window || !frames || n < 1 || M < 1 || fsize < 4 || (fsize & 3) || hop < 1 || left < 0) return DVAE_EINVAL;
  const int64_t work = (int64_t)M * (fsize >> 2);
  hipLaunchKernelGGL(stft_frames_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     wav, n, window, frames, M, fsize, hop, left);
  return dvae_check_launch();
}

DVAE_API int dvae_stft_magnitude(const float* reim, float* mag, int64_t rows, int nbins_padded, void* stream) {
  if (!reim || !mag || rows < 1 || nbins_padded < 1) return DVAE_EINVAL;
  const int64_t work = rows * nbins_padded;
  hipLaunchKernelGGL(stft_magnitude_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reim, mag, rows, nbins_padded);
  return dvae_check_launch();
}

DVAE_API int dvae_mel_db_normalize(const float* mel, float* out, int M, int n_mels, int64_t ld_out, int64_t col0,
                                   float min_level, float ref_level_db, float min_level_db, void* stream) {
  if (!mel || !out || M < 1 || n_mels < 1 || ld_out < M || col0 < 0 || !(min_level > 0.f) || !(min_level_db < 0.f))
    return DVAE_EINVAL;
  dim3 grid((M + 31) / 32, (n_mels + 31) / 32);
  hipLaunchKernelGGL(mel_db_normalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, out, M, n_mels, ld_out,
                     col0, min_level, ref_level_db, min_level_db);
  return dvae_check_launch();
}
