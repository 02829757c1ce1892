// Device-side image augmentation on uint8 NCHW batches (gfx950), fused with the batch gather of the
// HBM-resident dataset cache:
//
//   out[n][c][h][w] = lut[n][ src[ sel[n] ][c][ geom_n^-1(h, w) ] ]
//
// * geometry per sample (packed in geom[n]): k quarter turns counter-clockwise (np.rot90 / PIL
//   rotate(90k), square images), then horizontal flip, then vertical flip — the order of the
//   reference's albumentations pipeline (dataset/example_dataset.py:37-39: RandomRotate90,
//   HorizontalFlip, VerticalFlip);
// * photometric per sample: ONE 256-entry table lut[n] that the host composes from
//   RandomBrightnessContrast and RandomGamma (example_dataset.py:43-44) with the very numpy
//   expressions of data/image_folder.py — so the device result is bit-identical to the CPU pipeline
//   for these five transforms (a powf() on the device would differ from numpy's in the last ulp
//   and flip truncations); the table lives in LDS (256 B per sample);
// * sel (optional, int64): the gather of this batch from the un-augmented cache — the augmented batch
//   is written once, nothing else is materialised.
//
// Blur / median blur / CLAHE / JPEG round trip of the reference's pipeline stay on the CPU path
// (data/image_folder.py); the device pipeline does not apply them (documented in docs/KERNELS.md).
//
// Memory-bound and tiny next to the step it feeds (154 MB in + out per 1024-image batch, ~0.1 ms):
// one workgroup per (sample, channel, 64-row band); a thread produces 4 consecutive output pixels
// (one 4-byte store).  Quarter turns read the source column-wise — byte gathers that L2 absorbs
// (every fetched 128-byte line is used by 128 consecutive output rows of the same workgroup column).
#include "dtp_common.h"

__global__ __launch_bounds__(256) void augment_u8_kernel(const unsigned char* __restrict__ src,
                                                         const long long* __restrict__ sel,
                                                         const int* __restrict__ geom,
                                                         const unsigned char* __restrict__ lut,
                                                         unsigned char* __restrict__ out, int C, int H, int W,
                                                         int bands) {
  __shared__ unsigned char tab[256];
  const int band = blockIdx.x % bands;
  const int c = (blockIdx.x / bands) % C;
  const int n = blockIdx.x / (bands * C);
  tab[threadIdx.x] = lut ? lut[(size_t)n * 256 + threadIdx.x] : (unsigned char)threadIdx.x;
  __syncthreads();
  const int g = geom ? geom[n] : 0;
  const int k = g & 3, hflip = (g >> 2) & 1, vflip = (g >> 3) & 1;
  const size_t plane = (size_t)H * W;
  const unsigned char* s = src + ((size_t)(sel ? sel[n] : n) * C + c) * plane;
  unsigned char* o = out + ((size_t)n * C + c) * plane;
  const int wq = (W + 3) >> 2;                      // 4-pixel groups per row
  const int h0 = band * 64, h1 = h0 + 64 < H ? h0 + 64 : H;
  for (int i = threadIdx.x; i < (h1 - h0) * wq; i += 256) {
    const int h = h0 + i / wq, w4 = (i % wq) * 4;
    unsigned v = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int w = w4 + e;
      if (w >= W) break;
      // undo the flips (applied last), then the rotation: rot90(a, 1)[i][j] = a[j][S-1-i]
      const int hh = vflip ? H - 1 - h : h;
      const int ww = hflip ? W - 1 - w : w;
      int sh, sw;
      if (k == 0) { sh = hh; sw = ww; }
      else if (k == 1) { sh = ww; sw = W - 1 - hh; }
      else if (k == 2) { sh = H - 1 - hh; sw = W - 1 - ww; }
      else { sh = H - 1 - ww; sw = hh; }
      v |= (unsigned)tab[s[(size_t)sh * W + sw]] << (8 * e);
    }
    if (w4 + 4 <= W && ((W & 3) == 0)) *(unsigned*)(o + (size_t)h * W + w4) = v;
    else {
      for (int e = 0; e < 4 && w4 + e < W; ++e) o[(size_t)h * W + w4 + e] = (unsigned char)(v >> (8 * e));
    }
  }
}

// src: [n_src][C][H][W] u8; sel: optional [N] int64 indices into src (null: sample n reads src[n]);
// geom: optional [N] int32 (bits 0-1 quarter turns, bit 2 h-flip, bit 3 v-flip; turns need H == W);
// lut: optional [N][256] u8; out: [N][C][H][W] u8 (must not alias src).
extern "C" int dtp_augment_u8(const void* src, const void* sel, const void* geom, const void* lut, void* out, int N,
                              int C, int H, int W, int square_checked, hipStream_t stream) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || !src || !out || src == out) return -10;
  if (geom && H != W && !square_checked) return -11;      // caller promises no quarter turns on non-square images
  const int bands = (H + 63) / 64;
  const long long blocks = (long long)N * C * bands;
  if (blocks > 0x7FFFFFFFll) return -12;
  hipLaunchKernelGGL(augment_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned char*)src,
                     (const long long*)sel, (const int*)geom, (const unsigned char*)lut, (unsigned char*)out, C, H, W,
                     bands);
  return DTP_CHECK_LAUNCH();
}
