// Device-resident MNIST input pipeline for gfx950 (K19/K20 of SURVEY.md §2.5).
// The reference builds every batch on the CPU with PIL (RandomRotation(5) -> RandomResizedCrop(D) -> ToTensor ->
// repeat to 3 channels -> Normalize; eval: Resize(D) -> CenterCrop(D) -> ..., /root/reference/dataloader.py:101-116)
// in 2 DataLoader workers per rank and ships 38.5 MB per step over PCIe (classif.py:43-44).  Here the whole uint8
// dataset (47 MB) lives in HBM and one kernel produces the normalised NHWC batch straight from the sample indices:
// only the indices (a few hundred bytes) cross the host link.  Per-sample augmentation parameters come from a
// Philox counter keyed by (seed, epoch, dataset index), so a sample's augmentation does not depend on the world
// size or on which rank draws it.
#include "common.h"

struct AugP {
  int B, S, D, Cpad;       // batch, source side (28), output side (224/299), padded channel count (4 fp32 / 8 bf16)
  int train;
  float mean, std;
  unsigned long long seed, epoch;
};

template <typename T>
__global__ void __launch_bounds__(256) augment_mnist_kernel(const unsigned char* __restrict__ images, const long long* __restrict__ labels,
                                                            const int* __restrict__ index, T* __restrict__ out,
                                                            long long* __restrict__ labels_out, AugP p) {
  __shared__ float prm[8];   // cos, sin, top, left, h, w
  __shared__ unsigned char img[64 * 64];
  const int b = blockIdx.x;
  const int sidx = index[b];
  const unsigned char* src = images + (size_t)sidx * p.S * p.S;
  for (int i = threadIdx.x; i < p.S * p.S; i += blockDim.x) img[i] = src[i];
  if (threadIdx.x == 0) {
    float ang = 0.f, top = 0.f, left = 0.f, h = (float)p.S, w = (float)p.S;
    if (p.train) {
      const Philox ph(p.seed);
      uint32_t r[4];
      ph.gen((unsigned long long)sidx * 16ull, p.epoch, r);
      ang = (u32_to_unit(r[0]) * 10.f - 5.f) * 0.017453292519943295f;
      // torchvision RandomResizedCrop.get_params: scale (0.08, 1), log-uniform ratio (3/4, 4/3), 10 attempts
      const float area = (float)(p.S * p.S);
      const float lr0 = -0.2876820724517809f, lr1 = 0.2876820724517809f;
      bool found = false;
      for (int a = 0; a < 10 && !found; ++a) {
        ph.gen((unsigned long long)sidx * 16ull + 1 + a, p.epoch, r);
        const float ta = area * (0.08f + 0.92f * u32_to_unit(r[0]));
        const float ar = __expf(lr0 + (lr1 - lr0) * u32_to_unit(r[1]));
        const int cw = (int)rintf(sqrtf(ta * ar)), chh = (int)rintf(sqrtf(ta / ar));
        if (cw > 0 && cw <= p.S && chh > 0 && chh <= p.S) {
          top = (float)(int)(u32_to_unit(r[2]) * (float)(p.S - chh + 1));
          left = (float)(int)(u32_to_unit(r[3]) * (float)(p.S - cw + 1));
          h = (float)chh; w = (float)cw; found = true;
        }
      }
      // fallback: whole image (in_ratio == 1 lies inside [3/4, 4/3])
    }
    prm[0] = cosf(ang); prm[1] = sinf(ang); prm[2] = top; prm[3] = left; prm[4] = h; prm[5] = w;
    if (blockIdx.y == 0) labels_out[b] = labels[sidx];
  }
  __syncthreads();
  const float ca = prm[0], sa = prm[1], top = prm[2], left = prm[3], ch = prm[4], cw = prm[5];
  const float ctr = 0.5f * (float)p.S;
  const float inv_std = 1.0f / p.std;
  const bool rot = p.train != 0;
  auto rotated = [&](int yy, int xx) -> float {   // pixel (yy,xx) of the rotated 28x28 image, nearest, fill 0
    if (!rot) return (float)img[yy * p.S + xx];
    const float dx = (float)xx + 0.5f - ctr, dy = (float)yy + 0.5f - ctr;
    const int xs = (int)floorf(ca * dx - sa * dy + ctr), ys = (int)floorf(sa * dx + ca * dy + ctr);
    if ((unsigned)xs >= (unsigned)p.S || (unsigned)ys >= (unsigned)p.S) return 0.f;
    return (float)img[ys * p.S + xs];
  };
  const int rows_per_block = (p.D + gridDim.y - 1) / gridDim.y;
  const int y_begin = blockIdx.y * rows_per_block;
  const int y_end = min(p.D, y_begin + rows_per_block);
  const int npix = (y_end - y_begin) * p.D;
  constexpr int VEC = 16 / sizeof(T);
  for (int i = threadIdx.x; i < npix; i += blockDim.x) {
    const int oy = y_begin + i / p.D, ox = i % p.D;
    const float sy = ((float)oy + 0.5f) * ch / (float)p.D - 0.5f;
    const float sx = ((float)ox + 0.5f) * cw / (float)p.D - 0.5f;
    const float fy0 = floorf(sy), fx0 = floorf(sx);
    const float wy = sy - fy0, wx = sx - fx0;
    const int y0 = (int)fminf(fmaxf(fy0, 0.f), ch - 1.f) + (int)top, y1 = (int)fminf(fmaxf(fy0 + 1.f, 0.f), ch - 1.f) + (int)top;
    const int x0 = (int)fminf(fmaxf(fx0, 0.f), cw - 1.f) + (int)left, x1 = (int)fminf(fmaxf(fx0 + 1.f, 0.f), cw - 1.f) + (int)left;
    const float v = (1.f - wy) * ((1.f - wx) * rotated(y0, x0) + wx * rotated(y0, x1)) +
                    wy * ((1.f - wx) * rotated(y1, x0) + wx * rotated(y1, x1));
    const float nv = (v * (1.0f / 255.0f) - p.mean) * inv_std;
    T* o = out + ((size_t)(b * p.D + oy) * p.D + ox) * p.Cpad;
    for (int c0 = 0; c0 < p.Cpad; c0 += VEC) {
      Vec16<T> vec;
#pragma unroll
      for (int e = 0; e < VEC; ++e) vec.set(e, (c0 + e < 3) ? nv : 0.f);
      vec.store(o + c0);
    }
  }
}

DPA_API int dpa_augment_mnist(const unsigned char* images, const long long* labels, const int* index, void* out,
                               long long* labels_out, const AugP* pp, int dtype, hipStream_t s) {
  AugP p = *pp;
  int ychunks = 2048 / (p.B > 0 ? p.B : 1);
  if (ychunks < 1) ychunks = 1;
  if (ychunks > p.D / 8) ychunks = p.D / 8 > 0 ? p.D / 8 : 1;
  dim3 grid(p.B, ychunks);
  if (dtype == 1) hipLaunchKernelGGL(augment_mnist_kernel<bf16_t>, grid, dim3(256), 0, s, images, labels, index, (bf16_t*)out, labels_out, p);
  else            hipLaunchKernelGGL(augment_mnist_kernel<float>, grid, dim3(256), 0, s, images, labels, index, (float*)out, labels_out, p);
  return dpa_launch_rc();
}
DPA_API int dpa_aug_params_size() { return (int)sizeof(AugP); }
