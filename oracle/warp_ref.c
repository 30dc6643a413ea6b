/* TEST INFRASTRUCTURE ONLY (see oracle/tecogan_oracle.py): plain-C restatement of the two pieces of the path whose
 * results are gated BIT-EXACT - the x4 bilinear upsample that turns an LR frame into the pseudo-flow
 * (code/ops.py:98-100 = nn.Upsample(scale_factor=4, mode='bilinear'), used at code/train.py:71-77) and the corner
 * indices / weights of F.grid_sample(bilinear, zeros, align_corners=False) (code/train.py:81-84,98,165,187;
 * main.py:203).  The arithmetic follows ATen's CPU kernels (UpSampleKernel.cpp area_pixel_compute_source_index,
 * GridSamplerKernel.cpp grid_sampler_unnormalize / compute_interp_params) operation by operation in IEEE fp32 with no
 * contraction (-ffp-contract=off), except where ATen itself uses an fma (noted).  Pinned by tests/test_oracle_golden.py
 * against the fixtures produced by the real reference and against torch on random data.
 *   gcc -O2 -ffp-contract=off -shared -fPIC -o libwarp_ref.so warp_ref.c -lm                                         */
#include <math.h>
#include <stdint.h>

/* source index and weight of output coordinate d for scale 1/4, align_corners=False: src = 0.25*(d+0.5)-0.5, clamped at 0 */
static void up4_coord(int d, int in_size, int* i0, int* i1, float* l1) {
  float s = 0.25f * ((float)d + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  *i0 = (int)s;
  *i1 = *i0 + ((*i0 < in_size - 1) ? 1 : 0);
  *l1 = s - (float)*i0;
}

/* out[4h][4w] = post_a * bilinear_x4(pre * in[h][w]) + post_b.  ATen evaluates each lerp as fmaf(w0, v0, w1*v1)
 * (measured bit-exact against torch 2.10 CPU in the build container). */
void warp_ref_up4(const float* in, int h, int w, float* out, float pre, float post_a, float post_b) {
  const int H = 4 * h, W = 4 * w;
  for (int Y = 0; Y < H; ++Y) {
    int y0, y1;
    float ly;
    up4_coord(Y, h, &y0, &y1, &ly);
    const float hy = 1.f - ly;
    for (int X = 0; X < W; ++X) {
      int x0, x1;
      float lx;
      up4_coord(X, w, &x0, &x1, &lx);
      const float hx = 1.f - lx;
      const float a = in[y0 * w + x0] * pre, b = in[y0 * w + x1] * pre;
      const float c = in[y1 * w + x0] * pre, d = in[y1 * w + x1] * pre;
      const float top = fmaf(hx, a, lx * b);
      const float bot = fmaf(hx, c, lx * d);
      const float v = fmaf(hy, top, ly * bot);
      out[Y * W + X] = post_a * v + post_b;
    }
  }
}

/* round to nearest-even fp16 and back (the `.half()` of code/train.py:98,187) without relying on _Float16 */
static float fp16_round(float f) {
  union { float f; uint32_t u; } v = {f};
  const uint32_t sign = v.u & 0x80000000u;
  uint32_t a = v.u & 0x7fffffffu;
  if (a >= 0x7f800000u) return f;                  /* inf / nan */
  if (a >= 0x477ff000u) {                          /* >= 65520: rounds to inf */
    v.u = sign | 0x7f800000u;
    return v.f;
  }
  if (a < 0x38800000u) {                           /* below the smallest normal half: quantum 2^-24 */
    const float q = 5.9604644775390625e-08f;       /* 2^-24 */
    float r = nearbyintf(fabsf(f) / q) * q;        /* default rounding mode: to nearest even */
    return sign ? -r : r;
  }
  const uint32_t lsb = (a >> 13) & 1u;             /* keep 10 mantissa bits */
  a += 0xfffu + lsb;
  a &= ~0x1fffu;
  v.u = sign | a;
  return v.f;
}

/* grid [N][H][W][2] (x, y in [-1,1] units, arbitrary values allowed) sampled on an image of IH x IW:
 * corner[n][y][x] = {x0, y0} (floor indices, clamped to [-2, size+1] so that far-away coordinates stay "outside"),
 * weights[n][y][x] = {nw, ne, sw, se}.  half_grid != 0: the grid is rounded to fp16 first. */
void warp_ref_corners(const float* grid, long n_px, int IH, int IW, int half_grid, int32_t* corner, float* weights) {
  for (long i = 0; i < n_px; ++i) {
    float gx = grid[2 * i], gy = grid[2 * i + 1];
    if (half_grid) {
      gx = fp16_round(gx);
      gy = fp16_round(gy);
    }
    const float ix = ((gx + 1.f) * (float)IW - 1.f) * 0.5f;   /* grid_sampler_unnormalize, align_corners=False */
    const float iy = ((gy + 1.f) * (float)IH - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const float cx = fminf(fmaxf(fx, -2.f), (float)IW + 1.f), cy = fminf(fmaxf(fy, -2.f), (float)IH + 1.f);
    corner[2 * i] = (int32_t)cx;
    corner[2 * i + 1] = (int32_t)cy;
    if (weights) {
      const float tx = ix - fx, ty = iy - fy;                  /* "east"/"south" weights */
      weights[4 * i + 0] = (1.f - tx) * (1.f - ty);
      weights[4 * i + 1] = tx * (1.f - ty);
      weights[4 * i + 2] = (1.f - tx) * ty;
      weights[4 * i + 3] = tx * ty;
    }
  }
}

/* bilinear sample with zero padding from the corners above: out[c][px] for one image [C][IH][IW] */
void warp_ref_sample(const float* img, int C, int IH, int IW, const int32_t* corner, const float* weights, long n_px,
                     float* out) {
  for (long i = 0; i < n_px; ++i) {
    const int x0 = corner[2 * i], y0 = corner[2 * i + 1];
    for (int c = 0; c < C; ++c) {
      const float* pl = img + (long)c * IH * IW;
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        if (x >= 0 && x < IW && y >= 0 && y < IH) acc += pl[y * IW + x] * weights[4 * i + k];
      }
      out[(long)c * n_px + i] = acc;
    }
  }
}
