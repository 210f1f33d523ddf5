// Hardware probe for the fp8 attention path (gfx950): prints what the kernels rely on.
//   1. ds_read_b64_tr_b8: which LDS byte every lane receives, for lane-linear addresses
//   2. v_mfma_f32_32x32x64_f8f6f4: the k index of byte j of lane half h, for A and for B
//   3. v_cvt_pk_fp8_f32: rounding / saturation of out-of-range and tiny values
//   4. issue rate of the fp8 (plain and block-scaled) and bf16 32x32 MFMAs
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe_fp8.hip -o /tmp/probe_fp8 && /tmp/probe_fp8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define LDS_AS __attribute__((address_space(3)))

__global__ void tr8_kernel(uint16_t* out, int stride) {
  __shared__ uint16_t sm16[4096];  // unused layout helper
  __shared__ unsigned char lo[8192], hi[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) { lo[i] = i & 255; hi[i] = i >> 8; }
  __syncthreads();
  const int addr = threadIdx.x * stride;
  i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_AS i32x2*)(lo + addr));
  i32x2 b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_AS i32x2*)(hi + addr));
  unsigned char ab[8], bb[8];
  memcpy(ab, &a, 8); memcpy(bb, &b, 8);
  for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = (uint16_t)(ab[j] | (bb[j] << 8));
  (void)sm16;
}

// one-hot A (row 0, half h, byte j) against B column 0 holding slot ids -> C[0][0] names the matching B slot
__global__ void mfma_map_kernel(float* out) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int run = 0; run < 2; ++run)
    for (int slot = 0; slot < 64; ++slot) {
      unsigned char a[32], b[32];
      for (int j = 0; j < 32; ++j) { a[j] = 0; b[j] = 0; }
      if (r == 0 && h == slot / 32) a[slot % 32] = 0x38;  // 1.0 in e4m3 (bias 7: exponent 7 -> 0x38)
      if (r == 0) {
        for (int j = 0; j < 32; ++j) {
          const int s = h * 32 + j;
          const int v = run == 0 ? (s & 7) + 1 : (s >> 3) + 1;  // 1..8
          // e4m3 encodings of 1..8
          const unsigned char enc[9] = {0, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
          b[j] = enc[v];
        }
      }
      i32x8 av, bv;
      memcpy(&av, a, 32); memcpy(&bv, b, 32);
      f32x16 c;
      for (int i = 0; i < 16; ++i) c[i] = 0.f;
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0, 0, 0);
      if (lane == 0) out[run * 64 + slot] = c[0];
    }
}

// full check of the assumed layout with asymmetric integer data: A[m][k], B[k][n]; lane (r,h) byte j <-> k = 32h + j
__global__ void mfma_full_kernel(const unsigned char* A, const unsigned char* B, float* C) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  unsigned char a[32], b[32];
  for (int j = 0; j < 32; ++j) { a[j] = A[r * 64 + 32 * h + j]; b[j] = B[(32 * h + j) * 32 + r]; }
  i32x8 av, bv;
  memcpy(&av, a, 32); memcpy(&bv, b, 32);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0, 0, 0);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = c[i];
  }
}

__global__ void cvt_kernel(const float* x, int n, unsigned char* o) {
  for (int i = 0; i < n; i += 2) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i + 1], r, false);
    o[i] = r & 255; o[i + 1] = (r >> 8) & 255;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(long long* out, int iters, int sa, int sb) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b[i] = 0x3c3c3c3c ^ (threadIdx.x << (i & 3)); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(1.0f + threadIdx.x * 0.01f); hb[i] = (__bf16)(0.5f + i * 0.1f); }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 0, 0, 0, 0, 0, 0);
      } else if (MODE == 1) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, sa, 0, sb);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 0, 0, 0, sa, 0, sb);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, ha, c1, 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}

static float e4m3_to_float(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

int main() {
  // ---- 1. tr8 ----
  for (int stride : {8, 16}) {
    uint16_t* d; hipMalloc(&d, 64 * 8 * 2);
    tr8_kernel<<<1, 64>>>(d, stride);
    uint16_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b8, lane address = lane*%d: source byte address of result byte j ==\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) printf(" %4d", h[l * 8 + j]);
      printf("   (src lane:");
      for (int j = 0; j < 8; ++j) printf(" %2d.%d", h[l * 8 + j] / stride, h[l * 8 + j] % stride);
      printf(")\n");
    }
    hipFree(d);
  }
  // ---- 2. MFMA k map ----
  {
    float* d; hipMalloc(&d, 128 * 4);
    mfma_map_kernel<<<1, 64>>>(d);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== v_mfma_f32_32x32x64_f8f6f4: A slot (h,j) pairs with B slot ==\n");
    int ident = 1;
    for (int s = 0; s < 64; ++s) {
      const int lo = (int)h[s] - 1, hi = (int)h[64 + s] - 1, bs = lo + 8 * hi;
      if (bs != s) ident = 0;
      printf("A(%d,%2d)->B(%d,%2d)%s", s / 32, s % 32, bs / 32, bs % 32, (s % 4 == 3) ? "\n" : "  ");
    }
    printf("A/B k maps identical: %s\n", ident ? "YES" : "NO");
    hipFree(d);
    unsigned char A[32 * 64], B[64 * 32];
    const unsigned char enc[9] = {0, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
    int Ai[32 * 64], Bi[64 * 32];
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 64; ++k) { Ai[m * 64 + k] = (m * 3 + k * 5) % 7; A[m * 64 + k] = enc[Ai[m * 64 + k]]; }
    for (int k = 0; k < 64; ++k) for (int n = 0; n < 32; ++n) { Bi[k * 32 + n] = (k * 2 + n * 3 + 1) % 9; B[k * 32 + n] = enc[Bi[k * 32 + n]]; }
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    mfma_full_kernel<<<1, 64>>>(dA, dB, dC);
    float C[32 * 32]; hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += Ai[m * 64 + k] * Bi[k * 32 + n];
      if ((int)C[m * 32 + n] != ref) ++bad;
    }
    printf("full 32x32x64 check with k = 32h + j, C row=(i&3)+8(i>>2)+4h col=lane&31: %d mismatches\n", bad);
  }
  // ---- 3. cvt ----
  {
    float x[] = {448.f, 449.f, 464.f, 465.f, 480.f, 500.f, 1e6f, INFINITY, NAN, -500.f, 0.001953125f, 0.0009765625f,
                 0.00146484375f, 0.0009f, 256.f, 272.f, 0.017f, 1.0625f, 1.1875f, 63.9f, 3.75f, 3.76f};
    const int n = sizeof(x) / 4;
    float* dx; unsigned char* dout;
    hipMalloc(&dx, sizeof(x)); hipMalloc(&dout, n);
    hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice);
    cvt_kernel<<<1, 1>>>(dx, n, dout);
    unsigned char o[64]; hipMemcpy(o, dout, n, hipMemcpyDeviceToHost);
    printf("== v_cvt_pk_fp8_f32 ==\n");
    for (int i = 0; i < n; ++i) printf("%12g -> 0x%02x = %g\n", x[i], o[i], e4m3_to_float(o[i]));
  }
  // ---- 4. rates ----
  {
    long long* d; hipMalloc(&d, 16);
    long long h[2];
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      rate_kernel<0><<<1, 256>>>(d, iters, 127, 127); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("fp8 32x32x64 plain : %.1f cycles per MFMA per SIMD\n", (double)h[0] / (iters * 16));
      rate_kernel<1><<<1, 256>>>(d, iters, 127, 127); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("fp8 32x32x64 scaled: %.1f cycles per MFMA per SIMD\n", (double)h[0] / (iters * 16));
      rate_kernel<2><<<1, 256>>>(d, iters, 127, 127); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("bf16 32x32x16      : %.1f cycles per MFMA per SIMD\n", (double)h[0] / (iters * 16));
    }
  }
  return 0;
}
