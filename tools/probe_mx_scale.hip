// Which lane's scale byte multiplies which 32-element block of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3)?  (round 5: the
// probabilities of the int8-score attention kernel get one power-of-two scale per lane = per (query row, 32 keys of the block).)
// A = B = 1.0 everywhere, scale_a = 2^0, scale_b[lane] = 2^t[lane] with t = lane % 7 - 3: if lane l scales the 32 k-values it
// holds itself, out[i][j] = 32 (2^t[j] + 2^t[j + 32]) for every row i.  Also: opsel picks byte 0..3 of the scale register.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx_scale.hip -o /tmp/probe_mx && /tmp/probe_mx
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OPSEL>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }  // e4m3 1.0
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int t = lane % 7 - 3;
  const int sb = ((127 + t) & 0xff) << (8 * OPSEL) | (OPSEL == 0 ? 0x11223300 : 0x00000011);  // junk in the other bytes
  const int sa = 127 << (8 * OPSEL);
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, sa, OPSEL, sb);
  for (int i = 0; i < 16; ++i) out[(OPSEL * 64 + lane) * 16 + i] = c[i];
}

int main() {
  float* d;
  hipMalloc(&d, 4 * 64 * 16 * sizeof(float));
  hipLaunchKernelGGL((k<0>), dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL((k<1>), dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL((k<2>), dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL((k<3>), dim3(1), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  static float h[4 * 64 * 16];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int op = 0; op < 4; ++op) {
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 16; ++i) {
        const int j = lane & 31;  // C/D: column on the lane
        const double want = 32.0 * (exp2((double)(j % 7 - 3)) + exp2((double)((j + 32) % 7 - 3)));
        if (fabs(h[(op * 64 + lane) * 16 + i] - want) > 1e-3 * want) ++bad;
      }
    printf("opsel %d: %d of 1024 outputs differ from 32 (2^t[col] + 2^t[col + 32])   e.g. out[0][0..3] = %g %g %g %g (want %g %g %g %g)\n", op, bad,
           h[(op * 64 + 0) * 16], h[(op * 64 + 1) * 16], h[(op * 64 + 2) * 16], h[(op * 64 + 3) * 16],
           32.0 * (exp2(-3.0) + exp2((32 % 7) - 3.0)), 32.0 * (exp2(-2.0) + exp2((33 % 7) - 3.0)), 32.0 * (exp2(-1.0) + exp2((34 % 7) - 3.0)),
           32.0 * (exp2(0.0) + exp2((35 % 7) - 3.0)));
  }
  return 0;
}
