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

// second question: WHICH lane's byte scales k-block 0 (keys 0-31) of column j, lane j or lane j + 32?  A = 1 only in k-block
// kb (the lanes of half kb), B = 1 everywhere: out[i][j] = 32 * 2^t[lane that scales (k-block kb, column j)].  FP4: the A
// operand in e2m1 (cbsz 4, as the attention kernels' row-sum MFMA)
template <int KB, bool FP4>
__global__ void k2(float* out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (lane >> 5) == KB ? (FP4 ? 0x22222222 : 0x38383838) : 0; b[i] = 0x38383838; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int t = lane % 7 - 3;
  if (FP4) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 0, 0, 127, 0, 127 + t);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127 + t);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

// third question (8-bit operands): the 32 bytes of a lane are NOT one scale block.  A = 1 in bytes 16 s ... 16 s + 15 of EVERY lane,
// 0 elsewhere: if byte j of lane half h is element 16 h + (j & 15) of scale block j >> 4, out[i][j] = 32 * 2^t[j + 32 s].
template <int SEL>
__global__ void k3(float* out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (i >> 2) == SEL ? 0x38383838 : 0; b[i] = 0x38383838; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int t = lane % 7 - 3;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127 + t);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

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
  for (int v = 0; v < 4; ++v) {
    if (v == 0) hipLaunchKernelGGL((k2<0, false>), dim3(1), dim3(64), 0, 0, d);
    if (v == 1) hipLaunchKernelGGL((k2<1, false>), dim3(1), dim3(64), 0, 0, d);
    if (v == 2) hipLaunchKernelGGL((k2<0, true>), dim3(1), dim3(64), 0, 0, d);
    if (v == 3) hipLaunchKernelGGL((k2<1, true>), dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64 * 16 * sizeof(float), hipMemcpyDeviceToHost);
    int own = 0, other = 0;
    const int kb = v & 1;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 16; ++i) {
        const int j = lane & 31;
        const double w_own = 32.0 * exp2((double)((j + 32 * kb) % 7 - 3)), w_other = 32.0 * exp2((double)((j + 32 * (1 - kb)) % 7 - 3));
        if (fabs(h[lane * 16 + i] - w_own) <= 1e-3 * w_own) ++own;
        if (fabs(h[lane * 16 + i] - w_other) <= 1e-3 * w_other) ++other;
      }
    printf("A %s, nonzero in k-block %d only: %d of 1024 outputs = 32 * 2^t[lane j + 32 * %d] (the lanes that HOLD that block of B), %d = the other half's\n",
           v >= 2 ? "e2m1" : "e4m3", kb, own, kb, other);
  }
  for (int sel = 0; sel < 2; ++sel) {
    if (sel == 0) hipLaunchKernelGGL((k3<0>), dim3(1), dim3(64), 0, 0, d);
    else hipLaunchKernelGGL((k3<1>), dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64 * 16 * sizeof(float), hipMemcpyDeviceToHost);
    int ok = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 16; ++i) {
        const double w = 32.0 * exp2((double)(((lane & 31) + 32 * sel) % 7 - 3));
        if (fabs(h[lane * 16 + i] - w) <= 1e-3 * w) ++ok;
      }
    printf("e4m3 operands, A nonzero in bytes %d-%d of every lane: %d of 1024 outputs = 32 * 2^t[column + %d]: scale block %d = those bytes of BOTH lane halves, its scale from lane half %d\n",
           16 * sel, 16 * sel + 15, ok, 32 * sel, sel, sel);
  }
  return 0;
}
