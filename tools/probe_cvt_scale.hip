// What does v_cvt_scalef32_pk_fp8_f32 do with its scale operand?  (round 5: the mixed kernel's probabilities get one power-of-two
// scale per (query row, 32 keys); if the conversion divides by the scale, P' / 2^e costs no instruction of its own.)
// Prints e4m3(x, scale) for a few x and scales, as decoded values.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_cvt_scale.hip -o /tmp/probe_cvt && /tmp/probe_cvt
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(2))) short s2;

__global__ void k(const float* x, const float* sc, int nx, int ns, int* out) {
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ns; ++j) {
      s2 old = {0x1234, 0x5678};
      s2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], -x[i], sc[j], false);
      s2 hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], -x[i], sc[j], true);
      out[(i * ns + j) * 2] = *(int*)&lo;
      out[(i * ns + j) * 2 + 1] = *(int*)&hi;
    }
}

static double dec(int b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  double v = e == 0 ? m * pow(2.0, -9) : (e == 15 && m == 7 ? NAN : (1 + m / 8.0) * pow(2.0, e - 7));
  return s ? -v : v;
}

int main() {
  const float x[] = {1.f, 3.f, 448.f, 1000.f, 0.001f, 100.f, 1e10f};
  const float sc[] = {1.f, 2.f, 4.f, 0.5f, 1024.f, 3.f, 0.f};
  const int nx = 7, ns = 7;
  float *dx, *ds; int* dout;
  hipMalloc(&dx, sizeof x); hipMalloc(&ds, sizeof sc); hipMalloc(&dout, nx * ns * 2 * sizeof(int));
  hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice); hipMemcpy(ds, sc, sizeof sc, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, dx, ds, nx, ns, dout);
  int h[7 * 7 * 2];
  hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ns; ++j) {
      const int lo = h[(i * ns + j) * 2], hi = h[(i * ns + j) * 2 + 1];
      printf("x = %-8g scale = %-6g : word(opsel 0) %08x -> bytes %g, %g | word(opsel 1) %08x -> bytes %g, %g   (x / scale = %g)\n", x[i], sc[j],
             lo, dec(lo & 255), dec((lo >> 8) & 255), hi, dec((hi >> 16) & 255), dec((hi >> 24) & 255), x[i] / sc[j]);
    }
  return 0;
}
