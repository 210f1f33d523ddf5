// Shared device/host helpers for libvorta_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_AS __attribute__((address_space(3)))

// records the failing hipError_t for vorta_last_hip_error() and returns VORTA_ELAUNCH
int vorta_set_hip_error(hipError_t e);

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
