// Shared helpers for libhotk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hotk.h"

#define HOTK_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return HOTK_EINVAL; \
  } while (0)

static inline int hotk_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? HOTK_OK : HOTK_ELAUNCH;
}

// grid for a grid-stride loop over `work` items with `block` threads: enough blocks to fill
// 256 CUs x 8 blocks, never more than needed.
static inline unsigned hotk_grid(int64_t work, int block, int64_t cap = 256 * 16) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

static inline bool hotk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ bool hotk_keep(unsigned keep, uint8_t kind) { return (keep >> kind) & 1u; }
