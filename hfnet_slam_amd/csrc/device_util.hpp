// device_util.hpp -- small device-side helpers shared by the kernel files (included inside namespace hfnet, after f32x4)
#pragma once

// base + uniform byte offset, pinned to scalar registers: a load through it is "scalar base + 32-bit lane offset" in the
// GLOBAL address space and costs no vector instruction for its address (left alone the compiler folds the uniform part
// into 64-bit vector adds; through a pointer that came out of a struct in memory it even issues FLAT loads)
typedef const __attribute__((address_space(1))) char* gbase_t;
typedef const __attribute__((address_space(1))) f32x4* gvec4_t;
typedef const __attribute__((address_space(1))) float* gf32_t;
__device__ __forceinline__ gbase_t sgpr_base(const void* base, unsigned uniform_bytes) {
    gbase_t p = (gbase_t)(const char*)base + uniform_bytes;
    asm("" : "+s"(p));
    return p;
}
// a lane offset re-"defined" where it is used: hoisted out of a loop it is widened to 64 bits once and every load through it
// then pays a 64-bit vector add instead of using its scalar-base + 32-bit-offset form
__device__ __forceinline__ unsigned fresh(unsigned v) { asm volatile("" : "+v"(v)); return v; }
