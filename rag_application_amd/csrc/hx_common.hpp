// Shared host/device helpers for libhx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <stdexcept>

namespace hx {

// ---- error plumbing ---------------------------------------------------------
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void set_last_error(const std::string& s);

#define HX_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      throw hx::Error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +      \
                      __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
  } while (0)

#define HX_CHECK(cond, msg)                                                           \
  do {                                                                                \
    if (!(cond)) throw hx::Error(std::string(msg));                                   \
  } while (0)

// ---- total order: (score desc, id asc) as one descending u64 -----------------
__host__ __device__ inline uint32_t f32_orderable(float s) {
  uint32_t u;
  __builtin_memcpy(&u, &s, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float orderable_f32(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float s;
  __builtin_memcpy(&s, &u, 4);
  return s;
}
__host__ __device__ inline uint64_t make_key(float score, uint32_t id) {
  return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
__host__ __device__ inline float key_score(uint64_t k) { return orderable_f32((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_id(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)k; }

// ---- synthetic-data hash (oracle/oracle.py hash2) -----------------------------
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t hash2(uint32_t seed, uint32_t a, uint32_t b) {
  uint32_t h = fmix32(seed + a * 0x9E3779B1u);
  return fmix32(h ^ (b * 0x85EBCA77u));
}
__host__ __device__ inline float synth_value(uint32_t seed, uint32_t r, uint32_t c) {
  int32_t h = (int32_t)hash2(seed, r, c);
  return (float)(h >> 8) * 1.1920928955078125e-07f;  // 2^-23, exact
}

constexpr int WAVE = 64;
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Capacity of one per-query candidate buffer (keys).  Bitonic-sortable in LDS.
constexpr int CAND_CAP = 8192;
// Largest list a stage may be asked to keep.
constexpr int MAX_LIMIT = 2048;

}  // namespace hx
