// common.h -- error plumbing shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/orbslam3_hip.h"

namespace osh {

// thread-local text returned by osh_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

#define OSH_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      osh::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return OSH_ERR_DEVICE;                                                               \
    }                                                                                      \
  } while (0)

// Simple growable device buffer (never shrinks; reused across batches).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }   // owners that live in thread_local storage give their memory back when the thread exits
  int reserve(size_t bytes) {
    if (bytes <= cap) return OSH_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    OSH_HIP(hipMalloc(&p, want));
    cap = want;
    return OSH_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host buffer (grow-only).
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { release(); }
  void* reserve(size_t bytes) {
    if (bytes <= cap) return p;
    release();
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&p, want) != hipSuccess) { p = nullptr; cap = 0; return nullptr; }
    cap = want;
    return p;
  }
  void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
};

// Per-kernel HIP-event timing on one stream.
struct KernelTimer {
  static constexpr int kMaxPending = 4096;
  hipEvent_t ev[kMaxPending][2];
  int kid[kMaxPending];
  int n_pending = 0;
  bool created = false;
  bool enabled = false;
  int64_t launches[16] = {0};
  double total_ms[16] = {0};
  int init() {
    if (created) return OSH_OK;
    for (int i = 0; i < kMaxPending; ++i) {
      OSH_HIP(hipEventCreate(&ev[i][0]));
      OSH_HIP(hipEventCreate(&ev[i][1]));
    }
    created = true;
    return OSH_OK;
  }
  void destroy() {
    if (!created) return;
    for (int i = 0; i < kMaxPending; ++i) { (void)hipEventDestroy(ev[i][0]); (void)hipEventDestroy(ev[i][1]); }
    created = false;
  }
  void reset() { for (int i = 0; i < 16; ++i) { launches[i] = 0; total_ms[i] = 0; } n_pending = 0; }
  inline bool begin(int k, hipStream_t s) {
    if (!enabled || n_pending >= kMaxPending) return false;
    kid[n_pending] = k;
    (void)hipEventRecord(ev[n_pending][0], s);
    return true;
  }
  inline void end(hipStream_t s) { (void)hipEventRecord(ev[n_pending][1], s); ++n_pending; }
  // call after the stream has been synchronised
  void collect() {
    for (int i = 0; i < n_pending; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[i][0], ev[i][1]) == hipSuccess) { launches[kid[i]]++; total_ms[kid[i]] += ms; }
    }
    n_pending = 0;
  }
};

}  // namespace osh
