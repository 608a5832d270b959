// lba_pack_device.h -- the batch packer of osh_lba_upload on the device (lba_pack_device.hip).
//
// Same output as lba_pack.h (the role of SparseOptimizer::initializeOptimization + BlockSolver::buildStructure,
// Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:199-267, block_solver.hpp:143-295), byte for byte, but produced by HIP kernels
// from the caller's edges in the caller's order: the host only narrows the observation records to float32 on their way into
// pinned staging.  lba_pack.h stays as the checker (tests compare the two section by section) and as the packer of
// a handful of windows (one host thread packs a window faster than three one-block kernels and two round trips).
#pragma once
#include "common.h"
#include "lba_pack.h"

namespace osh {

struct DevPackState {
  PinBuf h_raw, h_ctl;
  DevBuf d_raw, d_ctl, d_s1, d_s2, d_s3;
  // milliseconds of the last device_pack_batch: host staging pass, H2D + the three kernels (wall clock incl. the two round trips)
  double host_ms = 0, device_ms = 0;
  // HIP-event times of the last call: H2D of the staging, k_pack_pre1, k_pack_pre2, k_pack_post
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool timing = false;
  double ev_ms[4] = {0, 0, 0, 0};
  double cyc_mean[24] = {0};   // shader-clock cycles per phase of the three kernels, mean over the windows
  size_t raw_bytes = 0;
  void release_events() { for (hipEvent_t& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } }
};

// true when the batch can be packed on the device (always, since fisheye-rig pairs are merged there too)
bool device_pack_supported(int nw, const osh_lba_problem* pr);

// Packs `nw` problems into d_arena[0..1] (layout of PackedBatch, host pointers pb.arena[] stay null) and d_ptwin (window of every
// landmark).  Returns pb.err; message in pb.msg.  Synchronises `s` (the staging may be reused on return).
int device_pack_batch(DevPackState& st, hipStream_t s, int nw, const osh_lba_problem* pr, int n_threads, PackedBatch& pb, DevBuf* d_arena, DevBuf& d_ptwin);

// Test hook: packs with both packers and compares every section; stats[0..3] = bytes compared, sections compared, items, records.
int device_pack_compare(DevPackState& st, hipStream_t s, int nw, const osh_lba_problem* pr, int64_t stats[4]);

}  // namespace osh
