// zl_copy_pair.h -- the two streams that carry the host entries' uploads and downloads, one pair per device and PROCESS.
//
// A host entry that moves a large buffer runs its upload beside its download (zl_compress2's pipeline; zl_decompressDCtx's
// slabs).  Whether two copies on two HIP streams run at the same time depends on the copy engine each stream got.  Measured on
// the runtime a Python process gets (the one PyTorch-ROCm brings; scripts/dev/copy_engine_policy.py, copy_pair_state.py):
//   * a stream keeps the engine of its FIRST copy, and that copy is given an engine that is idle at that moment: two streams
//     whose first copies ran one after the other share an engine for good (upload and download take turns: 1.19 ms for
//     2 x 32 MiB instead of 0.70), two whose first copies overlapped in time do not;
//   * now and then a fresh stream's downloads run at half the rate even alone (27 GB/s instead of 56) -- seen on streams created
//     right after large device buffers had been freed (copy_thread_exit.py; after zl_freeCCtx in copy_pair_state.py), not
//     understood; pairs made a moment later were fine again.
// (A bare process on the system's newer runtime gives every fresh pair two engines whatever the order of its first copies:
// scripts/dev/src/copy_engines.cpp.)  Until round 4 every context bound a pair of its own and tried fresh streams until the pair
// overlapped; a context created after another one had been freed could end up with the slow kind for all eight attempts.
// Now the pair is made once per device and process -- first copies that overlap, checked for overlap AND for the rate of either
// direction --, never destroyed, and shared by the contexts (copies of two contexts that run at the same time take turns on it).
#pragma once
#include <hip/hip_runtime.h>

struct ZlCopyPair {
    hipStream_t up, down;
    int duplex;  // 1: the two directions overlap at full rate; 0: they were not seen to
};
// the current device's pair; nullptr when the streams cannot be had
__attribute__((visibility("hidden"))) const ZlCopyPair* zl_copy_pair(void);
