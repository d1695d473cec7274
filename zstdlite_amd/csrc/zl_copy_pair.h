// zl_copy_pair.h -- the two streams that carry the host entries' uploads and downloads, one pair per device and PROCESS.
//
// A host entry that moves a large buffer runs its upload beside its download (zl_compress2's pipeline; zl_decompressDCtx's
// slabs).  Whether two copies on two HIP streams run at the same time depends on which copy engine each stream got: on this
// runtime (the one PyTorch-ROCm brings) a stream keeps the engine its FIRST copy was given, that copy is given an engine that is
// idle at that moment, the engines are not equally fast (a download at 27 GB/s instead of 56 on some), and an engine a
// destroyed stream held does not come back into the choice (scripts/dev/copy_pair_state.py, scripts/dev/src/copy_engines.cpp:
// a bare process on the system's newer runtime gives every fresh pair two fast engines).  Until round 4 every context bound a
// pair of its own and tried fresh streams until the pair overlapped -- which worked for the first context of a process and less
// and less for later ones.  Now the pair is made once per device, with first copies that overlap in time, checked for overlap
// AND for the rate of either direction, and never destroyed; contexts share it (copies of two contexts that run at the same
// time take turns on it).
#pragma once
#include <hip/hip_runtime.h>

struct ZlCopyPair {
    hipStream_t up, down;
    int duplex;  // 1: the two directions overlap at full rate; 0: they were not seen to
};
// the current device's pair; nullptr when the streams cannot be had
__attribute__((visibility("hidden"))) const ZlCopyPair* zl_copy_pair(void);
