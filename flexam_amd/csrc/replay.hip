// flexam_amd/csrc/replay.hip -- command lists: a recorded sequence of this library's own stream-ordered calls re-issued from ONE C call.
//
// Why: a denoise step is ~420 kernel launches (30 blocks x 13-16 calls), each reached through a Python wrapper and ctypes (20-30 us of
// host time per call).  On one GPU the host stays ahead of a 250 ms step, but one rank of eight runs the same launch sequence in ~40 ms:
// the host's enqueue time (11-21 ms per step measured in r5) becomes 0.25-0.5 of the rank's step and leaves no room for RCCL's own
// host cost.  The launch sequence of a block does not change from step to step -- same kernels, same buffers, same scalars; only the
// CONTENTS of the buffers change -- so flexam_amd/hip.py records it once (function id + argument words per call) and every later step
// hands the list back: flexam_replay walks it and calls the same extern "C" entry points, natively, in ~1 us per launch.
// This is a user-space command buffer, not a hipGraph: the launches stay ordinary launches on the caller's stream (collectives, events
// and other streams interleave between lists exactly as before), every call runs its own argument checks again, and nothing is
// retained here -- the list is the caller's memory (include/flexam_hip.h: no allocation, no retained pointers).
#include <string.h>

#include "common.h"
#include "flexam_hip.h"

#include "replay_table.inc"

extern "C" int flexam_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < FLEXAM_REPLAY_N; ++i)
    if (strcmp(name, FLEXAM_REPLAY_NAMES[i]) == 0) return i;
  return -1;
}

extern "C" int flexam_fn_count(void) { return FLEXAM_REPLAY_N; }

extern "C" const char* flexam_fn_name(int id) { return id >= 0 && id < FLEXAM_REPLAY_N ? FLEXAM_REPLAY_NAMES[id] : nullptr; }

extern "C" int flexam_replay(const flexam_cmd* cmds, int64_t n, int64_t* failed_at, void* stream) {
  FX_REQUIRE(cmds || n == 0, FLEXAM_E_ARG, "replay: null command list");
  FX_REQUIRE(n >= 0, FLEXAM_E_ARG, "replay: negative length");
  for (int64_t i = 0; i < n; ++i) {
    const flexam_cmd& c = cmds[i];
    if (c.fn < 0 || c.fn >= FLEXAM_REPLAY_N || c.nargs != FLEXAM_REPLAY_NARGS[c.fn] - 1) {
      if (failed_at) *failed_at = i;
      return flexam_fail(FLEXAM_E_ARG, "replay: command %ld has function id %d with %d argument words (the library's %s takes %d + the stream)", (long)i,
                         c.fn, c.nargs, c.fn >= 0 && c.fn < FLEXAM_REPLAY_N ? FLEXAM_REPLAY_NAMES[c.fn] : "?",
                         c.fn >= 0 && c.fn < FLEXAM_REPLAY_N ? FLEXAM_REPLAY_NARGS[c.fn] - 1 : -1);
    }
    const int rc = flexam_replay_dispatch(c.fn, c.a, stream);
    if (rc != FLEXAM_OK) {                      // the failing call left its own message in flexam_last_error()
      if (failed_at) *failed_at = i;
      return rc;
    }
  }
  if (failed_at) *failed_at = -1;
  return FLEXAM_OK;
}

// ---- flexam_delay_us: one wave that waits `us` microseconds of the constant 100 MHz counter (s_memrealtime: independent of the shader
// clock, which moves with the power cap) and does nothing else.  The emulation of a multi-GPU rank on one GPU (flexam_amd.dist.LoopbackGroup,
// bench.py --emulate-rank) puts it on the side stream where a collective's transfer time would sit: link time = bytes per link / an ASSUMED
// link rate.  Not used by the product path.
namespace {
__global__ __launch_bounds__(64) void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int flexam_delay_us(float us, void* stream) {
  FX_REQUIRE(us >= 0.f && us <= 1e6f, FLEXAM_E_ARG, "delay_us: %g us (0 .. 1e6)", (double)us);
  const unsigned long long ticks = (unsigned long long)((double)us * 100.0);
  if (ticks == 0) return FLEXAM_OK;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
  return flexam_check_launch("flexam_delay_us");
}
