// Error plumbing of the C ABI (include/grl_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/grl_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

int grl_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int grl_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return grl_fail(GRL_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GRL_OK;
}

extern "C" const char* grl_last_error(void) { return g_err; }
extern "C" int grl_abi_version(void) { return GRL_ABI_VERSION; }

// Stream ordering without a round trip through the host framework (round 6: a training step issues ~1500 launches and,
// in bf16 storage, takes as long as the HOST needs to issue them; `with torch.cuda.stream(..)` + torch.cuda.Event cost
// 15-30 us per hand-off in Python, this call ~1 us): `waiter` waits for everything enqueued on `signaler` so far.
// Events come from a per-thread ring: an event may be re-recorded while an earlier wait on it is pending (a wait captures
// the record that precedes it).  This is plain HIP plumbing, it replaces nothing in the reference (its DataParallel
// streams live inside torch).
extern "C" int grl_stream_wait_stream(void* waiter, void* signaler) {
    constexpr int RING = 64;
    static thread_local hipEvent_t ring[RING];
    static thread_local int used = 0, next = 0;
    if (waiter == signaler) return GRL_OK;
    if (used < RING) {
        if (hipEventCreateWithFlags(&ring[used], hipEventDisableTiming) != hipSuccess) return grl_check_launch("grl_stream_wait_stream (event)");
        ++used;
    }
    hipEvent_t ev = ring[next % used];
    next = (next + 1) % RING;
    if (hipEventRecord(ev, (hipStream_t)signaler) != hipSuccess) return grl_check_launch("grl_stream_wait_stream (record)");
    if (hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) return grl_check_launch("grl_stream_wait_stream (wait)");
    return GRL_OK;
}
