// Error reporting and library-level entry points of libapr_hip.so.
#include <stdarg.h>
#include <sys/prctl.h>
#include <time.h>

#include "common.h"

static thread_local char g_err[512] = "";

void apr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

APR_API const char* apr_last_error(void) { return g_err; }

APR_API int apr_version(void) { return 100; }

// sizeof of every struct that crosses the boundary, in declaration order: a binding checks its own layout against these
// (tests/test_library_cpu.py does for apr_amd/_lib.py): a field added on one side only would otherwise corrupt silently
APR_API int32_t apr_struct_sizes(int32_t* out, int32_t n) {
  const int32_t sizes[] = {(int32_t)sizeof(apr_pair_desc),     (int32_t)sizeof(apr_spconv_desc),  (int32_t)sizeof(apr_resunet_layer),
                           (int32_t)sizeof(apr_resunet_plan),  (int32_t)sizeof(apr_level_map),    (int32_t)sizeof(apr_pyramid),
                           (int32_t)sizeof(apr_kp_resnet_desc), (int32_t)sizeof(apr_gcn_layer),   (int32_t)sizeof(apr_gcn_desc)};
  const int32_t have = (int32_t)(sizeof(sizes) / sizeof(sizes[0]));
  for (int32_t i = 0; i < have && i < n && out; ++i) out[i] = sizes[i];
  return have;
}

APR_API int apr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// Wait for a HIP event WITHOUT a spinning CPU and WITHOUT the caller's interpreter lock (ctypes releases the GIL around the
// call): hipEventQuery + nanosleep(poll_us).  hipEventSynchronize burns a CPU for the whole wait on this stack whether or
// not the event was created with hipEventBlockingSync (measured per thread: scripts/host_cpu_split.py), and polling the
// event from Python instead made every waiting thread take the interpreter lock 25 000 times a second -- on a box with a
// slow host that cost the enqueueing threads 14 % of the throughput.  The thread's timer slack is set to 1 us (the
// default 50 us would turn a 25 us sleep into a 75 us one) FOR THE DURATION OF THE WAIT and put back before returning: the
// caller's thread keeps its own setting (round-4 advice).  apr_event_wait_timeout adds a deadline: APR_ETIMEOUT instead of
// waiting for ever on a wedged queue.
static int event_wait_impl(void* event, int32_t poll_us, int64_t timeout_us) {
  APR_CHECK_ARG(event != nullptr && poll_us >= 0, "apr_event_wait: bad arguments");
  const int old_slack = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
  const bool changed = poll_us > 0 && old_slack > 1000 && prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL) == 0;
  struct timespec t_start;
  clock_gettime(CLOCK_MONOTONIC, &t_start);
  int rc = APR_OK;
  struct timespec ts;
  ts.tv_sec = poll_us / 1000000;
  ts.tv_nsec = (long)(poll_us % 1000000) * 1000L;
  for (;;) {
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) {
      apr_set_error("apr_event_wait: hipEventQuery -> %s", hipGetErrorString(e));
      rc = APR_EHIP;
      break;
    }
    (void)hipGetLastError();      // hipErrorNotReady is not an error
    if (timeout_us >= 0) {
      struct timespec now;
      clock_gettime(CLOCK_MONOTONIC, &now);
      const int64_t us = (int64_t)(now.tv_sec - t_start.tv_sec) * 1000000 + (now.tv_nsec - t_start.tv_nsec) / 1000;
      if (us > timeout_us) {
        apr_set_error("apr_event_wait: event not reached after %lld us", (long long)timeout_us);
        rc = APR_ETIMEOUT;
        break;
      }
    }
    if (poll_us > 0) nanosleep(&ts, nullptr);
  }
  if (changed) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack, 0UL, 0UL, 0UL);
  return rc;
}

APR_API int apr_event_wait(void* event, int32_t poll_us) { return event_wait_impl(event, poll_us, -1); }

APR_API int apr_event_wait_timeout(void* event, int32_t poll_us, int64_t timeout_us) {
  APR_CHECK_ARG(timeout_us >= 0, "apr_event_wait_timeout: negative timeout");
  return event_wait_impl(event, poll_us, timeout_us);
}
