// runtime.hip -- error string, version, profiling slots.
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace gpz {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Ownership rule for HIP objects (streams, events): the library creates none implicitly and keeps none in storage
// with a destructor.  Streams always come from the caller.  The only HIP objects the library ever creates are the
// profiling events below, created between gpz_profile_enable(1) and gpz_profile_enable(0) and destroyed by the
// latter -- so a process that has switched profiling off holds nothing of ours when the HIP runtime and any
// profiler tool layered on it tear down at exit.  (Round 2's experimental look-ahead Cholesky kept two extra
// streams, one of them CU-masked, alive until process exit; under rocprofv3 that build died with SIGSEGV inside
// __cxa_finalize after the tool's own finalisation -- the runtime destroying streams through an interception layer
// that was already gone.  Any future multi-stream schedule gets an explicit create / destroy pair on a handle.)
//
// Profiling: per slot, a growing list of (start, stop) event pairs recorded on
// the caller's stream; read back (and reset) by gpz_profile_read.
struct ProfState {
  bool on = false;
  static constexpr int MAXEV = 4096;
  hipEvent_t ev[PROF_NSLOTS][MAXEV][2];
  int created[PROF_NSLOTS] = {0};
  int used[PROF_NSLOTS] = {0};
};
static ProfState g_prof;

bool prof_enabled() { return g_prof.on; }

void prof_begin(int slot, hipStream_t s) {
  if (!g_prof.on) return;
  int& u = g_prof.used[slot];
  if (u >= ProfState::MAXEV) return;
  if (u >= g_prof.created[slot]) {
    (void)hipEventCreate(&g_prof.ev[slot][u][0]);
    (void)hipEventCreate(&g_prof.ev[slot][u][1]);
    g_prof.created[slot] = u + 1;
  }
  (void)hipEventRecord(g_prof.ev[slot][u][0], s);
}

void prof_end(int slot, hipStream_t s) {
  if (!g_prof.on) return;
  int& u = g_prof.used[slot];
  if (u >= ProfState::MAXEV) return;
  (void)hipEventRecord(g_prof.ev[slot][u][1], s);
  ++u;
}

}  // namespace gpz

extern "C" int gpz_version(void) { return GPZ_VERSION; }

// sha256 of the sources this binary was built from (gpzoo_amd/build.py passes it in); the marker prefix lets the
// build script find the value in the file without loading it.
#ifndef GPZ_SOURCE_HASH
#define GPZ_SOURCE_HASH "unknown"
#endif
extern "C" const char* gpz_source_hash(void) {
  static const char marked[] = "GPZ_SRC_HASH:" GPZ_SOURCE_HASH;
  return marked + 13;
}

extern "C" const char* gpz_last_error(void) { return gpz::g_err; }

extern "C" int gpz_profile_enable(int32_t on) {
  using namespace gpz;
  g_prof.on = on != 0;
  for (int s = 0; s < PROF_NSLOTS; ++s) {
    g_prof.used[s] = 0;
    if (!on) {       // switching off releases every event: nothing of ours outlives the caller's use of the profiler
      for (int i = 0; i < g_prof.created[s]; ++i) {
        (void)hipEventDestroy(g_prof.ev[s][i][0]);
        (void)hipEventDestroy(g_prof.ev[s][i][1]);
      }
      g_prof.created[s] = 0;
    }
  }
  return 0;
}

// Sums the elapsed time of every recorded interval per slot (host sync on the
// events), then resets the slots.  ms_out / counts_out: host arrays of n_slots.
extern "C" int gpz_profile_read(double* ms_out, int32_t* counts_out, int32_t n_slots) {
  using namespace gpz;
  GPZ_REQUIRE(ms_out && counts_out && n_slots >= 1, "gpz_profile_read: bad arguments");
  for (int s = 0; s < n_slots; ++s) { ms_out[s] = 0.0; counts_out[s] = 0; }
  for (int s = 0; s < PROF_NSLOTS && s < n_slots; ++s) {
    for (int i = 0; i < g_prof.used[s]; ++i) {
      GPZ_HIP_OK(hipEventSynchronize(g_prof.ev[s][i][1]));
      float ms = 0.f;
      GPZ_HIP_OK(hipEventElapsedTime(&ms, g_prof.ev[s][i][0], g_prof.ev[s][i][1]));
      ms_out[s] += ms;
    }
    counts_out[s] = g_prof.used[s];
    g_prof.used[s] = 0;
  }
  return 0;
}
