// tape.hip -- C-side step executor (include/bdm_hip.h section 5): a recorded reverse step -- the ~230 C-ABI calls of
// conditioning + denoiser forward, the memsets / device copies and the stream / event edges between them -- replayed by ONE call
// from the host framework.  Replaces the per-step Python of the reference's reverse loop (experiments/model/model.py:275-287).
// Host code only; calls are dispatched through thunks generated from the header (tools/gen_tape_thunks.py), so every function is
// called with its declared prototype (no libffi, no ABI tricks).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bdm_hip.h"
#include "common.h"

namespace {

inline double slot_double(uint64_t bits) {
  double d;
  memcpy(&d, &bits, sizeof(d));
  return d;
}

struct TapeThunk {
  const char *name;
  int (*fn)(const uint64_t *);
  int n_args;
};

#include "tape_thunks.inc"

enum Kind : int { CALL = 0, MEMSET, MEMCPY, WAIT_STREAM, EVENT_RECORD, EVENT_WAIT };

struct Entry {
  Kind kind;
  int (*fn)(const uint64_t *);
  uint32_t arg0;      // CALL: offset of the first argument slot in Tape::slots
  const char *name;   // CALL: function name (static storage of the thunk table)
  void *p0, *p1;      // MEMSET: dst | MEMCPY: dst, src | WAIT_STREAM: waiter, other | EVENT_*: event / stream
  size_t bytes;
  int value;
  hipEvent_t edge;    // WAIT_STREAM: the tape's own event
};

struct Tape {
  std::vector<Entry> entries;
  std::vector<uint64_t> slots;
  int failed = -1;
};

const std::unordered_map<std::string, const TapeThunk *> &thunk_index() {
  static const std::unordered_map<std::string, const TapeThunk *> index = [] {
    std::unordered_map<std::string, const TapeThunk *> m;
    for (const TapeThunk &t : kThunks) m.emplace(t.name, &t);
    return m;
  }();
  return index;
}

}  // namespace

extern "C" void *bdm_tape_create(void) { return new Tape(); }

extern "C" void bdm_tape_destroy(void *tape) {
  Tape *t = static_cast<Tape *>(tape);
  if (t == nullptr) return;
  for (Entry &e : t->entries)
    if (e.kind == WAIT_STREAM && e.edge != nullptr) (void)hipEventDestroy(e.edge);
  delete t;
}

extern "C" int bdm_tape_length(const void *tape) { return tape ? (int)static_cast<const Tape *>(tape)->entries.size() : 0; }

extern "C" int bdm_tape_failed_entry(const void *tape) { return tape ? static_cast<const Tape *>(tape)->failed : -1; }

extern "C" int bdm_tape_append_call(void *tape, const char *function, const unsigned long long *args, int n_args) {
  BDM_REQUIRE(tape != nullptr && function != nullptr && n_args >= 0 && (args != nullptr || n_args == 0), "tape_append_call: bad arguments");
  auto it = thunk_index().find(function);
  BDM_REQUIRE(it != thunk_index().end(), "tape_append_call: %s is not an `int bdm_*(...)` entry point of bdm_hip.h", function);
  BDM_REQUIRE(it->second->n_args == n_args, "tape_append_call: %s takes %d arguments, got %d", function, it->second->n_args, n_args);
  Tape *t = static_cast<Tape *>(tape);
  Entry e{};
  e.kind = CALL;
  e.fn = it->second->fn;
  e.name = it->second->name;
  e.arg0 = (uint32_t)t->slots.size();
  for (int i = 0; i < n_args; ++i) t->slots.push_back((uint64_t)args[i]);
  t->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_append_memset(void *tape, void *dst, int byte_value, size_t bytes, void *stream) {
  BDM_REQUIRE(tape != nullptr && (dst != nullptr || bytes == 0), "tape_append_memset: bad arguments");
  Entry e{};
  e.kind = MEMSET; e.p0 = dst; e.p1 = stream; e.bytes = bytes; e.value = byte_value;
  static_cast<Tape *>(tape)->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_append_memcpy(void *tape, void *dst, const void *src, size_t bytes, void *stream) {
  BDM_REQUIRE(tape != nullptr && ((dst != nullptr && src != nullptr) || bytes == 0), "tape_append_memcpy: bad arguments");
  Entry e{};
  e.kind = MEMCPY; e.p0 = dst; e.p1 = const_cast<void *>(src); e.bytes = bytes;
  e.edge = reinterpret_cast<hipEvent_t>(stream);  // (the stream rides in the spare pointer)
  static_cast<Tape *>(tape)->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_append_wait_stream(void *tape, void *waiter, void *other) {
  BDM_REQUIRE(tape != nullptr, "tape_append_wait_stream: bad arguments");
  Entry e{};
  e.kind = WAIT_STREAM; e.p0 = waiter; e.p1 = other;
  if (hipEventCreateWithFlags(&e.edge, hipEventDisableTiming) != hipSuccess) {
    bdm::set_error("tape_append_wait_stream: hipEventCreateWithFlags failed");
    return BDM_ERR_LAUNCH;
  }
  static_cast<Tape *>(tape)->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_append_event_record(void *tape, void *event, void *stream) {
  BDM_REQUIRE(tape != nullptr && event != nullptr, "tape_append_event_record: bad arguments");
  Entry e{};
  e.kind = EVENT_RECORD; e.p0 = event; e.p1 = stream;
  static_cast<Tape *>(tape)->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_append_event_wait(void *tape, void *stream, void *event) {
  BDM_REQUIRE(tape != nullptr && event != nullptr, "tape_append_event_wait: bad arguments");
  Entry e{};
  e.kind = EVENT_WAIT; e.p0 = event; e.p1 = stream;
  static_cast<Tape *>(tape)->entries.push_back(e);
  return BDM_OK;
}

extern "C" int bdm_tape_replay(void *tape, int first, int count) {
  BDM_REQUIRE(tape != nullptr && first >= 0, "tape_replay: bad arguments");
  Tape *t = static_cast<Tape *>(tape);
  const int n = (int)t->entries.size();
  const int last = count < 0 ? n : (first + count < n ? first + count : n);
  t->failed = -1;
  const uint64_t *slots = t->slots.data();
  for (int i = first; i < last; ++i) {
    const Entry &e = t->entries[i];
    int rc = BDM_OK;
    hipError_t he = hipSuccess;
    switch (e.kind) {
      case CALL: rc = e.fn(slots + e.arg0); break;
      case MEMSET: if (e.bytes) he = hipMemsetAsync(e.p0, e.value, e.bytes, (hipStream_t)e.p1); break;
      case MEMCPY: if (e.bytes) he = hipMemcpyAsync(e.p0, e.p1, e.bytes, hipMemcpyDeviceToDevice, (hipStream_t)e.edge); break;
      case WAIT_STREAM:
        he = hipEventRecord(e.edge, (hipStream_t)e.p1);
        if (he == hipSuccess) he = hipStreamWaitEvent((hipStream_t)e.p0, e.edge, 0);
        break;
      case EVENT_RECORD: he = hipEventRecord((hipEvent_t)e.p0, (hipStream_t)e.p1); break;
      case EVENT_WAIT: he = hipStreamWaitEvent((hipStream_t)e.p1, (hipEvent_t)e.p0, 0); break;
    }
    if (he != hipSuccess) {
      bdm::set_error("tape_replay: entry %d (kind %d): %s", i, (int)e.kind, hipGetErrorString(he));
      rc = BDM_ERR_LAUNCH;
    }
    if (rc != BDM_OK) {  // a failing call has left its own message (bdm_last_error)
      t->failed = i;
      return rc;
    }
  }
  return BDM_OK;
}

extern "C" int bdm_tape_echo(int a, long long b, float c, const void *d, unsigned int e, float f, int g, void *out16) {
  BDM_REQUIRE(out16 != nullptr, "tape_echo: out16 is NULL");
  double *o = static_cast<double *>(out16);
  o[0] = a; o[1] = (double)b; o[2] = c; o[3] = (double)(uintptr_t)d; o[4] = e; o[5] = f; o[6] = g;
  return g < 0 ? BDM_ERR_ARG : BDM_OK;
}
