// prof.h -- optional per-launch timing with HIP events recorded on the launch stream.  Off by default
// (zero cost: one branch per launch); bench.py switches it on for an instrumented pass to obtain each
// kernel's average duration and algorithmic bytes/flops for the roofline line.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace mvae {

struct ProfRec { int tag; hipEvent_t a, b; double bytes, flops; };

struct Profiler {
  bool on = false;
  bool by_size = getenv("MVAE_PROF_SIZES") != nullptr;   // diagnostics: split every tag by log2(bytes)
  std::vector<std::string> tags;
  std::vector<ProfRec> recs;
  int cur_scale = -1;                         // set by the runtime around each pyramid scale's launches (diagnostics)
  int tag_id(const char* name) {
    for (size_t i = 0; i < tags.size(); ++i)
      if (tags[i] == name) return (int)i;
    tags.emplace_back(name);
    return (int)tags.size() - 1;
  }
};
Profiler& profiler();

struct ProfScope {
  int idx = -1;
  hipStream_t s;
  ProfScope(const char* tag, double bytes, double flops, hipStream_t stream) : s(stream) {
    Profiler& p = profiler();
    if (!p.on) return;
    ProfRec r;
    if (p.by_size) {
      char buf[96];
      snprintf(buf, sizeof(buf), "%s#%.0fMB/q%d", tag, bytes / 1e6, p.cur_scale);
      r.tag = p.tag_id(buf);
    } else {
      r.tag = p.tag_id(tag);
    }
    r.bytes = bytes; r.flops = flops;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    p.recs.push_back(r);
    idx = (int)p.recs.size() - 1;
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(profiler().recs[idx].b, s);
  }
};

}  // namespace mvae
