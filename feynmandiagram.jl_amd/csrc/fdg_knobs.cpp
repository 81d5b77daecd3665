// see fdg_knobs.h
#include "fdg_knobs.h"

#include <cstring>
#include <mutex>

extern char **environ;

namespace fdg {
namespace {
thread_local const KnobMap *t_active = nullptr;
// what an installation may set through the environment; everything else is an experiment switch
const char *const kSupported[] = {"FDG_CACHE_DIR", "FDG_CACHE_RO_DIR", "FDG_CACHE_TRUST", "FDG_LLVM_BIN", "FDG_HIPCC", "FDG_JIT", "FDG_MC_ROUTE",
                                  "FDG_EVAL_CHUNK", "FDG_MC_CHUNK", "FDG_SM_CHUNK_MB", "FDG_IGNORE_TUNED", "FDG_LEAF_GENERIC", "FDG_TUNE_VERBOSE",
                                  "FDG_ISA_NO_POOL", "FDG_ISA_POOL", "FDG_ISA_NO_RL", "FDG_ISA_RL", "FDG_ISA_ALIGN"};
}  // namespace

bool knob_supported_from_env(const std::string &name) {
#ifdef FDG_DEV_SWITCHES
  return name.compare(0, 4, "FDG_") == 0;
#else
  for (const char *s : kSupported) if (name == s) return true;
  return false;
#endif
}

namespace {
std::mutex g_defaults_mu;
KnobMap &defaults_locked() {          // caller holds g_defaults_mu
  static KnobMap m = [] {
    KnobMap d;
    for (char **e = environ; e && *e; ++e) {
      if (std::strncmp(*e, "FDG_", 4) != 0) continue;
      const char *eq = std::strchr(*e, '=');
      if (!eq) continue;
      std::string name(*e, (size_t)(eq - *e));
      if (knob_supported_from_env(name)) d[name] = eq + 1;
    }
    return d;
  }();
  return m;
}
}  // namespace

KnobMap env_snapshot() {
  std::lock_guard<std::mutex> lk(g_defaults_mu);
  return defaults_locked();
}

void set_default_knob(const char *name, const char *value) {
  std::lock_guard<std::mutex> lk(g_defaults_mu);
  if (value) defaults_locked()[name] = value; else defaults_locked().erase(name);
}

const char *knob(const char *name) {
  if (t_active) {
    auto it = t_active->find(name);
    return it == t_active->end() ? nullptr : it->second.c_str();
  }
  // no handle on this thread (entry points without one: fdg_leaf_eval_device, fdg_copy_device): the process defaults, copied out
  thread_local std::string hold[4];
  thread_local unsigned slot = 0;
  std::lock_guard<std::mutex> lk(g_defaults_mu);
  const KnobMap &m = defaults_locked();
  auto it = m.find(name);
  if (it == m.end()) return nullptr;
  std::string &h = hold[slot++ & 3u];
  h = it->second;
  return h.c_str();
}

KnobScope::KnobScope(const KnobMap *m) : prev_(t_active) { t_active = m; }
KnobScope::~KnobScope() { t_active = prev_; }
}  // namespace fdg
