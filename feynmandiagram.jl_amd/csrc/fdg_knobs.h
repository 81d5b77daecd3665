// Options of a handle ("knobs").  Nothing in the library reads the process environment on a launch or a specialisation path: the FDG_*
// variables are copied ONCE per process into a snapshot (env_snapshot), every handle starts with a copy of it, and from then on a handle's
// behaviour is a function of its own option map, changed only through fdg_graph_set_option (include/fdg.h).  Code that used to call
// getenv("FDG_X") calls knob("FDG_X"): the option of the handle whose entry point is running on this thread (KnobScope), else the snapshot.
//
// A product build copies only the SUPPORTED variables from the environment; the experiment switches of the dev tools under
// tools/ reach the library through the environment only in a build with -DFDG_DEV_SWITCHES (`make dev`), and through
// fdg_graph_set_option in any build (tests flip kernel variants that way).
#pragma once
#include <map>
#include <string>

namespace fdg {
using KnobMap = std::map<std::string, std::string>;
KnobMap env_snapshot();                                  // the process defaults: the supported FDG_* variables of the environment as they
                                                         // were at first use, plus what fdg_set_default_option changed since
void set_default_knob(const char *name, const char *value);
const char *knob(const char *name);                       // nullptr when unset
struct KnobScope {
  explicit KnobScope(const KnobMap *m);
  ~KnobScope();
  KnobScope(const KnobScope &) = delete;
  KnobScope &operator=(const KnobScope &) = delete;
 private:
  const KnobMap *prev_;
};
bool knob_supported_from_env(const std::string &name);
}  // namespace fdg
