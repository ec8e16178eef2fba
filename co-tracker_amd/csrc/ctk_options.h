// Process-wide back-end options (include/ctk.h: ctk_set_option / ctk_get_option).  Every choice a host may legitimately make
// between two kernels of the library lives in ONE table of relaxed atomics: read on the enqueue path with ctk_opt(), written
// only by ctk_set_option() (validated) and, once, by the table's static initialiser from the environment variables listed in
// include/ctk.h.  Nothing under csrc/ calls getenv() on an enqueue path (a getenv beside a setenv in another host thread is a
// data race), and no option of the release build can change a RESULT beyond the documented last-bit differences between two
// back ends of the same operator.
#pragma once
#include "../../include/ctk.h"

int ctk_opt(int key);  // relaxed load; key must be a valid CTK_OPT_* (not checked)

// Experiment knobs of DEV builds (make dev -> libctk_hip_dev.so, used by tools/): an environment variable read once.  In the
// release library the macro is the compile-time default -- the variable's name does not even appear in the binary.
#ifdef CTK_DEV
#include <cstdlib>
#define CTK_DEV_KNOB(name, def) ([]() -> long { static const long v_ = [] { const char* e_ = getenv(name); return e_ ? atol(e_) : (long)(def); }(); return v_; }())
#else
#define CTK_DEV_KNOB(name, def) ((long)(def))
#endif
