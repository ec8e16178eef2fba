// The option table behind ctk_set_option / ctk_get_option (ctk_options.h).
#include "ctk_options.h"
#include <atomic>
#include <cstdlib>

namespace {

struct OptSpec {
  const char* env;  // environment variable read ONCE, when the library is loaded (initial value)
  int def, lo, hi;  // default and valid range
  int mask;         // for bit-set options: the bits a release build accepts (0 = plain range check)
};

#ifdef CTK_DEV
constexpr int PP_MASK = 0x7fffffff;  // dev build: experiment bits of the persistent GEMMs (gemm_pp.hip)
#else
constexpr int PP_MASK = 1 | 32;
#endif

constexpr OptSpec kSpec[CTK_OPT_COUNT] = {
    /* CTK_OPT_GEMM_PP                   */ {"CTK_GEMM_PP", 33, 0, 0x7fffffff, PP_MASK},
    /* CTK_OPT_GEMM_TAIL_PCT             */ {"CTK_GEMM_TAIL_PCT", 25, 0, 100, 0},
    /* CTK_OPT_CORR_VERSION              */ {"CTK_CORR", 3, 1, 3, 0},
    /* CTK_OPT_CORR_MAP                  */ {"CTK_CORR_MAP", 3, 0, 4, 0},
    /* CTK_OPT_ATTENTION_VALU            */ {"CTK_ATTN", 0, 0, 1, 0},
    /* CTK_OPT_ATTENTION_TIME_PERSISTENT */ {"CTK_ATTN_TIME", 1, 0, 1, 0},
    /* CTK_OPT_OVERLAP                   */ {"CTK_OVERLAP", 0, 0, 3, 3},
};

bool valid(int key, int v) {
  const OptSpec& s = kSpec[key];
  if (v < s.lo || v > s.hi) return false;
  if (s.mask && (v & ~s.mask) != 0) return false;
  if (key == CTK_OPT_CORR_VERSION && v == 2) return false;  // version 2 left the library in round 6
  return true;
}

struct Table {
  std::atomic<int> v[CTK_OPT_COUNT];
  Table() {
    for (int k = 0; k < CTK_OPT_COUNT; ++k) {
      int x = kSpec[k].def;
      if (const char* e = getenv(kSpec[k].env)) {
        const int y = atoi(e);
        if (valid(k, y)) x = y;  // an invalid value in the environment leaves the default in place
      }
      v[k].store(x, std::memory_order_relaxed);
    }
  }
};
Table g_table;  // constructed when the library is loaded

}  // namespace

int ctk_opt(int key) { return g_table.v[key].load(std::memory_order_relaxed); }

extern "C" int ctk_set_option(int key, int value) {
  if (key < 0 || key >= CTK_OPT_COUNT || !valid(key, value)) return CTK_E_SHAPE;
  g_table.v[key].store(value, std::memory_order_relaxed);
  return CTK_OK;
}

extern "C" int ctk_get_option(int key, int* value) {
  if (!value) return CTK_E_NULL;
  if (key < 0 || key >= CTK_OPT_COUNT) return CTK_E_SHAPE;
  *value = ctk_opt(key);
  return CTK_OK;
}
