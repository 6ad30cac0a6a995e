// capi.hip — version + error plumbing of the C ABI (include/sgf.h).
#include "common.h"

namespace sgf {
namespace {
// errno-style: the text belongs to the calling thread (the autograd engine calls the backward entry
// points from its own thread; a failure there must not overwrite or tear the main thread's message)
thread_local char g_err[1024] = "";
}  // namespace

int g_env_epoch = 0;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sgf

extern "C" int sgf_version(void) { return SGF_VERSION; }

extern "C" const char* sgf_last_error(void) { return sgf::g_err; }

extern "C" int sgf_reload_env(void) { return ++sgf::g_env_epoch; }
