// capi.hip — version + error plumbing of the C ABI (include/sgf.h).
#include "common.h"

#include <mutex>

namespace sgf {
namespace {
std::mutex g_err_mu;
char g_err[1024] = "";
}  // namespace

void set_error(const char* fmt, ...) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sgf

extern "C" int sgf_version(void) { return SGF_VERSION; }

extern "C" const char* sgf_last_error(void) { return sgf::g_err; }
