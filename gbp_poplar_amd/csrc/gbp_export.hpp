// gbp_export.hpp — how a function of the C-ABI (include/gbp_mi355x*.h) is defined.  No HIP dependency: the pure-host translation
// unit (gbp_host.cpp) uses it as well as the device-facing ones (gbp_api_*.cpp, through gbp_ctx.hpp).
#pragma once
#include "../../include/gbp_mi355x.h"

#include <exception>
#include <new>
#include <string>

namespace gbp {
namespace api {

// text into the ctx (gbp_last_error(ctx)) or, c == NULL, the calling thread's ctx-less error (gbp_last_error(NULL)); returns `code`
int fail(gbp_ctx* c, int code, const std::string& msg);

// No C++ exception crosses the C-ABI (include/gbp_mi355x.h): every exported function runs its body inside this guard.
template <class F> int guarded(gbp_ctx* c, const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(c, GBP_ERR_NOMEM, std::string(what) + ": out of host memory");
  } catch (const std::exception& e) {
    return fail(c, GBP_ERR_INVALID, std::string(what) + ": " + e.what());
  } catch (...) {
    return fail(c, GBP_ERR_INVALID, std::string(what) + ": unknown exception");
  }
}

// EVERY function the headers under include/ declare is DEFINED through one of these three macros — the exported symbol is a shell
// around `<name>_body`, which the macro leaves open for the function's body — so that a new entry point cannot forget the guard
// (tests/test_cabi_symbols.py greps the translation units for a gbp_* export defined any other way).
//   GBP_EXPORT(gbp_sync, c, (gbp_ctx* c), (c)) { ...body returning a gbp_status... }
//   ctx: where the text of a caught exception goes (a gbp_ctx* parameter, or nullptr: the thread's create error)
#define GBP_EXPORT(name, ctx, sig, call)                                                                               \
  static int name##_body sig;                                                                                          \
  extern "C" GBP_API int name sig { return ::gbp::api::guarded(ctx, #name, [&]() -> int { return name##_body call; }); } \
  static int name##_body sig
// the few exports that return something else than a status: `fallback` is what a caught exception returns
#define GBP_EXPORT_T(ret, fallback, name, sig, call)                                                      \
  static ret name##_body sig;                                                                             \
  extern "C" GBP_API ret name sig { try { return name##_body call; } catch (...) { return fallback; } }   \
  static ret name##_body sig
#define GBP_EXPORT_VOID(name, sig, call)                                                 \
  static void name##_body sig;                                                           \
  extern "C" GBP_API void name sig { try { name##_body call; } catch (...) {} }          \
  static void name##_body sig

}  // namespace api
}  // namespace gbp
