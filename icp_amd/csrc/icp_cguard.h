// icp_cguard.h — the C boundary never lets a C++ exception through.
#pragma once
#include "../../include/icp_amd.h"
#include <new>

// SURVEY.md §8b "Errors" (what not to inherit: the reference's entry points throw, src/ICP/algorithms.cpp:4411-4427): every extern "C"
// function that returns a status is a function-try-block ending in ICP_CATCH_ALL — std::bad_alloc (a handle's containers, an error text)
// becomes ICP_ENOMEM, anything else ICP_EHIP; no allocation on this path (the handle's error text stays what it was).
namespace icp_host {
inline int on_exception () noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { return ICP_ENOMEM; }
    catch (...) { return ICP_EHIP; }
}
}  // namespace icp_host
#define ICP_CATCH_ALL catch (...) { return icp_host::on_exception (); }
