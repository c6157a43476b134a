// Internal to libpfnl_hip: the error slot behind pfnl_last_error(), shared by the translation units of the C-ABI.
#pragma once
#include <string>

int pfnl_internal_fail(int code, const std::string& msg);   // records msg for pfnl_last_error(); returns code
