// ptmi_build_id.cpp -- what this library was built from (include/ptmi.h: ptmi_build_id).
// Compiled at every link by _build.py with -DPTMI_BUILD_ID="<source hash>[+<extra flags>]"; the string stands behind a marker so
// that the FILE can be asked without loading it (_build.read_build_id), and the loaded library through the entry point.
#include "../../include/ptmi.h"

#ifndef PTMI_BUILD_ID
#error "ptmi_build_id.cpp is compiled by _build.py, which defines PTMI_BUILD_ID"
#endif

namespace {
constexpr char kMarkedId[] = "PTMI_BUILD_ID=" PTMI_BUILD_ID;
}

extern "C" const char *ptmi_build_id(void) { return kMarkedId + (sizeof("PTMI_BUILD_ID=") - 1); }
