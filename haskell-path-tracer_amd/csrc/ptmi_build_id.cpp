// ptmi_build_id.cpp -- which code this library holds (include/ptmi.h: ptmi_build_id).
// Compiled at every link by _build.py with -DPTMI_BUILD_ID="<code id>[+<extra flags>]" -- the hash over the allocated sections of
// the objects being linked (host code and the gfx950 code objects; _build.code_id_of) -- and -DPTMI_SOURCE_HASH="<hash of the
// source text>", which only lets the build see without a compiler that nothing has been edited since the link.  Both stand
// behind markers so that the FILE can be asked without loading it (_build.read_build_id / read_source_hash).
#include "../../include/ptmi.h"

#if !defined(PTMI_BUILD_ID) || !defined(PTMI_SOURCE_HASH)
#error "ptmi_build_id.cpp is compiled by _build.py, which defines PTMI_BUILD_ID and PTMI_SOURCE_HASH"
#endif

namespace {
constexpr char kMarkedId[] = "PTMI_BUILD_ID=" PTMI_BUILD_ID;
// (volatile use below: the linker must not drop a string nobody reads at run time)
constexpr char kMarkedText[] = "PTMI_SOURCE_HASH=" PTMI_SOURCE_HASH;
}

extern "C" const char *ptmi_build_id(void)
{
    static const char *volatile keep = kMarkedText;
    (void)keep;
    return kMarkedId + (sizeof("PTMI_BUILD_ID=") - 1);
}
