// runcuda.cpp -- `int runcuda(GlobalState&)` with the reference's exact signature (gipuma.h:2),
// compiled against the reference's OWN headers (gipuma.h -> globalstate.h -> camera.h ...) through
// cuda_compat/.  Linking this object (plus libgipuma_hip.so) in place of gipuma.cu lets the
// reference's main.cpp call the MI355X path unchanged.  See INTEGRATION.md.
#include "gipuma.h"  // the reference's header, found via -I<reference dir>

#include "runcuda_adapter.h"

int runcuda(GlobalState &gs)
{
    unsigned seed = 1;  // extension: the reference seeds from clock64() (gipuma.cu:1019)
    if (const char *s = getenv("GIPUMA_SEED")) seed = (unsigned)strtoul(s, nullptr, 0);
    return gipuma_amd::runcuda_impl(gs, seed);
}
