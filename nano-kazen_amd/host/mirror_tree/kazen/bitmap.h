// Header name of the reference (include/kazen/bitmap.h) forwarded to the host mirror, so that adapter/renderer_mi355x.cpp and
// adapter/kazen/mi355x.h - the files INTEGRATION.md adds to a kazen tree - compile UNCHANGED against it.
#include "../../kazen_host.hpp"
