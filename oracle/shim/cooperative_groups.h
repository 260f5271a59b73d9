/* oracle shim: see simlod_host_shim.h (test infrastructure only) */
#pragma once
#include "simlod_host_shim.h"
