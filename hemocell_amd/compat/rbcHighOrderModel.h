// compat forwarding header: the reference spreads this interface over several headers; here everything
// lives in hemocell.h (host facade over the C ABI)
#pragma once
#include "hemocell.h"
