// include-name shim: the reference header cellMechanics.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
