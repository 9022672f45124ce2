// include-name shim: the reference header immersedBoundaryMethod.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
