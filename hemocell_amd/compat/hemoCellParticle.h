// include-name shim: the reference header hemoCellParticle.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
