// include-name shim: the reference header hemoCellParticleField.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
