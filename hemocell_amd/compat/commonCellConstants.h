// include-name shim: the reference header commonCellConstants.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
