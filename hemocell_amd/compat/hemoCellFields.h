// include-name shim: the reference header hemoCellFields.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
