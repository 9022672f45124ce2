// include-name shim: the reference header hemoCellField.h; everything lives in hemocell.h
#pragma once
#include "hemocell.h"
