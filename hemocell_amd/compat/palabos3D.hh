#pragma once
#include "palabos3D.h"
