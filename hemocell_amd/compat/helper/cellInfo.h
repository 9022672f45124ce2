#pragma once
#include "../hemocell.h"
