// Build shim: see ImfShim.h
#pragma once
#include "ImfShim.h"
