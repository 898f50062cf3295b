// Build shim for OpenEXR's ImfThreading.h (reference use: pbrt.cpp:63).
#pragma once
namespace Imf { static inline void setGlobalThreadCount(int) {} }
