// Build shim: opaque GLFW types so util/gui.h parses; the GUI is never instantiated.
#pragma once
struct GLFWwindow;
