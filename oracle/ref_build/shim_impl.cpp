// oracle/ref_build/shim_impl.cpp — TEST INFRASTRUCTURE ONLY (part of the shimmed reference build).
// Supplies the symbols whose reference sources are absent or excluded:
//  * util/bluenoise.cpp and util/pmj02tables.cpp are listed in /root/reference/.MISSING_LARGE_BLOBS;
//    the tables are zero-filled here, so the "pmj02bn" sampler must not be used with this build.
//  * util/gui.cpp (GLFW/OpenGL window) is not compiled; the GUI members referenced by
//    wavefront/integrator.cpp:305-478 abort if ever reached (--interactive is never passed).
//  * util/stbimage.cpp (stb implementation TU) is replaced by the inline failing stubs in shims/stb.
#include <pbrt/util/bluenoise.h>
#include <pbrt/util/gui.h>
#include <pbrt/util/pmj02tables.h>

#include <cstdio>
#include <cstdlib>

namespace pbrt {

PBRT_CONST uint16_t
    BlueNoiseTextures[NumBlueNoiseTextures][BlueNoiseResolution][BlueNoiseResolution] = {};
PBRT_CONST uint32_t pmj02bnSamples[nPMJ02bnSets][nPMJ02bnSamples][2] = {};

static void noGUI() {
    fprintf(stderr, "GUI is not available in the oracle build\n");
    abort();
}
GUI::GUI(std::string, Vector2i, Bounds3f) { noGUI(); }
GUI::~GUI() {}
DisplayState GUI::RefreshDisplay() { noGUI(); return DisplayState::EXIT; }
void GUI::Initialize() { noGUI(); }
Point2i GUI::GetResolution() { noGUI(); return Point2i(0, 0); }

}  // namespace pbrt
