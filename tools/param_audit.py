#!/usr/bin/env python3
"""tools/param_audit.py — every parameter name the reference's Create() functions look up on this path must be looked up by this build
too: ParameterDictionary::ReportUnused (paramdict.cpp:612-636) is an ErrorExit in both, so a parameter the reference reads and this
build does not would turn a valid scene into an "unused parameter" error ("faceIndices").

    python tools/param_audit.py [--write]      # --write: refresh tests/golden/reference_param_names.txt from /root/reference

Without /root/reference the committed list is used.  Prints the names that no Get*("name") / GetTexture / texture-lookup call of
pbrt-v4_amd/csrc/host/*.cpp mentions and that are not in EXEMPT (parameters of entities outside SURVEY 8: the other integrators).
Exit status 1 if any.  Test infrastructure (tests/test_host.py runs it)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/pbrt"
LIST = os.path.join(ROOT, "tests", "golden", "reference_param_names.txt")
FILES = ["shapes.cpp", "materials.cpp", "lights.cpp", "cameras.cpp", "film.cpp", "filters.cpp", "samplers.cpp", "media.cpp", "textures.cpp", "scene.cpp",
         "lightsamplers.cpp", "cpu/aggregates.cpp", "cpu/integrators.cpp", "wavefront/integrator.cpp", "util/loopsubdiv.cpp"]
# parameters of what SURVEY 8 leaves out: the CPU integrators other than the path this build replaces (cpu/integrators.cpp), read by
# Integrator::Create only — the wavefront integrator reads maxdepth / regularize / lightsampler (wavefront/integrator.cpp:123-180)
EXEMPT = {"bootstrapsamples": "MLT", "chains": "MLT", "largestepprobability": "MLT", "mutationsperpixel": "MLT", "cossample": "AO", "maxdistance": "AO",
          "function": "function integrator", "imagefilename": "function integrator", "skipbad": "function integrator", "photonsperiteration": "SPPM", "radius": None,
          "samplebsdf": "path / simplepath", "samplelights": "simplepath", "visualizestrategies": "BDPT", "visualizeweights": "BDPT"}
del EXEMPT["radius"]   # ("radius" is also a shape parameter: looked up)


def reference_names():
    names = {}
    for f in FILES:
        p = os.path.join(REF, f)
        if not os.path.exists(p):
            continue
        for m in re.finditer(r'(GetOne[A-Za-z0-9]*|Get[A-Za-z0-9]*Array|Get(?:Float|Spectrum)Texture(?:OrNull)?|GetTexture)\(\s*"([A-Za-z_0-9.]+)"', open(p).read()):
            names.setdefault(m.group(2), set()).add(f)
    return names


def main():
    if os.path.isdir(REF):
        names = reference_names()
        if "--write" in sys.argv:
            with open(LIST, "w") as f:
                f.write("# parameter names looked up by the reference's Create() functions (tools/param_audit.py --write): name<TAB>files\n")
                for n in sorted(names):
                    f.write("%s\t%s\n" % (n, ",".join(sorted(names[n]))))
    else:
        names = {}
        for l in open(LIST):
            if l.startswith("#") or not l.strip():
                continue
            n, fs = l.rstrip("\n").split("\t")
            names[n] = set(fs.split(","))
    src = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "pbrt-v4_amd", "csrc", "host", "*.cpp")) + glob.glob(os.path.join(ROOT, "pbrt-v4_amd", "csrc", "host", "*.h")))
    missing = [n for n in sorted(names) if n not in EXEMPT and not re.search(r'"%s"' % re.escape(n), src)]
    for n in missing:
        print("NOT LOOKED UP: %-28s (reference: %s)" % (n, ", ".join(sorted(names[n]))))
    print("%d reference parameter names, %d exempt (other integrators), %d not looked up" % (len(names), sum(1 for n in names if n in EXEMPT), len(missing)))
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
