// oracle/wf_cpu/wf_kat.cpp — TEST INFRASTRUCTURE ONLY.  Runs the known-answer probe of pbrt-v4_amd/csrc/common/wf_kat.h on the host over
// the records of <golden_dir>/kat_in.bin (written by oracle/ref_build/ref_kat.cpp with the reference's own routines) and writes the
// restated routines' answers in the same layout:   wf_kat <in.bin> <out.bin>
#include "../../pbrt-v4_amd/csrc/common/wf_kat.h"

#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: wf_kat <in.bin> <out.bin>\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> in(sz / 8);
    if (fread(in.data(), 8, in.size(), f) != in.size()) return 1;
    fclose(f);
    const size_t n = in.size() / 16;
    std::vector<uint64_t> out(n * 8);
    for (size_t i = 0; i < n; ++i) wf::KatRun(&in[16 * i], &out[8 * i]);
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(out.data(), 8, out.size(), f);
    fclose(f);
    return 0;
}
