// nanovdb_io.cpp — an own reader for NanoVDB files (.nvdb), for `MakeNamedMedium "..." "string type" "nanovdb"`
// (NanoVDBMedium::Create -> readGrid, media.cpp:488-510,640-660: nanovdb::io::readGrid(filename, gridName)).
//
// PARITY UNPINNED.  NanoVDB is a third-party submodule that is absent from the reference checkout (src/ext/openvdb, pinned at
// 414bed84, .gitmodules) and stubbed in the oracle build (oracle/ref_build/shims/nanovdb): nothing here can be checked against the
// reference or against a file NanoVDB itself wrote.  The container and node layouts below restate NanoVDB.h / io/IO.h of the 32.x
// ABI from its published description (FileHeader / FileMetaData / GridData 672 B / TreeData 64 B / RootData + 32-byte tiles /
// InternalNode<5> / InternalNode<4> / LeafNode<float> 8^3); files of another major ABI version are refused with the version in the
// message.  tools/make_nanovdb.py writes files of exactly this layout (fixtures of the tests).
//
// What the renderer needs from a grid is a function index -> float (SampleFromVoxels<Tree, 1, false> = trilinear interpolation of
// ReadAccessor::getValue at the eight surrounding voxels, media.h:626-629): on a 288 GB part the tree is expanded into a DENSE block
// over the grid's index bounding box (the quarter-resolution Disney cloud: 0.4 GB; full resolution: 26 GB) — one coalescable
// gather per corner instead of a three-level pointer chase per corner.
#include "scene.h"

#include <cstring>
#include <zlib.h>

namespace wf {

namespace {
[[noreturn]] void Fail(const std::string &fn, const std::string &why) { throw SceneError("Error: nanovdb: " + fn + ": " + why); }

constexpr uint64_t kMagic = 0x304244566f6e614eull;   // "NanoVDB0"
#pragma pack(push, 1)
struct FileHeader { uint64_t magic; uint32_t version; uint16_t gridCount, codec; };
struct FileMetaData {
    uint64_t gridSize, fileSize, nameKey, voxelCount;
    uint32_t gridType, gridClass;
    double worldBBox[6];
    int32_t indexBBox[6];
    double voxelSize[3];
    uint32_t nameSize, nodeCount[4], tileCount[3];
    uint16_t codec, padding;
    uint32_t version;
};
struct MapData { float matF[9], invMatF[9], vecF[3], taperF; double matD[9], invMatD[9], vecD[3], taperD; };
struct GridData {
    uint64_t magic, checksum;
    uint32_t version, flags, gridIndex, gridCount;
    uint64_t gridSize;
    char gridName[256];
    MapData map;
    double worldBBox[6];
    double voxelSize[3];
    uint32_t gridClass, gridType;
    int64_t blindMetadataOffset;
    uint32_t blindMetadataCount, data0;
    uint64_t data1, data2;
};
struct TreeData { uint64_t nodeOffset[4]; uint32_t nodeCount[3], tileCount[3]; uint64_t voxelCount; };
struct RootData { int32_t bbox[6]; uint32_t tableSize; float background, minimum, maximum, average, stdDevi; uint8_t pad[16]; };
struct RootTile { uint64_t key; int64_t child; uint32_t state; float value; uint8_t pad[8]; };
#pragma pack(pop)
static_assert(sizeof(FileHeader) == 16 && sizeof(FileMetaData) == 176 && sizeof(MapData) == 264 && sizeof(GridData) == 672 && sizeof(TreeData) == 64 &&
                  sizeof(RootData) == 64 && sizeof(RootTile) == 32,
              "NanoVDB 32.x layout");
// InternalNode<LOG2DIM>: bbox 24, flags 8, value mask, child mask, min max avg dev 16, padded to 32, then (1 << 3 LOG2DIM) 8-byte entries
constexpr size_t InternalHeader(int log2dim) { return (24 + 8 + 2 * ((size_t)1 << (3 * log2dim)) / 8 + 16 + 31) / 32 * 32; }
constexpr size_t kLeafHeader = 96;   // bboxMin 12, bboxDif 3, flags 1, value mask 64, min max avg dev 16
constexpr uint32_t kGridTypeFloat = 1, kClassUnknown = 0, kClassFog = 2;

struct Blob {
    const uint8_t *p;
    size_t n;
    const std::string &fn;
    template <typename T>
    T at(size_t off) const {
        if (off > n || n - off < sizeof(T)) Fail(fn, "grid data is truncated or corrupt (offset outside the grid)");
        T v;
        memcpy(&v, p + off, sizeof(T));
        return v;
    }
    const uint8_t *span(size_t off, size_t bytes) const {
        if (off > n || n - off < bytes) Fail(fn, "grid data is truncated or corrupt (node outside the grid)");
        return p + off;
    }
};

// the tree expanded into g->values over g->min .. g->min + g->dim - 1 (everything ReadAccessor::getValue returns there: leaf voxels,
// tile values — active or not —, the background elsewhere)
void Densify(const Blob &b, VdbGrid *g) {
    const GridData gd = b.at<GridData>(0);
    if (gd.magic != kMagic) Fail(b.fn, "grid magic number mismatch");
    if (gd.gridType != kGridTypeFloat) Fail(b.fn, "only float grids are supported (grid type " + std::to_string(gd.gridType) + ")");
    const size_t treeOff = sizeof(GridData);
    const TreeData td = b.at<TreeData>(treeOff);
    const size_t rootOff = treeOff + td.nodeOffset[3];
    const RootData rd = b.at<RootData>(rootOff);
    g->background = rd.background;
    memcpy(g->invMat, gd.map.invMatF, sizeof(g->invMat));
    memcpy(g->vec, gd.map.vecF, sizeof(g->vec));
    for (int i = 0; i < 6; ++i) g->worldBBox[i] = gd.worldBBox[i];
    g->gridClass = (int)gd.gridClass;
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        g->min[a] = rd.bbox[a];
        g->dim[a] = rd.bbox[3 + a] >= rd.bbox[a] ? rd.bbox[3 + a] - rd.bbox[a] + 1 : 0;
        cells *= g->dim[a];
    }
    if (cells <= 0) { g->dim[0] = g->dim[1] = g->dim[2] = 0; g->values.clear(); return; }   // an empty grid: background everywhere
    // (the dense expansion is bounded by what the FILE can describe, not only by the address range: a few header bytes must not ask for
    //  gigabytes — every active leaf of 512 voxels takes at least 2 KB of the file)
    if (cells > 4096ll * (long long)b.n + (1ll << 24))
        Fail(b.fn, "the grid's index bounding box holds " + std::to_string(cells) + " voxels, implausibly many for a file of " + std::to_string(b.n) + " bytes");
    if (cells > (1ll << 31) - 1) Fail(b.fn, "the grid's index bounding box holds " + std::to_string(cells) + " voxels: more than this build's dense expansion addresses");
    g->values.assign((size_t)cells, rd.background);
    auto fill = [&](const int o[3], int size, float v) {   // a tile [o, o + size)^3 clipped to the block
        // (64-bit bounds: a root key near 2^31 of a malformed file would overflow o + size; such a tile lies outside the block and is dropped)
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            const long long l = std::max<long long>(o[a], g->min[a]), h = std::min<long long>((long long)o[a] + size, (long long)g->min[a] + g->dim[a]);
            if (l >= h) return;
            lo[a] = (int)l; hi[a] = (int)h;
        }
        for (int z = lo[2]; z < hi[2]; ++z)
            for (int y = lo[1]; y < hi[1]; ++y) {
                float *row = &g->values[((size_t)(z - g->min[2]) * g->dim[1] + (y - g->min[1])) * g->dim[0]];
                for (int x = lo[0]; x < hi[0]; ++x) row[x - g->min[0]] = v;
            }
    };
    auto leaf = [&](size_t off, const int o[3]) {
        const uint8_t *vals = b.span(off + kLeafHeader, 512 * sizeof(float));
        for (int n = 0; n < 512; ++n) {   // LeafNode::CoordToOffset: x << 6 | y << 3 | z
            const long long c[3] = {(long long)o[0] + (n >> 6), (long long)o[1] + ((n >> 3) & 7), (long long)o[2] + (n & 7)};
            bool in = true;
            for (int a = 0; a < 3; ++a) in &= c[a] >= g->min[a] && c[a] < (long long)g->min[a] + g->dim[a];
            if (!in) continue;
            float v;
            memcpy(&v, vals + 4 * (size_t)n, 4);
            g->values[((size_t)(c[2] - g->min[2]) * g->dim[1] + (c[1] - g->min[1])) * g->dim[0] + (c[0] - g->min[0])] = v;
        }
    };
    // internal node at `off` with origin o: LOG2DIM 5 (children of 128^3) or 4 (children = leaves of 8^3)
    std::function<void(size_t, const int *, int)> internal = [&](size_t off, const int o[3], int log2dim) {
        const size_t nEntries = (size_t)1 << (3 * log2dim), maskBytes = nEntries / 8;
        const uint8_t *childMask = b.span(off + 24 + 8 + maskBytes, maskBytes);
        const size_t table = off + InternalHeader(log2dim);
        b.span(table, nEntries * 8);
        const int childSize = log2dim == 5 ? 128 : 8;
        for (size_t n = 0; n < nEntries; ++n) {
            const int dimMask = (1 << log2dim) - 1;
            const int co[3] = {o[0] + (int)((n >> (2 * log2dim)) & dimMask) * childSize, o[1] + (int)((n >> log2dim) & dimMask) * childSize,
                               o[2] + (int)(n & dimMask) * childSize};
            if (childMask[n >> 3] & (1u << (n & 7))) {
                const int64_t rel = b.at<int64_t>(table + 8 * n);
                if (rel <= 0 || (uint64_t)rel > b.n) Fail(b.fn, "corrupt child offset");
                if (log2dim == 5) internal(off + (size_t)rel, co, 4);
                else leaf(off + (size_t)rel, co);
            } else fill(co, childSize, b.at<float>(table + 8 * n));
        }
    };
    if (rd.tableSize > (1u << 20)) Fail(b.fn, "corrupt root table size");
    for (uint32_t t = 0; t < rd.tableSize; ++t) {
        const RootTile tile = b.at<RootTile>(rootOff + sizeof(RootData) + (size_t)t * sizeof(RootTile));
        // RootData::KeyToCoord: 21 bits per axis of (coordinate >> 12), x in the top field
        const int o[3] = {(int)(uint32_t)(((tile.key >> 42) & 0x1fffff) << 12), (int)(uint32_t)(((tile.key >> 21) & 0x1fffff) << 12), (int)(uint32_t)((tile.key & 0x1fffff) << 12)};
        if (tile.child != 0) {
            if (tile.child < 0 || (uint64_t)tile.child > b.n) Fail(b.fn, "corrupt root child offset");
            internal(rootOff + (size_t)tile.child, o, 5);
        } else fill(o, 4096, tile.value);
    }
}
}  // namespace

// nanovdb::io::readGrid(filename, gridName): the first grid of that name in the file; found = false when there is none
void ReadNanoVDBGrid(const std::string &fn, const std::string &gridName, VdbGrid *out) {
    *out = VdbGrid();
    std::vector<uint8_t> file;
    {
        FILE *f = fopen(fn.c_str(), "rb");
        if (!f) Fail(fn, "cannot open the file");
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        file.resize(n > 0 ? n : 0);
        if (n > 0 && fread(file.data(), 1, n, f) != (size_t)n) { fclose(f); Fail(fn, "read error"); }
        fclose(f);
    }
    size_t pos = 0;
    auto need = [&](size_t n) { if (pos > file.size() || file.size() - pos < n) Fail(fn, "file is truncated"); };
    while (pos < file.size()) {   // a file is a sequence of segments: header, the meta data of its grids, the grids
        need(sizeof(FileHeader));
        FileHeader h;
        memcpy(&h, &file[pos], sizeof(h));
        pos += sizeof(h);
        if (h.magic != kMagic) Fail(fn, "not a NanoVDB file (magic number)");
        const uint32_t major = h.version >> 21, minor = (h.version >> 10) & 0x7ff;
        if (major != 32) Fail(fn, "NanoVDB ABI version " + std::to_string(major) + "." + std::to_string(minor) + " is not supported by this build (32.x is)");
        std::vector<FileMetaData> metas(h.gridCount);
        std::vector<std::string> names(h.gridCount);
        for (int i = 0; i < h.gridCount; ++i) {
            need(sizeof(FileMetaData));
            memcpy(&metas[i], &file[pos], sizeof(FileMetaData));
            pos += sizeof(FileMetaData);
            need(metas[i].nameSize);
            names[i].assign((const char *)&file[pos], metas[i].nameSize ? metas[i].nameSize - 1 : 0);
            pos += metas[i].nameSize;
        }
        for (int i = 0; i < h.gridCount; ++i) {
            const FileMetaData &m = metas[i];
            const bool wanted = !out->found && names[i] == gridName;
            std::vector<uint8_t> grid;
            if (m.codec == 0) {
                need(m.gridSize);
                if (wanted) grid.assign(&file[pos], &file[pos] + m.gridSize);
                pos += m.gridSize;
            } else if (m.codec == 1) {   // ZIP: a 64-bit compressed size, then one zlib stream
                need(8);
                uint64_t csize;
                memcpy(&csize, &file[pos], 8);
                pos += 8;
                need(csize);
                if (wanted) {
                    if (m.gridSize > ((uint64_t)1 << 36)) Fail(fn, "implausible grid size");
                    grid.resize(m.gridSize);
                    uLongf outLen = (uLongf)m.gridSize;
                    if (uncompress(grid.data(), &outLen, &file[pos], (uLong)csize) != Z_OK || outLen != m.gridSize) Fail(fn, "corrupt ZIP-compressed grid");
                }
                pos += csize;
            } else Fail(fn, "the grids are compressed with codec " + std::to_string(m.codec) + " (BLOSC): only uncompressed and ZIP-compressed files are supported by this build");
            if (!wanted) continue;
            if (m.gridType != kGridTypeFloat) Fail(fn, "\"" + gridName + "\" is not a float grid");
            Blob b{grid.data(), grid.size(), fn};
            Densify(b, out);
            out->found = true;
            // readGrid's check (media.cpp:500-502)
            if (out->gridClass != (int)kClassFog && out->gridClass != (int)kClassUnknown) Fail(fn, "\"" + gridName + "\" isn't a FogVolume grid?");
        }
        if (out->found) return;
    }
}

}  // namespace wf
