// loopsubdiv.cpp — Shape "loopsubdiv": Loop subdivision of a closed or open triangle control mesh to its limit surface
// (util/loopsubdiv.cpp:133-397: neighbour finding, `levels` refinement steps with the even / odd / boundary rules, the limit
// positions, tangents from the one-ring).  The reference links vertices and faces by pointer; here they are indices into two
// growing arrays, visited in the same orders (face order for new odd vertices, the one-ring walk from startFace), so positions,
// normals and the output numbering are the reference's.
#include "scene.h"

#include <cmath>
#include <map>
#include <string>
#include <utility>

namespace wf {
namespace {
inline int NEXT(int i) { return (i + 1) % 3; }
inline int PREV(int i) { return (i + 2) % 3; }
struct SDV { V3 p{0, 0, 0}; int startFace = -1, child = -1; bool regular = false, boundary = false; };
struct SDF { int v[3] = {-1, -1, -1}, f[3] = {-1, -1, -1}, children[4] = {-1, -1, -1, -1}; };
struct Mesh {
    std::vector<SDV> V;
    std::vector<SDF> F;
    int vnum(int f, int vert) const { for (int i = 0; i < 3; ++i) if (F[f].v[i] == vert) return i; throw SceneError("Error: Basic logic error in SDFace::vnum()"); }
    int nextFace(int f, int vert) const { return F[f].f[vnum(f, vert)]; }
    int prevFace(int f, int vert) const { return F[f].f[PREV(vnum(f, vert))]; }
    int nextVert(int f, int vert) const { return F[f].v[NEXT(vnum(f, vert))]; }
    int prevVert(int f, int vert) const { return F[f].v[PREV(vnum(f, vert))]; }
    int otherVert(int f, int v0, int v1) const { for (int i = 0; i < 3; ++i) if (F[f].v[i] != v0 && F[f].v[i] != v1) return F[f].v[i]; throw SceneError("Error: Basic logic error in SDFace::otherVert()"); }
    int valence(int vert) const {
        int f = V[vert].startFace;
        if (!V[vert].boundary) {
            int nf = 1;
            while ((f = nextFace(f, vert)) != V[vert].startFace) ++nf;
            return nf;
        }
        int nf = 1;
        while ((f = nextFace(f, vert)) != -1) ++nf;
        f = V[vert].startFace;
        while ((f = prevFace(f, vert)) != -1) ++nf;
        return nf + 1;
    }
    void oneRing(int vert, std::vector<V3> *ring) const {
        ring->clear();
        if (!V[vert].boundary) {
            int face = V[vert].startFace;
            do {
                ring->push_back(V[nextVert(face, vert)].p);
                face = nextFace(face, vert);
            } while (face != V[vert].startFace);
        } else {
            int face = V[vert].startFace, f2;
            while ((f2 = nextFace(face, vert)) != -1) face = f2;
            ring->push_back(V[nextVert(face, vert)].p);
            do {
                ring->push_back(V[prevVert(face, vert)].p);
                face = prevFace(face, vert);
            } while (face != -1);
        }
    }
    V3 weightOneRing(int vert, float beta) const {
        std::vector<V3> ring;
        const int val = valence(vert);
        oneRing(vert, &ring);
        V3 p = (1 - val * beta) * V[vert].p;
        for (int i = 0; i < val; ++i) p = p + beta * ring[i];
        return p;
    }
    V3 weightBoundary(int vert, float beta) const {
        std::vector<V3> ring;
        const int val = valence(vert);
        oneRing(vert, &ring);
        V3 p = (1 - 2 * beta) * V[vert].p;
        p = p + beta * ring[0];
        p = p + beta * ring[val - 1];
        return p;
    }
};
inline float Beta(int valence) { return valence == 3 ? 3.f / 16.f : 3.f / (8.f * valence); }
inline float LoopGamma(int valence) { return 1.f / (valence + 3.f / (8.f * Beta(valence))); }
typedef std::pair<int, int> EdgeKey;
inline EdgeKey Edge(int a, int b) { return a < b ? EdgeKey(a, b) : EdgeKey(b, a); }
}  // namespace

void LoopSubdivide(int nLevels, const std::vector<int> &vertexIndices, const std::vector<V3> &P, std::vector<int> *outIndices,
                   std::vector<V3> *outP, std::vector<V3> *outN) {
    Mesh m;
    m.V.resize(P.size());
    for (size_t i = 0; i < P.size(); ++i) m.V[i].p = P[i];
    const size_t nFaces = vertexIndices.size() / 3;
    m.F.resize(nFaces);
    for (size_t i = 0; i < nFaces; ++i)
        for (int j = 0; j < 3; ++j) {
            const int v = vertexIndices[3 * i + j];
            if (v < 0 || (size_t)v >= P.size()) throw SceneError("Error: loopsubdiv: vertex index out of range");
            m.F[i].v[j] = v;
            m.V[v].startFace = (int)i;
        }
    // (the reference dereferences a null startFace for a control vertex that no face uses, and links only the first two faces of an
    // edge: both are input errors here, reported instead of crashing the embedding process)
    for (size_t i = 0; i < P.size(); ++i)
        if (m.V[i].startFace < 0) throw SceneError("Error: loopsubdiv: control vertex " + std::to_string(i) + " is not used by any face");
    // neighbour pointers: an edge is remembered until its second face arrives
    {
        struct Half { int f, edgeNum; };
        std::map<EdgeKey, Half> edges;
        std::map<EdgeKey, int> uses;
        for (size_t i = 0; i < nFaces; ++i)
            for (int edgeNum = 0; edgeNum < 3; ++edgeNum) {
                const EdgeKey e = Edge(m.F[i].v[edgeNum], m.F[i].v[NEXT(edgeNum)]);
                if (e.first == e.second) throw SceneError("Error: loopsubdiv: degenerate face (repeated vertex)");
                if (++uses[e] > 2) throw SceneError("Error: loopsubdiv: non-manifold control mesh (an edge is shared by more than two faces)");
                auto it = edges.find(e);
                if (it == edges.end()) edges[e] = Half{(int)i, edgeNum};
                else {
                    m.F[it->second.f].f[it->second.edgeNum] = (int)i;
                    m.F[i].f[edgeNum] = it->second.f;
                    edges.erase(it);
                }
            }
    }
    for (size_t i = 0; i < P.size(); ++i) {
        SDV &v = m.V[i];
        if (v.startFace < 0) continue;   // (a vertex no face uses: the reference would dereference a null startFace)
        int f = v.startFace;
        do { f = m.nextFace(f, (int)i); } while (f != -1 && f != v.startFace);
        v.boundary = (f == -1);
        const int val = m.valence((int)i);
        v.regular = (!v.boundary && val == 6) || (v.boundary && val == 4);
    }
    std::vector<int> f(nFaces), v(P.size());
    for (size_t i = 0; i < nFaces; ++i) f[i] = (int)i;
    for (size_t i = 0; i < P.size(); ++i) v[i] = (int)i;
    for (int level = 0; level < nLevels; ++level) {
        std::vector<int> newFaces, newVertices;
        for (int vert : v) {
            const int c = (int)m.V.size();
            m.V.push_back(SDV());
            m.V[vert].child = c;
            m.V[c].regular = m.V[vert].regular;
            m.V[c].boundary = m.V[vert].boundary;
            newVertices.push_back(c);
        }
        for (int face : f)
            for (int k = 0; k < 4; ++k) {
                const int c = (int)m.F.size();
                m.F.push_back(SDF());
                m.F[face].children[k] = c;
                newFaces.push_back(c);
            }
        // even vertices
        for (int vert : v) {
            V3 p;
            if (!m.V[vert].boundary) p = m.V[vert].regular ? m.weightOneRing(vert, 1.f / 16.f) : m.weightOneRing(vert, Beta(m.valence(vert)));
            else p = m.weightBoundary(vert, 1.f / 8.f);
            m.V[m.V[vert].child].p = p;
        }
        // odd (edge) vertices, created in face order
        std::map<EdgeKey, int> edgeVerts;
        for (int face : f)
            for (int k = 0; k < 3; ++k) {
                const int a = m.F[face].v[k], b = m.F[face].v[NEXT(k)];
                const EdgeKey edge = Edge(a, b);
                if (edgeVerts.count(edge)) continue;
                const int vert = (int)m.V.size();
                m.V.push_back(SDV());
                newVertices.push_back(vert);
                m.V[vert].regular = true;
                m.V[vert].boundary = (m.F[face].f[k] == -1);
                m.V[vert].startFace = m.F[face].children[3];
                // (the reference orders the edge's two ends by address; a + b is the same either way)
                const V3 p0 = m.V[edge.first].p, p1 = m.V[edge.second].p;
                V3 p;
                if (m.V[vert].boundary) { p = 0.5f * p0; p = p + 0.5f * p1; }
                else {
                    p = 3.f / 8.f * p0;
                    p = p + 3.f / 8.f * p1;
                    p = p + 1.f / 8.f * m.V[m.otherVert(face, a, b)].p;
                    p = p + 1.f / 8.f * m.V[m.otherVert(m.F[face].f[k], a, b)].p;
                }
                m.V[vert].p = p;
                edgeVerts[edge] = vert;
            }
        // topology of the new level
        for (int vert : v) {
            const int vertNum = m.vnum(m.V[vert].startFace, vert);
            m.V[m.V[vert].child].startFace = m.F[m.V[vert].startFace].children[vertNum];
        }
        for (int face : f)
            for (int j = 0; j < 3; ++j) {
                const SDF &F = m.F[face];
                m.F[F.children[3]].f[j] = F.children[NEXT(j)];
                m.F[F.children[j]].f[NEXT(j)] = F.children[3];
                int f2 = F.f[j];
                m.F[F.children[j]].f[j] = f2 != -1 ? m.F[f2].children[m.vnum(f2, F.v[j])] : -1;
                f2 = F.f[PREV(j)];
                m.F[F.children[j]].f[PREV(j)] = f2 != -1 ? m.F[f2].children[m.vnum(f2, F.v[j])] : -1;
            }
        for (int face : f)
            for (int j = 0; j < 3; ++j) {
                const SDF &F = m.F[face];
                m.F[F.children[j]].v[j] = m.V[F.v[j]].child;
                const int vert = edgeVerts[Edge(F.v[j], F.v[NEXT(j)])];
                m.F[F.children[j]].v[NEXT(j)] = vert;
                m.F[F.children[NEXT(j)]].v[j] = vert;
                m.F[F.children[3]].v[j] = vert;
            }
        f.swap(newFaces);
        v.swap(newVertices);
    }
    // limit surface
    std::vector<V3> pLimit(v.size());
    for (size_t i = 0; i < v.size(); ++i)
        pLimit[i] = m.V[v[i]].boundary ? m.weightBoundary(v[i], 1.f / 5.f) : m.weightOneRing(v[i], LoopGamma(m.valence(v[i])));
    for (size_t i = 0; i < v.size(); ++i) m.V[v[i]].p = pLimit[i];
    // tangents -> normals
    outN->clear();
    std::vector<V3> ring;
    for (int vert : v) {
        V3 S{0, 0, 0}, T{0, 0, 0};
        const int valence = m.valence(vert);
        m.oneRing(vert, &ring);
        if (!m.V[vert].boundary) {
            for (int j = 0; j < valence; ++j) {
                S = S + std::cos(2 * Pi * j / valence) * ring[j];
                T = T + std::sin(2 * Pi * j / valence) * ring[j];
            }
        } else {
            S = ring[valence - 1] - ring[0];
            if (valence == 2) T = ring[0] + ring[1] - 2 * m.V[vert].p;
            else if (valence == 3) T = ring[1] - m.V[vert].p;
            else if (valence == 4) T = -1 * ring[0] + 2 * ring[1] + 2 * ring[2] + -1 * ring[3] + -2 * m.V[vert].p;
            else {
                float theta = Pi / float(valence - 1);
                T = std::sin(theta) * (ring[0] + ring[valence - 1]);
                for (int k = 1; k < valence - 1; ++k) {
                    float wt = (2 * std::cos(theta) - 2) * std::sin((k)*theta);
                    T = T + wt * ring[k];
                }
                T = -T;
            }
        }
        outN->push_back(Cross(S, T));
    }
    // output numbering: the last level's vertices in their list order
    std::map<int, int> usedVerts;
    for (size_t i = 0; i < v.size(); ++i) usedVerts[v[i]] = (int)i;
    outIndices->clear();
    for (int face : f)
        for (int j = 0; j < 3; ++j) outIndices->push_back(usedVerts[m.F[face].v[j]]);
    *outP = pLimit;
}

}  // namespace wf
